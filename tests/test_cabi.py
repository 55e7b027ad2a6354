"""CPU tests of the drop-in boundary: libgem_hip.so builds for gfx950, loads without a GPU, exports
every symbol include/gem_hip.h declares, its structs match the ctypes mirrors byte for byte, it
fails loudly without a device (no CPU fallback), and the C++ host side (facade + the nine
GEM-signature adapter symbols) compiles and links against it.  No compute calls here."""
import ctypes as C
import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "gem_hip.h"


@pytest.fixture(scope="module")
def lib():
    from gem_amd import _lib
    return _lib.load()


def declared_symbols(header=HEADER):
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(gem_[a-z_0-9]+)\s*\(", text)))


def test_library_is_in_tree_and_built_for_gfx950(tmp_path):
    from gem_amd import build
    path = build.build()
    assert path.exists() and ROOT in path.parents
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    # llvm-objdump --offloading unbundles the code objects NEXT TO its input: run it on a copy outside the tree
    copy = tmp_path / path.name
    shutil.copy(path, copy)
    out = subprocess.run([objdump, "--offloading", str(copy)], capture_output=True, text=True, cwd=tmp_path).stdout
    assert "gfx950" in out, out[:500]


def test_every_declared_symbol_is_exported_and_bound(lib):
    from gem_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gem_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype"
    assert lib.gem_abi_version() == 9
    dbg = declared_symbols(ROOT / "include" / "gem_hip_debug.h")         # knobs / profiling aids: exported, bound, not in gem_hip.h
    assert dbg and not set(dbg) & set(names)
    for n in dbg:
        assert hasattr(lib, n) and n in _lib.DEBUG_SIGNATURES, n


def test_the_product_reads_no_environment_variables():
    for p in list((ROOT / "gem_amd" / "csrc").glob("*")) + [ROOT / "gem_amd" / "api.py", ROOT / "gem_amd" / "_lib.py", ROOT / "gem_amd" / "tiling.py"]:
        text = p.read_text(errors="ignore")
        assert "getenv" not in text and "os.environ" not in text, p


def test_every_debug_key_is_documented_in_the_debug_header():
    """gem_debug_set / gem_debug_get accept a closed list of keys (gem_capi.cpp); include/gem_hip_debug.h names every one of them."""
    import re
    src = "".join((ROOT / "gem_amd" / "csrc" / f).read_text() for f in ("gem_capi.cpp", "gem_capi_core.cpp", "gem_capi_pipeline.cpp", "gem_capi_comm.cpp"))
    hdr = (ROOT / "include" / "gem_hip_debug.h").read_text()
    keys = sorted(set(re.findall(r'k == "([a-z_0-9]+)"', src)))
    assert len(keys) >= 40
    assert [k for k in keys if f'"{k}"' not in hdr] == []


def test_the_copy_thread_pool_moves_every_byte(tmp_path):
    """gem_amd/csrc/gem_hostcopy.hpp on its own (tests/cpp/hostcopy_pool.cpp): thread counts 1..8, sizes around the piece size, workers
    polling and asleep, concurrent callers."""
    exe = tmp_path / "hostcopy_pool"
    res = subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-pthread", str(ROOT / "tests" / "cpp" / "hostcopy_pool.cpp"), "-o", str(exe)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip() == "ok", run.stdout + run.stderr


def test_struct_layouts_match_the_header(tmp_path):
    from gem_amd import _lib
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gem_hip.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(gem_map_config), sizeof(gem_frame_params),'
                   ' offsetof(gem_frame_params, lower), offsetof(gem_frame_params, sensor_params), offsetof(gem_frame_params, sensor_jacobian),'
                   ' offsetof(gem_frame_params, filter), offsetof(gem_frame_params, original_width), sizeof(gem_stats));return 0;}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    got = list(map(int, subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()))
    F = _lib.FrameParams
    exp = [C.sizeof(_lib.MapConfig), C.sizeof(F), F.lower.offset, F.sensor_params.offset, F.sensor_jacobian.offset,
           F.filter.offset, F.original_width.offset, C.sizeof(_lib.Stats)]
    assert got == exp


def test_no_cpu_fallback(lib):
    from conftest import HAS_GPU
    if HAS_GPU:
        pytest.skip("a HIP device is present")
    from gem_amd import ElevationMap, GemError
    with pytest.raises(GemError, match="no HIP device"):
        ElevationMap(64, 0.1)


def test_product_never_imports_the_oracle():
    for p in list((ROOT / "gem_amd").rglob("*.py")) + list((ROOT / "gem_amd" / "csrc").glob("*")) + list((ROOT / "include").rglob("*.h*")):
        text = p.read_text(errors="ignore")
        assert "gem_oracle" not in text and "import oracle" not in text and "libgem_oracle" not in text, p


def build_facade_check(tmp_path) -> Path:
    exe = tmp_path / "facade_check"
    libdir = ROOT / "gem_amd" / "lib"
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", str(ROOT / "include"), "-I", str(ROOT / "tests" / "cpp" / "fake_eigen"),
           str(ROOT / "tests" / "cpp" / "facade_check.cpp"), "-o", str(exe), f"-L{libdir}", "-lgem_hip",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_cpp_host_side_links_and_fails_loudly_without_gpu(lib, tmp_path):
    from conftest import HAS_GPU
    exe = build_facade_check(tmp_path)
    nm = subprocess.run(["nm", "-C", str(exe)], capture_output=True, text=True).stdout
    for sym in ("Init_GPU_elevationmap(int, float, float, float)", "Move(float*, float, int, float*, int*, float*)",
                "Fuse(int, int, int*, int*, int*, int*, float*, float*, float*)", "Mapvar_update(int, float)",
                "Map_feature(", "Raytracing(int)", "Map_optmove(", "Map_closeloop(", "Process_points("):
        assert sym in nm, f"adapter symbol {sym} missing"
    if not HAS_GPU:
        res = subprocess.run([str(exe), "0"], capture_output=True, text=True)
        assert res.returncode == 0, res.stdout + res.stderr


@pytest.mark.gpu
@pytest.mark.one_pipeline            # (a C++ child process: the fixture's knobs never reach it)
def test_cpp_host_side_known_answers_on_gpu(lib, tmp_path):
    exe = build_facade_check(tmp_path)
    res = subprocess.run([str(exe), "1"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr


# ---- the boundary from the CALLERS' side -------------------------------------------------------------------------
REF_SRC = Path("/root/reference/elevation_mapping/elevation_mapping/src")
CALLER_DECLS = {          # where the reference's callers forward-declare the nine libgpu.so functions (no shared header exists)
    "ElevationMapping.cpp": ["Move", "Init_GPU_elevationmap", "Map_closeloop", "Raytracing", "Fuse", "Map_feature", "Map_optmove"],
    "sensor_processors/SensorProcessorBase.cpp": ["Process_points"],
    "RobotMotionMapUpdater.cpp": ["Mapvar_update"],
}


def callers_declarations():
    """The forward declarations, read from the reference WHERE IT LIES at test time (nothing of it is committed)."""
    out = []
    for rel, names in CALLER_DECLS.items():
        text = (REF_SRC / rel).read_text(errors="ignore")
        for n in names:
            m = re.search(r"^(?:void|int)\s+" + n + r"\s*\([^;{]*\)\s*;", text, flags=re.M)
            assert m, f"{n} is not forward-declared in {rel} any more"
            out.append(m.group(0))
    return out


def build_adapter_object(tmp_path) -> Path:
    src = tmp_path / "adapter.cpp"
    src.write_text('#include "gem/gem_compat_eigen.hpp"\n')
    obj = tmp_path / "adapter.o"
    res = subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-c", "-I", str(ROOT / "include"), "-I", str(ROOT / "tests" / "cpp" / "fake_eigen"),
                          str(src), "-o", str(obj)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return obj


def test_the_callers_own_declarations_link_against_the_adapter(lib, tmp_path):
    """A translation unit holding the reference callers' OWN forward declarations (ElevationMapping.cpp:44-50,
    SensorProcessorBase.cpp:34, RobotMotionMapUpdater.cpp:18) calls all nine functions and is linked against the adapter object
    (include/gem/gem_compat_eigen.hpp), with one Eigen stand-in on both sides: C++ mangling carries every parameter type, so a link
    without unresolved symbols means the nine signatures are the callers' -- including Mapvar_update, declared `int` by its caller
    and defined `void` by the library (the mangled name has no return type)."""
    if not REF_SRC.exists():
        pytest.skip("no /root/reference on this box")
    decls = callers_declarations()
    assert len(decls) == 9
    calls = """
int call_all(int n)
{
    float f3[3] = {0, 0, 0}, f2[2] = {0, 0}; int i2[2] = {0, 0};
    std::vector<int> iv(n > 0 ? n : 1); std::vector<float> fv(n > 0 ? n : 1);
    Init_GPU_elevationmap(64, 0.1f, 5.0f, 0.7f);
    Move(f3, 0.1f, 64, f2, i2, f2);
    Process_points(iv.data(), fv.data(), fv.data(), fv.data(), fv.data(), fv.data(), fv.data(), fv.data(), Eigen::Matrix4f(), n, -1.0, 1.0, 0.018f, 0.0006f, 0.0015f,
                   Eigen::RowVector3f(), Eigen::Matrix3f(), Eigen::Matrix3f(), Eigen::RowVector3f(), Eigen::Matrix3f());
    Fuse(64, n, iv.data(), iv.data(), iv.data(), iv.data(), fv.data(), fv.data(), fv.data());
    const int r = Mapvar_update(64, 1e-6f);
    Map_feature(64, fv.data(), fv.data(), iv.data(), iv.data(), iv.data(), fv.data(), fv.data(), fv.data(), fv.data());
    Raytracing(64);
    Map_optmove(f2, 0.0f, 0.1f, 64, f2);
    Map_closeloop(f2, 0.0f, 64, 0.1f);
    return r;
}
int main(int argc, char**) { return argc > 100 ? call_all(argc) : 0; }       // (linked, not run: this is a link test)
"""
    src = tmp_path / "callers.cpp"
    src.write_text("#include <Eigen/Core>\n#include <vector>\n" + "\n".join(decls) + "\n" + calls)
    obj = tmp_path / "callers.o"
    res = subprocess.run(["g++", "-std=c++17", "-O0", "-c", "-I", str(ROOT / "tests" / "cpp" / "fake_eigen"), str(src), "-o", str(obj)], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    adapter = build_adapter_object(tmp_path)
    libdir = ROOT / "gem_amd" / "lib"
    exe = tmp_path / "callers"
    res = subprocess.run(["g++", str(obj), str(adapter), "-o", str(exe), f"-L{libdir}", "-lgem_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"],
                         capture_output=True, text=True)
    assert res.returncode == 0, "the callers' declarations do not resolve against the adapter:\n" + res.stderr
    undefined = subprocess.run(["nm", "-C", "-u", str(obj)], capture_output=True, text=True).stdout
    for n in sum(CALLER_DECLS.values(), []):
        assert n + "(" in undefined, f"{n} is not referenced by the callers' translation unit"
    assert subprocess.run([str(exe)], capture_output=True).returncode == 0


@pytest.mark.gpu
@pytest.mark.one_pipeline            # (a C++ child process: the fixture's knobs never reach it)
def test_reference_motion_updater_drives_the_adapters_mapvar_update(lib, tmp_path, oracle_mod):
    """The reference's own RobotMotionMapUpdater::update (compiled from where it lies, oracle/_ref/libgem_ref_motion.so) computes the
    increments; each is handed to the ADAPTER's Mapvar_update (the symbol RMU.cpp:81 calls) on a populated GPU map; the variance layer
    read back through the adapter's Map_feature must equal the oracle's after the same calls."""
    import numpy as np
    import ref
    from gem_amd import synth
    if ref.motion_lib() is None:
        pytest.skip("no /root/reference to build from and no prebuilt oracle/_ref/libgem_ref_motion.so")
    adapter = build_adapter_object(tmp_path)
    libdir = ROOT / "gem_amd" / "lib"
    so = tmp_path / "libadapter.so"
    res = subprocess.run(["g++", "-shared", str(adapter), "-o", str(so), f"-L{libdir}", "-lgem_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    names = {}
    for line in subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True).stdout.splitlines():
        sym = line.split()[-1]
        for n in ("Init_GPU_elevationmap", "Fuse", "Mapvar_update", "Map_feature"):
            if re.fullmatch(r"_Z\d+" + n + r"[a-zA-Z_0-9]*", sym):
                names[n] = sym
    assert len(names) == 4, names
    ad = C.CDLL(str(so))
    L, res_m = 96, 0.1
    F, I = C.POINTER(C.c_float), C.POINTER(C.c_int)
    getattr(ad, names["Init_GPU_elevationmap"]).argtypes = [C.c_int, C.c_float, C.c_float, C.c_float]
    getattr(ad, names["Fuse"]).argtypes = [C.c_int, C.c_int, I, I, I, I, F, F, F]
    getattr(ad, names["Mapvar_update"]).argtypes = [C.c_int, C.c_float]
    getattr(ad, names["Map_feature"]).argtypes = [C.c_int, F, F, I, I, I, F, F, F, F]
    getattr(ad, names["Init_GPU_elevationmap"])(L, res_m, 5.0, 0.7)
    rng = np.random.default_rng(11)
    n = 5000
    idx = rng.integers(0, L * L, n).astype(np.int32); hgt = rng.normal(0, 0.2, n).astype(np.float32); var = rng.uniform(3e-4, 2e-3, n).astype(np.float32)
    zi = np.zeros(n, np.int32); zf = np.zeros(n, np.float32)
    p = lambda a, t: a.ctypes.data_as(t)
    getattr(ad, names["Fuse"])(L, n, p(idx, I), p(zi, I), p(zi, I), p(zi, I), p(zf, F), p(hgt, F), p(var, F))
    om = oracle_mod.OracleMap(L, res_m)
    om.fuse(idx, hgt, var)
    rm = ref.RefMotion(1.3, L)
    pos = np.zeros(3)
    for k in range(12):
        pos = pos + rng.normal(0, 0.2, 3)
        R = synth.rot_zyx(0.1 * k, rng.normal(0, 0.05), rng.normal(0, 0.05))
        a = rng.normal(size=(6, 6)) * 1e-2
        u = rm.compute(pos, R, a @ a.T * (1 + 0.1 * k))
        if u is None:
            continue
        getattr(ad, names["Mapvar_update"])(L, float(u))
        om.mapvar_update(float(u))
    out = [np.zeros(L * L, np.float32) for _ in range(6)]; outi = [np.zeros(L * L, np.int32) for _ in range(3)]
    getattr(ad, names["Map_feature"])(L, p(out[0], F), p(out[1], F), p(outi[0], I), p(outi[1], I), p(outi[2], I), p(out[2], F), p(out[3], F), p(out[4], F), p(out[5], F))
    assert np.array_equal(out[0].reshape(L, L), om.layer("elevation")) and np.array_equal(out[1].reshape(L, L), om.layer("variance"))
    assert (om.layer("variance") > 1e-4).sum() > 1000


# ---- the C++ header's host math against the reference's own code ---------------------------------------------------
def test_cpp_header_host_math_matches_the_references_code(lib, tmp_path):
    """include/gem/gem.hpp -- gem::RobotMotionMapUpdater::compute and gem::SensorProcessorBase::frameParams, what a C++ user of the
    header gets -- against RobotMotionMapUpdater.cpp (RMU.cpp:42-145) and SensorProcessorBase::readcomputerparam (SPB.cpp:270-290)
    compiled where they lie (oracle/_ref, oracle/ref_build/build_ref.py): random trajectories / transformations, <= 1 float ulp
    (tests/cpp/header_pin.cpp).  The header's Python twin has tests/test_motion_update.py."""
    sys.path.insert(0, str(ROOT / "oracle" / "ref_build"))
    import build_ref
    motion, sensors = build_ref.build_motion(), build_ref.build_sensors()
    if motion is None or sensors is None:
        pytest.skip("no /root/reference to build from and no prebuilt oracle/_ref libraries")
    if b"gemref_readcomputerparam" not in Path(sensors).read_bytes():
        pytest.skip("the prebuilt oracle/_ref/libgem_ref_sensors.so predates gemref_readcomputerparam and /root/reference is not here to rebuild it")
    exe = tmp_path / "header_pin"
    libdir = ROOT / "gem_amd" / "lib"
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wall", "-Werror", "-I", str(ROOT / "include"), str(ROOT / "tests" / "cpp" / "header_pin.cpp"), "-o", str(exe),
           f"-L{libdir}", "-lgem_hip", "-ldl", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([str(exe), str(motion), str(sensors)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.strip().endswith("ok"), run.stdout + run.stderr


@pytest.mark.gpu
@pytest.mark.one_pipeline
def test_roctx_ranges_around_the_entry_points(oracle_mod):
    """gem_debug_set "roctx" loads the ROCm marker library at run time and wraps the entry points in ranges;
    the map is what it is without them, and switching them off again works.  (What a profiler shows: profiles/r06_roctx_trace.txt.)"""
    import numpy as np
    import torch
    from gem_amd import ElevationMap, synth
    wl = synth.config_c1()
    ref = oracle_mod.OracleMap(wl.length, wl.resolution)
    m = ElevationMap(wl.length, wl.resolution)
    m.debug_set("roctx", 1)
    d = torch.from_numpy(wl.clouds[0]).cuda()
    for _ in range(2):
        m.mapvar_update(1e-5); ref.mapvar_update(1e-5)
        m.add(wl.frames[0], d); ref.add(wl.frames[0], wl.clouds[0])
    m.debug_set("roctx", 0)
    m.add(wl.frames[0], d); ref.add(wl.frames[0], wl.clouds[0])
    assert np.array_equal(m.layer("elevation"), ref.layer("elevation")) and np.array_equal(m.layer("variance"), ref.layer("variance"))
    m.close()
