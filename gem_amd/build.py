"""Builds libgem_hip.so (gfx950) in-tree with hipcc.

The shared library is the product: hand-written HIP kernels + the C ABI of include/gem_hip.h.
It is built into gem_amd/lib/ so that it travels with the source tree (no JIT cache).
hipcc cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent
CSRC = ROOT / "csrc"
LIBDIR = ROOT / "lib"
LIB = LIBDIR / "libgem_hip.so"

SOURCES = [CSRC / "gem_kernels.hip", CSRC / "gem_sort.hip", CSRC / "gem_capi.cpp", CSRC / "gem_capi_core.cpp", CSRC / "gem_capi_pipeline.cpp", CSRC / "gem_capi_comm.cpp"]
HEADERS = [CSRC / "gem_device.hpp", CSRC / "gem_kernels.hpp", CSRC / "gem_wave.hpp", CSRC / "gem_transport.hpp", CSRC / "gem_hostcopy.hpp", CSRC / "gem_capi_internal.hpp",
           ROOT.parent / "include" / "gem_hip_debug.h", ROOT.parent / "include" / "gem_hip.h"]

# -ffp-contract=off: cell indices must be bit-exact with the reference arithmetic, so no product+sum
# may be contracted into an FMA (see csrc/gem_device.hpp).  hipcc's default IEEE divide/sqrt stay on.
FLAGS = [
    "--offload-arch=gfx950", "-O3", *(["-g"] if os.environ.get("GEM_DEBUG_BUILD") else []), "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
    "-Wno-unused-value", "-Wno-unused-result",
    *os.environ.get("GEM_BUILD_DEFINES", "").split(),          # build-time experiments (A/B builds on the GPU box), e.g. -DGEM_X=1
]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libgem_hip.so)")


def is_stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS)


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile libgem_hip.so if missing or older than its sources.  Returns its path.
    One object per source (gem_amd/lib/obj, compiled side by side; only what changed is recompiled), then one link."""
    if not force and not is_stale():
        return LIB
    LIBDIR.mkdir(exist_ok=True)
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cflags = [f for f in FLAGS if f != "-shared"]
    tag = objdir / ".flags"
    flag_line = " ".join(cflags)
    if not tag.exists() or tag.read_text() != flag_line:               # different flags (GEM_BUILD_DEFINES, GEM_DEBUG_BUILD): everything again
        force = True
    jobs = []
    for src in SOURCES:
        obj = objdir / (src.stem + ".o")
        if force or _stale(obj, [src] + HEADERS):
            cmd = [hipcc_path(), *cflags, "-c", str(src), "-o", str(obj)]
            if verbose:
                print(" ".join(cmd))
            jobs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    errors = []
    for src, proc in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            errors.append(f"{src.name}: hipcc failed ({proc.returncode}):\n{out}")
    if errors:
        raise RuntimeError("\n".join(errors))
    tag.write_text(flag_line)
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *[str(objdir / (s.stem + ".o")) for s in SOURCES], "-o", str(LIB),
           f"-L{rocm}/lib", "-lrccl", "-pthread", f"-Wl,-rpath,{rocm}/lib"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc (link) failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    return LIB


def build_variant(name: str, defines: str, verbose: bool = False) -> Path:
    """An A/B build with extra -D flags into gem_amd/lib_ab/<name>/libgem_hip.so (tools only: tools/ab_run.sh copies it over the
    library of the GPU box's scratch copy of the tree, variant by variant; the product library here is not touched)."""
    out = ROOT / "lib_ab" / name
    objdir = out / "obj"
    objdir.mkdir(parents=True, exist_ok=True)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cflags = [f for f in FLAGS if f != "-shared"] + defines.split()
    jobs = []
    for src in SOURCES:
        obj = objdir / (src.stem + ".o")
        cmd = [hipcc_path(), *cflags, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        jobs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, proc in jobs:
        o, _ = proc.communicate()
        if proc.returncode != 0:
            raise RuntimeError(f"{src.name}: hipcc failed ({proc.returncode}):\n{o}")
    lib = out / "libgem_hip.so"
    res = subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *[str(objdir / (s.stem + ".o")) for s in SOURCES], "-o", str(lib),
                          f"-L{rocm}/lib", "-lrccl", "-pthread", f"-Wl,-rpath,{rocm}/lib"], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc (link) failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    return lib


if __name__ == "__main__":
    import sys
    if len(sys.argv) >= 3 and sys.argv[1] == "--variant":          # python -m gem_amd.build --variant NAME "-DX=1 -DY=2"
        print(build_variant(sys.argv[2], " ".join(sys.argv[3:]), verbose=True))
    else:
        print(build(force=True, verbose=True))
