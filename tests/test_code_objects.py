"""The built gfx950 code objects themselves: no kernel of libgem_hip.so spills a vector register or touches scratch.

A spill is scratch traffic in the middle of a latency-bound chain (round 5 found 5 MB of it per C2 frame behind ONE spilled address),
and it appears silently when a register budget (`__launch_bounds__`) and a kernel's body drift apart.  The metadata of every kernel is
read straight from the library (tools/code_objects.py: ELF notes, msgpack), the instruction streams from llvm-objdump where ROCm's LLVM
tools are present (this image and the GPU box).  CPU-only: nothing is launched."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))

import code_objects  # noqa: E402


@pytest.fixture(scope="module")
def lib_path():
    from gem_amd import build
    return build.build(force=False)


@pytest.fixture(scope="module")
def kernels(lib_path):
    ks = code_objects.all_kernels(lib_path)
    assert len(ks) > 100, f"only {len(ks)} kernels found in {lib_path}: the fat binary was not read"
    return ks


def test_only_gfx950_code_objects(lib_path):
    fb = code_objects.fatbin(lib_path)
    triples = set()
    at = 0
    import struct
    while True:
        at = fb.find(code_objects.MAGIC, at)
        if at < 0:
            break
        n, = struct.unpack_from("<Q", fb, at + len(code_objects.MAGIC))
        p = at + len(code_objects.MAGIC) + 8
        for _ in range(n):
            _off, size, tlen = struct.unpack_from("<QQQ", fb, p)
            t = fb[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if t.startswith("hip") and size:
                triples.add(t)
        at += len(code_objects.MAGIC)
    assert triples and all("gfx950" in t for t in triples), triples


def test_no_kernel_spills_vector_registers(kernels):
    """`.vgpr_spill_count == 0` for EVERY kernel -- the nine GEM symbols reach all of them through one knob or another (pipelines,
    colours, lowest tracking, tile shift), so there is no allow-list."""
    bad = [(k["name"], k["vgpr_spill"], k["scratch"]) for k in kernels if k["vgpr_spill"] != 0]
    assert not bad, "kernels that spill VGPRs (name, spilled, scratch bytes):\n" + "\n".join(map(str, bad))


def test_private_segment_only_where_scalars_were_spilled(kernels):
    """A kernel without VGPR spills may still carry a private-segment size: the frame slots of spilled SCALAR registers, which the
    backend then keeps in lanes of a VGPR (v_writelane / v_readlane) -- no memory involved.  Anything else is a stack object or a
    call frame, and there must be none."""
    bad = [(k["name"], k["scratch"]) for k in kernels if k["scratch"] != 0 and k["sgpr_spill"] == 0]
    assert not bad, "kernels with a private segment but no SGPR spills (a stack object?):\n" + "\n".join(map(str, bad))


def test_no_scratch_instruction_in_any_kernel(lib_path, kernels):
    """... and the instruction streams agree: not one scratch_* (or private buffer_*) instruction in the library."""
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not Path(objdump).exists():
        pytest.skip("no llvm-objdump here")
    import tempfile
    hits = {}
    with tempfile.TemporaryDirectory() as td:
        for i, (_triple, elf) in enumerate(code_objects.code_objects(lib_path)):
            f = Path(td) / f"co{i}.elf"
            f.write_bytes(elf)
            r = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(f)], capture_output=True, text=True, check=True)
            cur = None
            for line in r.stdout.splitlines():
                if line.endswith(">:") and "<" in line:
                    cur = line[line.index("<") + 1:-2]
                    continue
                s = line.strip()
                if not s or cur is None:
                    continue
                op = s.split()[0]
                if op.startswith("scratch_") or (op.startswith("buffer_") and "offen" in s and " s[0:3]" in s):
                    hits[cur] = hits.get(cur, 0) + 1
    assert not hits, f"kernels with scratch instructions: {hits}"


def test_register_budgets_of_the_hot_kernels(kernels):
    """The budgets DESIGN.md quotes: k_frame at six workgroups per CU (<= 80 VGPRs), the laser-only projection at four waves per SIMD."""
    by = {k["name"]: k for k in kernels}

    def find(sub):
        m = [k for n, k in by.items() if sub in n]
        assert m, sub
        return m

    for k in find("k_frameILi"):
        assert k["vgpr"] <= 80, (k["name"], k["vgpr"])
    for k in find("k_sort_projectILi4E"):
        assert k["vgpr"] <= 128, (k["name"], k["vgpr"])
