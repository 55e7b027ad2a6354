#!/usr/bin/env python3
"""Kernel metadata of the gfx950 code objects inside libgem_hip.so (no ROCm tool needed: ELF + msgpack).

The host library carries one clang offload bundle per translation unit in its .hip_fatbin section; every bundle holds an AMDGPU ELF whose
NT_AMDGPU_METADATA note (msgpack) lists, per kernel: VGPR / SGPR counts, spill counts, scratch (.private_segment_fixed_size), LDS.

    python tools/code_objects.py [path/to/libgem_hip.so] [--spills]       # table of every kernel (or only those that spill / use scratch)
"""
import struct
import sys
from pathlib import Path

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_sections(blob):
    assert blob[:4] == b"\x7fELF" and blob[4] == 2 and blob[5] == 1, "not a little-endian ELF64"
    shoff, = struct.unpack_from("<Q", blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", blob, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", blob, shoff + i * shentsize)
        secs.append((name, typ, off, size))
    stroff = secs[shstrndx][2]

    def nm(o):
        e = blob.index(b"\0", stroff + o)
        return blob[stroff + o:e].decode()
    return [(nm(n), t, o, s) for n, t, o, s in secs]


def fatbin(lib_path):
    blob = Path(lib_path).read_bytes()
    for name, _t, off, size in _elf_sections(blob):
        if name == ".hip_fatbin":
            return blob[off:off + size]
    raise RuntimeError(f"{lib_path}: no .hip_fatbin section")


def code_objects(lib_path, arch="gfx950"):
    """[(triple, ELF bytes)] of every device code object for `arch` in the library."""
    fb = fatbin(lib_path)
    out, at = [], 0
    while True:
        at = fb.find(MAGIC, at)
        if at < 0:
            break
        n, = struct.unpack_from("<Q", fb, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", fb, p)
            triple = fb[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if triple.startswith("hip") and arch in triple and size:
                out.append((triple, fb[at + off:at + off + size]))
        at += len(MAGIC)
    return out


def kernels_of(elf):
    import msgpack
    for name, typ, off, size in _elf_sections(elf):
        if typ != 7:                      # SHT_NOTE
            continue
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            nname = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if nname == b"AMDGPU" and ntype == 32:      # NT_AMDGPU_METADATA
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    yield k


def all_kernels(lib_path):
    out = []
    for triple, elf in code_objects(lib_path):
        for k in kernels_of(elf):
            out.append({
                "name": k[".name"], "vgpr": k.get(".vgpr_count", -1), "agpr": k.get(".agpr_count", 0), "sgpr": k.get(".sgpr_count", -1),
                "vgpr_spill": k.get(".vgpr_spill_count", 0), "sgpr_spill": k.get(".sgpr_spill_count", 0),
                "scratch": k.get(".private_segment_fixed_size", 0), "lds": k.get(".group_segment_fixed_size", 0),
                "wg_max": k.get(".max_flat_workgroup_size", 0), "triple": triple,
            })
    return out


def demangle(names):
    import subprocess
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return r.stdout.split("\n")[:len(names)]
    except Exception:
        return list(names)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = args[0] if args else str(Path(__file__).resolve().parent.parent / "gem_amd" / "lib" / "libgem_hip.so")
    ks = all_kernels(lib)
    only = "--spills" in sys.argv
    pretty = demangle([k["name"] for k in ks])
    print(f"# {lib}: {len(ks)} kernels in {len(code_objects(lib))} gfx950 code objects")
    for k, nm in sorted(zip(ks, pretty), key=lambda t: t[1]):
        if only and not (k["vgpr_spill"] or k["sgpr_spill"] or k["scratch"]):
            continue
        print(f"{nm[:110]:110s} vgpr {k['vgpr']:3d} agpr {k['agpr']:3d} sgpr {k['sgpr']:3d} spill v{k['vgpr_spill']:3d} s{k['sgpr_spill']:3d} scratch {k['scratch']:4d} lds {k['lds']:6d}")


if __name__ == "__main__":
    main()
