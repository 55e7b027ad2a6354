#!/usr/bin/env python3
"""Static look at one kernel's gfx950 ISA: every loop (backward branch) with its body's instruction mix, and the kernel's totals.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -Iinclude gem_amd/csrc/gem_sort.hip -o /tmp/gem_sort.s
    python tools/isa_loops.py /tmp/gem_sort.s 'k_fuse_blockILi0ELi1ELi2048ELb0'
"""
import re
import sys


def kind(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")) - 1
    body = lines[start:end + 1]
    ins, labels = [], {}
    for l in body:
        s = l.strip()
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not s or s.startswith((";", ".", "_Z")):
            continue
        op = s.split()[0]
        ins.append((op, s))
    tot = {}
    for op, _ in ins:
        tot[kind(op)] = tot.get(kind(op), 0) + 1
    print(f"{pat}: {len(ins)} instructions  {tot}")
    loops = []
    for i, (op, s) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, tgt))
    for a, b, tgt in sorted(loops):
        mix = {}
        for op, _ in ins[a:b + 1]:
            mix[kind(op)] = mix.get(kind(op), 0) + 1
        print(f"  loop {tgt:14s} [{a:5d}..{b:5d}] {b - a + 1:5d} instr  {mix}")
    if "--dump" in sys.argv:
        inv = {}
        for k, v in labels.items():
            inv.setdefault(v, []).append(k)
        for i, (op, s) in enumerate(ins):
            for k in inv.get(i, []):
                print(f"{k}:")
            print(f"{i:6d}  {s}")


if __name__ == "__main__":
    main()
