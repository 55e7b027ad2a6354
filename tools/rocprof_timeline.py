#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace run of tools/bench_configs.py: for every configuration (split at the long
idle gaps between them), three passes of its timed loop: kernel, stream / queue, start relative to the pass's first kernel, duration.
Shows what overlaps with what when the product runs its two streams.

    tools/rocprof_timeline.py TRACE_DIR > profiles/rNN_xxx_timeline.txt
"""
import csv
import glob
import os
import sys


def short(name):
    return name.replace("void ", "").split("(")[0]


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), r["Grid_Size_X"], r["Workgroup_Size_X"])
            for r in csv.DictReader(open(f))]
    rows.sort()
    # sections: split where the device idles for more than 20 ms (a new configuration is being generated on the host)
    sections, cur = [], []
    for r in rows:
        if cur and r[0] - max(x[1] for x in cur[-50:]) > 20_000_000:
            sections.append(cur); cur = []
        cur.append(r)
    if cur:
        sections.append(cur)
    print(f"# {f}: {len(rows)} dispatches, {len(sections)} sections")
    for si, sec in enumerate(sections):
        walks = [i for i, r in enumerate(sec) if r[2].startswith("gem::k_fuse_walk") or r[2].startswith("gem::k_fuse_block") or r[2].startswith("gem::k_fuse_list") or r[2].startswith("gem::k_frame")]
        if len(walks) < 3:
            continue
        # three passes from the first QUARTER of the section (the steady timed loop; its second half are the instrumented passes,
        # which do not overlap): from the first kernel after a fuse kernel's start to the end of the third fuse kernel after it
        mid = max(len(walks) // 4, 3)
        a, b = walks[mid - 3] + 1, walks[mid]
        part = sec[a:b + 1]
        t0 = part[0][0]
        print(f"\n## section {si}: {len(sec)} dispatches; three passes of the timed loop (us relative to the first kernel shown)")
        print(f"{'start':>9s} {'dur':>8s} {'queue':>6s} {'grid':>9s} {'wg':>5s}  kernel")
        for s, e, k, q, g, w in part:
            print(f"{(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {q:>6s} {g:>9s} {w:>5s}  {k[:90]}")
        print(f"# span {(part[-1][1] - t0) / 1e3:.1f} us for three passes")


if __name__ == "__main__":
    main()
