#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and / or PMC counter collection) of one or more runs.

    tools/rocprof_summary.py --trace DIR [--pmc DIR ...] [--json OUT.json] [--note "..."] > profiles/rNN_xxx.txt

Per kernel: calls, total / avg / min / max duration (us), launch geometry; per (kernel, counter): average per
dispatch.  FETCH_SIZE / WRITE_SIZE are in KB; the HBM-traffic estimate applies the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts 16-B/lane coalesced reads at half their bytes; our
own calibration in profiles/ shows 4-B/lane reads are counted in full, so the figure is bracketed).
"""
import argparse
import collections
import csv
import glob
import json
import os


def find(d, suffix):
    f = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return f[0] if f else None


def short(name):
    return name.replace("void ", "").split("(")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--pmc", action="append", default=[])
    ap.add_argument("--json")
    ap.add_argument("--note", action="append", default=[])
    a = ap.parse_args()
    out = {"kernels": {}, "counters": {}}
    if a.trace:
        f = find(a.trace, "kernel_trace.csv")
        rows = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            rows[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            meta[k] = (r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
        tot = sum(sum(v) for v in rows.values()) or 1.0
        print(f"# rocprofv3 --kernel-trace summary ({f})")
        print(f"{'kernel':52s} {'calls':>6s} {'total_us':>10s} {'avg_us':>8s} {'min_us':>8s} {'max_us':>8s} {'%':>6s} {'grid':>8s} {'wg':>5s} {'lds':>7s} {'vgpr':>5s} {'sgpr':>5s} {'scratch':>7s}")
        for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
            g, wg, lds, vg, sg, sc = meta[k]
            print(f"{k[:52]:52s} {len(v):6d} {sum(v):10.1f} {sum(v)/len(v):8.2f} {min(v):8.2f} {max(v):8.2f} {100*sum(v)/tot:6.1f} {g:>8s} {wg:>5s} {lds:>7s} {vg:>5s} {sg:>5s} {sc:>7s}")
            out["kernels"][k] = {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)}
    for d in a.pmc:
        f = find(d, "counter_collection.csv")
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        per_dispatch = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"\n# rocprofv3 --pmc summary ({f}): average per dispatch")
        for k in acc:
            for c, v in sorted(acc[k].items()):
                print(f"{k[:52]:52s} {c:24s} n={len(v):5d} avg={sum(v)/len(v):14.2f}")
                out["counters"].setdefault(k, {})[c] = sum(v) / len(v)
    for k, c in out["counters"].items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            lo = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            hi = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            c["hbm_bytes_low"] = lo; c["hbm_bytes_high"] = hi
            print(f"# {k}: HBM traffic per dispatch between {lo/1e6:.2f} MB (counters as read) and {hi/1e6:.2f} MB (FETCH_SIZE doubled, guide's gfx950 rule for 16-B/lane reads)")
    for n in a.note:
        print("# " + n)
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
