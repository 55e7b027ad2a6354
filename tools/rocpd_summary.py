#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / avg / min / max (us),
plus PMC counter sums when present.  Usage: tools/rocpd_summary.py results.db [> profiles/xxx.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(sgpr_count) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 kernel-trace summary of {path}")
    print(f"{'kernel':70s} {'calls':>7s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'%':>6s} {'grid':>8s} {'wg':>5s} {'lds':>7s} {'vgpr':>5s} {'sgpr':>5s}")
    for n, c, s, a, mn, mx, g, wg, lds, vg, sg in rows:
        print(f"{n[:70]:70s} {c:7d} {s/1e3:11.1f} {a/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/tot:6.1f} {g:8d} {wg:5d} {lds:7d} {vg:5d} {sg:5d}")
    try:
        pm = cur.execute("select name, counter_name, count(*), sum(counter_value), avg(counter_value) from pmc_events "
                         "group by name, counter_name order by sum(counter_value) desc").fetchall()
        if pm:
            print("\n# PMC counters (per kernel: counter, dispatches, sum, avg per dispatch)")
            for n, cn, c, s, a in pm:
                print(f"{n[:60]:60s} {cn:24s} {c:7d} {s:16.1f} {a:14.2f}")
    except sqlite3.Error as e:
        print(f"# (no PMC data: {e})")


if __name__ == "__main__":
    main(sys.argv[1])
