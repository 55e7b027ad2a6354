"""The node's frame with caller-owned host arrays for several download piece counts and copy-thread counts, same box.
usage: python tools/dbg/download_groups.py"""
import sys
sys.path.insert(0, ".")
import torch
import bench
from gem_amd import ElevationMap

dev = torch.device("cuda:0")
for rep in range(2):
    for t in (4, 8):
        for g in (8, 4, 12, 14):
            r = bench.node_host_arrays(ElevationMap, dev, reps=60, copy_threads=t, debug={"download_groups": g})
            print(f"copy_threads {t} groups {g:2d}: {r['us_per_frame']:7.1f} us/frame  " + "  ".join(f"{k} {v:6.1f}" for k, v in r["us_per_call"].items()) +
                  "  ||  " + "  ".join(f"{k} {v:5.1f}" for k, v in r["host_us_per_frame_in_transfers"].items()), flush=True)
