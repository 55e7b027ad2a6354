"""The node's frame with caller-owned host arrays (bench.node_host_arrays) for several copy-thread counts, same box.
usage: python tools/dbg/host_path.py [threads ...]"""
import sys, json
sys.path.insert(0, ".")
import torch
import bench
from gem_amd import ElevationMap

dev = torch.device("cuda:0")
for t in [int(a) for a in sys.argv[1:]] or [0, 2, 4, 6, 8, 0, 4]:
    r = bench.node_host_arrays(ElevationMap, dev, reps=60, copy_threads=t)
    print(f"copy_threads {t}: {r['us_per_frame']:7.1f} us/frame  " + "  ".join(f"{k} {v:6.1f}" for k, v in r["us_per_call"].items()) +
          "  ||  " + "  ".join(f"{k} {v:5.1f}" for k, v in r["host_us_per_frame_in_transfers"].items()), flush=True)
