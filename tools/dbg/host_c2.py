#!/usr/bin/env python3
"""C2 stream: host time to enqueue one gem_add_device (no synchronisation inside the loop) against the wall time per step,
through the Python binding and through ctypes directly (the binding's own overhead)."""
import sys, time, json, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from gem_amd import ElevationMap, synth, _lib

wl = synth.config_c4(n_sweeps=8, seed0=100)
d = [torch.from_numpy(c).cuda() for c in wl.clouds]
m = ElevationMap(wl.length, wl.resolution)
lib = _lib.load()
for k in range(40): m.add(wl.frames[k % 8], d[k % 8])
m.synchronize()
for reps in (20, 200, 2000):
    t0 = time.perf_counter()
    for k in range(reps): m.add(wl.frames[k % 8], d[k % 8])
    t1 = time.perf_counter(); m.synchronize(); t2 = time.perf_counter()
    print(json.dumps({"path": "ElevationMap.add", "reps": reps, "host_enqueue_us_per_call": (t1 - t0) / reps * 1e6, "wall_us_per_call": (t2 - t0) / reps * 1e6}))
# ctypes directly: the frame structs packed once
fp = [f.to_struct() for f in wl.frames]
if fp[0] is not None:
    ptrs = [C.c_void_p(x.data_ptr()) for x in d]
    n = wl.clouds[0].shape[0]
    for reps in (200, 2000):
        t0 = time.perf_counter()
        for k in range(reps): lib.gem_add_device(m._h, C.byref(fp[k % 8]), n, ptrs[k % 8], None, None)
        t1 = time.perf_counter(); m.synchronize(); t2 = time.perf_counter()
        print(json.dumps({"path": "ctypes gem_add_device", "reps": reps, "host_enqueue_us_per_call": (t1 - t0) / reps * 1e6, "wall_us_per_call": (t2 - t0) / reps * 1e6}))
