#!/usr/bin/env python3
"""C3 stream: host time to enqueue one frame (no synchronisation inside the loop) against the wall time per frame."""
import sys, time, json, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from gem_amd import ElevationMap, synth, _lib
wl = synth.config_c3()
d = torch.from_numpy(wl.clouds[0]).cuda()
lib = _lib.load()
for dbg in ({}, {"sort_chunk": 4096}, {"ride_events": 0}):
    m = ElevationMap(wl.length, wl.resolution, debug=dbg)
    m.move(wl.map_position)
    fp = wl.frames[0].to_struct(); n = d.shape[0]; ptr = C.c_void_p(d.data_ptr())
    for _ in range(10): m.add(wl.frames[0], d)
    m.synchronize()
    for reps in (8, 200):
        t0 = time.perf_counter()
        for _ in range(reps): lib.gem_add_device(m._h, C.byref(fp), n, ptr, None, None)
        t1 = time.perf_counter(); m.synchronize(); t2 = time.perf_counter()
        print(json.dumps({"debug": dbg, "reps": reps, "host_enqueue_us_per_frame": (t1 - t0) / reps * 1e6, "wall_us_per_frame": (t2 - t0) / reps * 1e6}))
    m.close()
