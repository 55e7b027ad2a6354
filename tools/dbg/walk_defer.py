#!/usr/bin/env python3
"""C3 / C4 / C5 streams with and without the deferred walk launch: wall per call and how many walks went out without a stream wait."""
import sys, time, json, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from gem_amd import ElevationMap, synth, _lib
lib = _lib.load()
def run(name, wl, batched, reps):
    for dbg in ({}, {"defer_walk": 0}, {}, {"defer_walk": 0}):
        m = ElevationMap(wl.length, wl.resolution, debug=dbg)
        if getattr(wl, "map_position", None) is not None: m.move(wl.map_position)
        if batched:
            cat = torch.from_numpy(np.concatenate(wl.clouds)).cuda()
            off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
            pb = m.pack_batch(wl.frames, off, wl.var_updates)
            fn = lambda: m.add_batch(pb, cat)
        else:
            d = torch.from_numpy(wl.clouds[0]).cuda()
            fn = lambda: m.add(wl.frames[0], d)
        for _ in range(10): fn()
        m.synchronize()
        u0 = m.debug_get("walks_unwaited")
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        m.synchronize(); dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"config": name, "debug": dbg, "us_per_call": round(dt * 1e6, 2), "walks_unwaited": m.debug_get("walks_unwaited") - u0, "calls": reps}))
        m.close()
run("c3", synth.config_c3(), False, 300)
run("c4", synth.config_c4(n_sweeps=32), True, 100)
run("c5", synth.config_c5(), True, 40)
