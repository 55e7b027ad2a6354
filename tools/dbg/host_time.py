#!/usr/bin/env python3
"""How long does the HOST take to enqueue one call (no synchronisation inside the loop) against the device period?"""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from gem_amd import ElevationMap, synth

def run(name, wl, with_vu, reps):
    cat = torch.from_numpy(np.concatenate(wl.clouds)).cuda()
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    m = ElevationMap(wl.length, wl.resolution)
    pb = m.pack_batch(wl.frames, off, wl.var_updates if with_vu else None)
    for _ in range(5): m.add_batch(pb, cat)
    m.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): m.add_batch(pb, cat)
    t1 = time.perf_counter()
    m.synchronize()
    t2 = time.perf_counter()
    print(json.dumps({"config": name, "host_enqueue_us_per_call": (t1 - t0) / reps * 1e6, "wall_us_per_call": (t2 - t0) / reps * 1e6}), flush=True)
    m.close()

run("c4", synth.config_c4(n_sweeps=32), True, 200)
run("c5", synth.config_c5(), False, 40)
wl = synth.config_c2()
d = torch.from_numpy(wl.clouds[0]).cuda()
m = ElevationMap(wl.length, wl.resolution)
for _ in range(20): m.add(wl.frames[0], d)
m.synchronize(); t0 = time.perf_counter()
for _ in range(2000): m.add(wl.frames[0], d)
t1 = time.perf_counter(); m.synchronize(); t2 = time.perf_counter()
print(json.dumps({"config": "c2", "host_enqueue_us_per_call": (t1 - t0) / 2000 * 1e6, "wall_us_per_call": (t2 - t0) / 2000 * 1e6}))

def first_calls(name, wl, with_vu):
    cat = torch.from_numpy(np.concatenate(wl.clouds)).cuda()
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    m = ElevationMap(wl.length, wl.resolution)
    pb = m.pack_batch(wl.frames, off, wl.var_updates if with_vu else None)
    for _ in range(6): m.add_batch(pb, cat)
    out = []
    for rep in range(3):
        m.synchronize()
        ts = [time.perf_counter()]
        for _ in range(6):
            m.add_batch(pb, cat); ts.append(time.perf_counter())
        out.append([round((b - a) * 1e6, 1) for a, b in zip(ts, ts[1:])])
    print(json.dumps({"config": name, "host_us_of_each_of_six_calls_after_a_sync": out}), flush=True)
    m.close()

first_calls("c4", synth.config_c4(n_sweeps=32), True)
first_calls("c5", synth.config_c5(), False)
