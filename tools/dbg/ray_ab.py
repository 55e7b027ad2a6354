#!/usr/bin/env python3
"""Same-box A/B of k_raytracing's loads-in-flight depth on the C2 node sequence (add with lowest tracking + Map_feature + Raytracing)."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from gem_amd import ElevationMap, synth

wl = synth.config_c2()
d = [torch.from_numpy(c).to("cuda:0") for c in (synth.config_c2(seed=2 + k).clouds[0] for k in range(8))]
frames = [synth.config_c2(seed=2 + k).frames[0] for k in range(8)]
m = ElevationMap(wl.length, wl.resolution)
m.set_lowest_tracking(True)
k = [0]
def g():
    m.add(frames[k[0] % 8], d[k[0] % 8]); k[0] += 1
    m.map_feature(fetch=False); m.raytracing()
def f():
    m.add(frames[k[0] % 8], d[k[0] % 8]); k[0] += 1
    m.map_feature(fetch=False)
def t(fn, reps=400):
    for _ in range(20): fn()
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    m.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
import numpy as np
f(); m.synchronize()
tr, el = m.layer("traver"), m.layer("elevation")
walk = (tr < 0.7) & (el != -10)
print(json.dumps({"walkers": int(walk.sum()), "non_empty": int((el != -10).sum())}), flush=True)
for rnd in range(3):
    row = {"without_raytracing": round(t(f), 2)}
    for lanes in (1, 4, 8, 16):
        for depth in (4, 8):
            m.debug_set("ray_depth", depth); m.debug_set("ray_lanes", lanes)
            row[f"lanes{lanes}_depth{depth}"] = round(t(g), 2)
    print(json.dumps(row), flush=True)
