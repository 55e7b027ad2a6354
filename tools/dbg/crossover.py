#!/usr/bin/env python3
"""Tile pipeline vs sorted pipeline by batch size (where does sort_min_points belong?)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from gem_amd import ElevationMap, synth

def bench(m, fn, reps=60):
    for _ in range(8): fn()
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    m.synchronize(); return (time.perf_counter() - t0) / reps * 1e6

wl = synth.config_c4(n_sweeps=16)
for ns in (1, 2, 3, 4, 6, 8, 16):
    cat = torch.from_numpy(np.concatenate(wl.clouds[:ns])).cuda()
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds[:ns]])])
    row = [f"{ns:2d} sweeps {cat.shape[0]:8d} pts"]
    for name, dbg in (("tile", {"sort_path": 0}), ("sorted", {"sort_min_points": 1})):
        for vu in (True, False):
            m = ElevationMap(wl.length, wl.resolution, debug=dbg)
            pb = m.pack_batch(wl.frames[:ns], off, wl.var_updates[:ns] if vu else None)
            row.append(f"{name}{'+vu' if vu else '   '} {bench(m, lambda: m.add_batch(pb, cat)):7.1f} us")
            m.close()
    print("  ".join(row), flush=True)
# a dense depth image cut to different sizes
wl3 = synth.config_c3()
for n in (40_000, 80_000, 150_000, 307_200):
    d = torch.from_numpy(wl3.clouds[0][:n]).cuda()
    row = [f"depth image {n:7d} pts"]
    for name, dbg in (("tile", {"sort_path": 0}), ("sorted", {"sort_min_points": 1})):
        m = ElevationMap(wl3.length, wl3.resolution, debug=dbg); m.move(wl3.map_position)
        row.append(f"{name} {bench(m, lambda: m.add(wl3.frames[0], d), 30):7.1f} us")
        m.close()
    print("  ".join(row), flush=True)
