"""C3 (depth image) per frame of a stream + k_fuse_walk alone, per debug-knob variant."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from gem_amd import ElevationMap, synth
variants = [a for a in sys.argv[1:] if not a.startswith("--")] or [""]
wl = synth.config_c3()
dc = torch.from_numpy(wl.clouds[0]).cuda()
def run(tag, dbg, reps=40):
    m = ElevationMap(wl.length, wl.resolution, debug=dbg)
    m.move(wl.map_position)
    for _ in range(5): m.add(wl.frames[0], dc)
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): m.add(wl.frames[0], dc)
    m.synchronize(); wall = (time.perf_counter() - t0) / reps * 1e6
    m.set_timing(True); m.stats(reset=True)
    for _ in range(reps): m.add(wl.frames[0], dc)
    st = m.stats(); m.close()
    print(f"{tag or 'default':32s} us/frame {wall:7.1f}  walk (timed loop) {1e3 * st['ms_walk'] / max(st['launches_walk'], 1):6.1f}  sort {[round(1e3 * v / max(st['launches_sort'], 1), 1) for v in st['ms_sort'] if v > 0]}", flush=True)
for rep in range(2):
    for v in variants:
        run(v, {k: int(x) for k, x in (kv.split("=") for kv in v.split(",") if kv)})
