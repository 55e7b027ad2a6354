"""A/B of the block walk's knobs on C4 (and C5 with --c5): wall per batch of a stream, k_fuse_block alone.
    python tools/dbg/c4_ab.py [--c5] key=v,key=v  key=v ..."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from gem_amd import ElevationMap, synth
c5 = "--c5" in sys.argv
variants = [a for a in sys.argv[1:] if not a.startswith("--")] or [""]
wl = synth.config_c5() if c5 else synth.config_c4(n_sweeps=32)
cat = torch.from_numpy(np.concatenate(wl.clouds)).cuda()
off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
upd = None if c5 else wl.var_updates
def run(tag, dbg, reps=30, warm=6):
    m = ElevationMap(wl.length, wl.resolution, device=0, debug=dbg)
    pb = m.pack_batch(wl.frames, off, upd)
    for _ in range(warm): m.add_batch(pb, cat)
    m.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): m.add_batch(pb, cat)
    m.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e6
    m.close()
    m = ElevationMap(wl.length, wl.resolution, device=0, debug=dict(dbg, overlap=0))
    for _ in range(3): m.add_batch(pb, cat)
    m.set_timing(True); m.stats(reset=True)
    for _ in range(10): m.add_batch(pb, cat)
    st = m.stats(); m.close()
    print(f"{tag or 'default':40s} us/batch {wall:7.1f}   walk alone {1e3 * st['ms_walk'] / st['launches_walk']:6.1f}   sort alone {[round(1e3 * v / st['launches_sort'], 1) for v in st['ms_sort'] if v > 0]}", flush=True)
for rep in range(2):
    for v in variants:
        run(v, {k: int(x) for k, x in (kv.split("=") for kv in v.split(",") if kv)})
