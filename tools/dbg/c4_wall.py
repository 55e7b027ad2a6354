import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from gem_amd import ElevationMap, synth
dev = torch.device("cuda", 0)
wl = synth.config_c4(n_sweeps=32)
cat = torch.from_numpy(np.concatenate(wl.clouds)).to(dev)
off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
def run(tag, other=None, reps=20, warm=6):
    m = ElevationMap(wl.length, wl.resolution, device=0)
    pb = m.pack_batch(wl.frames, off, wl.var_updates)
    for _ in range(warm): m.add_batch(pb, cat)
    m.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): m.add_batch(pb, cat)
    t1 = time.perf_counter()
    m.synchronize()
    t2 = time.perf_counter()
    print(tag, "us/batch", (t2 - t0) / reps * 1e6, "host enqueue us/batch", (t1 - t0) / reps * 1e6, flush=True)
    m.close()
run("alone")
run("alone reps 10 warm 3", reps=10, warm=3)
run("alone reps 100", reps=100)
o = ElevationMap(600, 0.05, device=0)
d0 = torch.from_numpy(wl.clouds[0]).to(dev)
for k in range(50): o.add(wl.frames[0], d0)
o.synchronize()
run("with another handle alive (deferred frame pending)")
o.synchronize()
run("with another handle alive (flushed)")
