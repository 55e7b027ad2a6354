"""What the two waits of bench.py's barrier cost on an IDLE device, and a 20-sweep region split into its parts (host enqueue, wait)."""
import sys, time, pathlib
import numpy as np, torch
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from gem_amd import ElevationMap
from gem_amd import synth

wl = synth.config_c4(n_sweeps=8, seed0=100)
dev = torch.device("cuda:0")
emap = ElevationMap(wl.length, wl.resolution, device=0)
clouds = [torch.from_numpy(c).to(dev) for c in wl.clouds]
for i in range(40):
    emap.add(wl.frames[i % 8], clouds[i % 8])
emap.synchronize(); torch.cuda.synchronize()

def med(f, n=200):
    v = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); v.append(1e6 * (time.perf_counter() - t0))
    v.sort(); return v[len(v) // 2], v[0], v[-1]
print("idle emap.synchronize():      %.2f us (min %.2f max %.2f)" % med(emap.synchronize))
print("idle torch.cuda.synchronize(): %.2f us (min %.2f max %.2f)" % med(torch.cuda.synchronize))
rows = []
for rep in range(30):
    emap.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        emap.add(wl.frames[i % 8], clouds[i % 8])
    t1 = time.perf_counter()
    emap.synchronize()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    rows.append((1e6 * (t1 - t0), 1e6 * (t2 - t1), 1e6 * (t3 - t2), 1e6 * (t3 - t0)))
rows = np.array(rows[5:])
print("20 sweeps: enqueue %.1f us, emap.synchronize %.1f us, torch.cuda.synchronize %.1f us, total %.1f us (medians; total min %.1f)" %
      (*np.median(rows, axis=0), rows[:, 3].min()))
