"""Does the overlap of a handle's streams depend on what other handles did to the process's HW queues before?"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from gem_amd import ElevationMap, synth

dev = torch.device("cuda", 0)
wl5 = synth.config_c5(n_points=10_000_000)
cat5 = torch.from_numpy(np.concatenate(wl5.clouds)).to(dev)
off5 = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl5.clouds])])
wl4 = synth.config_c4(n_sweeps=32)
cat4 = torch.from_numpy(np.concatenate(wl4.clouds)).to(dev)
off4 = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl4.clouds])])


def run(wl, cat, off, vu, reps, tag, keep=None):
    m = ElevationMap(wl.length, wl.resolution)
    pb = m.pack_batch(wl.frames, off, vu)
    for _ in range(6):
        m.add_batch(pb, cat)
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        m.add_batch(pb, cat)
    m.synchronize()
    print(tag, round(1e6 * (time.perf_counter() - t0) / reps, 1), "us", flush=True)
    if keep is not None:
        keep.append(m)
    else:
        m.close()


run(wl5, cat5, off5, None, 30, "C5 first handle of the process:")
run(wl5, cat5, off5, None, 30, "C5 second handle (first closed):")
run(wl4, cat4, off4, wl4.var_updates, 60, "C4 third handle:")
run(wl5, cat5, off5, None, 30, "C5 after a closed C4 handle:")
run(wl5, cat5, off5, None, 30, "C5 again:")
keep = []
run(wl4, cat4, off4, wl4.var_updates, 60, "C4 (kept open):", keep)
run(wl5, cat5, off5, None, 30, "C5 next to an open C4 handle:")
keep[0].close()
print("--- eight C4 handles in a row (each closed before the next: the pooled streams change roles every time)")
for k in range(8):
    run(wl4, cat4, off4, wl4.var_updates, 100, f"C4 handle {k}:")
