"""Which of a handle's four streams (creation order 0..3 in a fresh process) should take which role?"""
import sys, time, itertools
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from gem_amd import ElevationMap, synth

dev = torch.device("cuda", 0)
c5 = len(sys.argv) > 1 and sys.argv[1] == "c5"
wl4 = synth.config_c5(n_points=10_000_000) if c5 else synth.config_c4(n_sweeps=32)
cat4 = torch.from_numpy(np.concatenate(wl4.clouds)).to(dev)
off4 = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl4.clouds])])
m = ElevationMap(wl4.length, wl4.resolution)
pb = m.pack_batch(wl4.frames, off4, None if c5 else wl4.var_updates)
for _ in range(6):
    m.add_batch(pb, cat4)
m.synchronize()
ident = [0, 1, 2, 3]           # roles (own, bin, bin2, tab) -> creation index
for perm in itertools.permutations(range(4)):
    # gem_debug_set permutes relative to the CURRENT assignment: compose
    rel = [ident.index(p) for p in perm]
    m.debug_set("stream_roles", int("".join(map(str, rel))))
    ident = list(perm)
    for _ in range(6):
        m.add_batch(pb, cat4)
    m.synchronize(); t0 = time.perf_counter()
    reps = 40 if c5 else 100
    for _ in range(reps):
        m.add_batch(pb, cat4)
    m.synchronize()
    print("own,bin,bin2,tab =", perm, round(1e6 * (time.perf_counter() - t0) / reps, 1), "us", flush=True)
