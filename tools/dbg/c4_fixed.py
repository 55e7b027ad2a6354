"""What a C4 walk costs when it has (almost) nothing to walk: the batch with every point but one per sweep thrown out of the map (x 100),
with and without variance increments -- the fixed part of k_fuse_block<.., 2048> over 1444 blocks.  usage: python tools/dbg/c4_fixed.py"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gem_amd import ElevationMap, synth

wl = synth.config_c4()
for name, scale, vu in (("full batch", 1.0, True), ("full batch, no increments", 1.0, False), ("no records", 100.0, True), ("no records, no increments", 100.0, False)):
    clouds = []
    for c in wl.clouds:
        c = c.copy(); c[1:, :3] *= scale; clouds.append(c)
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])])
    cat = torch.from_numpy(np.concatenate(clouds)).cuda()
    m = ElevationMap(wl.length, wl.resolution, debug={"overlap": 0})
    pb = m.pack_batch(wl.frames, off, wl.var_updates if vu else None)
    for _ in range(6):
        m.add_batch(pb, cat)
    m.synchronize()
    m.set_timing(True); m.stats(reset=True)
    for _ in range(10):
        m.add_batch(pb, cat)
    st = m.stats()
    k = max(st["launches_walk"], 1)
    print(f"{name:28s}: walk {1e3 * st['ms_walk'] / k:6.1f} us   sort kernels " + " ".join(f"{1e3 * v / k:5.1f}" for v in st["ms_sort"]), flush=True)
    m.close()
