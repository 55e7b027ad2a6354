#!/usr/bin/env python3
"""Profiling aid: per-phase cycle stamps of k_fuse (thread 0 of every tile) on the C2 workload."""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from gem_amd import ElevationMap, synth, _lib

wl = synth.config_c4(n_sweeps=4)
m = ElevationMap(wl.length, wl.resolution)
lib = _lib.load()
d = [torch.from_numpy(c).cuda() for c in wl.clouds]
for k in range(3):
    m.add(wl.frames[k], d[k])
lib.gem_debug_fuse_stamps(m._h, 1, None, 0)
if len(sys.argv) > 1 and sys.argv[1] == "c3":          # dense depth image: generic (linked-list) path, many batches per tile
    wl3 = synth.config_c3()
    m = ElevationMap(wl3.length, wl3.resolution)
    m.move(wl3.map_position)
    lib.gem_debug_fuse_stamps(m._h, 1, None, 0)
    m.add(wl3.frames[0], torch.from_numpy(wl3.clouds[0]).cuda())
elif len(sys.argv) > 1 and sys.argv[1] == "batch":     # first sweeps of a batched call (16 stamps per tile = ~3 sweeps)
    import numpy as _np
    cat = torch.from_numpy(_np.concatenate(wl.clouds)).cuda()
    off = _np.concatenate([[0], _np.cumsum([c.shape[0] for c in wl.clouds])])
    m.add_batch(wl.frames, cat, off, wl.var_updates)
else:
    m.add(wl.frames[3], d[3])
T = 4096
buf = np.zeros((T, 16), np.uint64)
n = lib.gem_debug_fuse_stamps(m._h, 0, buf.ctypes.data_as(C.c_void_p), T)
if len(sys.argv) > 1 and sys.argv[1] == "c3":          # stamps 8..13 of a tile: the dense path of sweep `dbg_sweep` (gem_debug_set)
    d = buf[:n, 8:14].astype(np.int64)
    dn = d[d[:, 0] > 0]
    dur = dn[:, 5] - dn[:, 0]
    print("dense tiles", len(dn), "of", n)
    for i in np.argsort(-dur)[:6]:
        print("dense tile: total", dur[i], "count/cursors/place/walk(thread 0)/walk(all)", list(np.diff(dn[i])))
    buf[:, 8:] = 0
st = buf[:n].astype(np.int64)
names = ["start", "tile_ld", "heads", "filled+rec_issue", "rec+count", "cellscan", "placed", "walked", "stores", "x"]
t0 = st[:, 0].min()
nb = (st > 0).sum(1)
tot = np.array([st[i, nb[i] - 1] - st[i, 0] for i in range(n)])
print("tiles", n, "stamps/tile hist", np.bincount(nb))
print("kernel span (cycles): first start -> last end:", st.max() - t0)
order = np.argsort(-tot)
for label, idx in (("slowest", order[:3]), ("median", order[len(order) // 2: len(order) // 2 + 2])):
    for i in idx:
        s = st[i, :nb[i]]
        print(label, "tile", i, "stamps", nb[i], "total", tot[i], "start@", s[0] - t0, "deltas", list(np.diff(s)))
full = nb == 99
if full.any():
    dd = np.diff(st[full, :10], axis=1)
    print("mean deltas over single-batch tiles:", dict(zip(names[1:], dd.mean(0).astype(int))))
    print("max  deltas over single-batch tiles:", dict(zip(names[1:], dd.max(0).astype(int))))
