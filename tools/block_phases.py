#!/usr/bin/env python3
"""Profiling aid: per-block cycle stamps of k_fuse_block (thread 0 of every block) on a batched configuration.

    python tools/block_phases.py [c4|c5] [--debug key=value,...]
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from gem_amd import ElevationMap, _lib, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "c4"
dbg = {"sort_form": 2}
for a in sys.argv[1:]:
    if a.startswith("--debug="):
        dbg.update({k: int(v) for k, v in (kv.split("=") for kv in a[8:].split(","))})
wl = synth.config_c4(n_sweeps=32) if which in ("c4", "c4nv") else synth.config_c5()
cat = torch.from_numpy(np.concatenate(wl.clouds)).cuda()
off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
m = ElevationMap(wl.length, wl.resolution, debug=dbg)
lib = _lib.load()
pb = m.pack_batch(wl.frames, off, wl.var_updates if which == "c4" else None)      # c4nv: the same sweeps without variance increments
for _ in range(3):
    m.add_batch(pb, cat)
m.synchronize()
lib.gem_debug_fuse_stamps(m._h, 1, None, 0)
m.add_batch(pb, cat)
m.synchronize()
rows = 4 * ((wl.length + 31) // 32) ** 2
buf = np.zeros((rows, 16), np.uint64)
n = lib.gem_debug_fuse_stamps(m._h, 0, buf.ctypes.data_as(C.c_void_p), rows)
st = buf[:n].astype(np.int64)
ran = st[:, 6] > 0
print(f"{which}: blocks {n}, ran to the end {int(ran.sum())}, debug {dbg}")
s = st[ran]
t0 = s[:, 0].min()
span = s[:, 6].max() - t0
print(f"kernel span (cycles, first start -> last end): {span}")
tot = s[:, 6] - s[:, 0]
names = ["setup", "rank+wait", "bases", "place", "chains(own)"]
vals = np.stack([s[:, 1] - s[:, 0], s[:, 2], s[:, 3], s[:, 4], s[:, 5]], 1)
print("sum over blocks (Mcycles):", {k: round(float(v) / 1e6, 2) for k, v in zip(names, vals.sum(0))}, "total", round(float(tot.sum()) / 1e6, 2))
print("records/block: mean %.0f max %d; batches: mean %.2f max %d; sum of longest chains (wave 0): mean %.1f max %d" %
      (s[:, 7].mean(), s[:, 7].max(), s[:, 8].mean(), s[:, 8].max(), s[:, 9].mean(), s[:, 9].max()))
print("blocks by records: " + ", ".join(f">= {t}: {int((s[:, 7] >= t).sum())}" for t in (2048, 4096, 6144, 8192, 12288)))
order = np.argsort(-tot)
for i in order[:8]:
    print(f"  ticks per chain step {vals[i][4] / max(s[i, 9], 1):.0f} rare steps {s[i, 10]} before-the-loops {s[i, 11]}", end=" ")
    print(f"slow block: total {tot[i]:7d} start@{s[i, 0] - t0:7d} end@{s[i, 6] - t0:7d} R {s[i, 7]:6d} batches {s[i, 8]:3d} chains {s[i, 9]:4d} | " +
          " ".join(f"{k} {v}" for k, v in zip(names, vals[i])))
late = np.argsort(-s[:, 6])[:5]
for i in late:
    print(f"last to end: end@{s[i, 6] - t0:7d} start@{s[i, 0] - t0:7d} total {tot[i]:7d} R {s[i, 7]:6d} batches {s[i, 8]:3d} chains {s[i, 9]:4d}")
h = np.histogram(s[:, 0] - t0, bins=8)
print("start-time histogram:", list(h[0]), [int(x) for x in h[1]])
# lane use of the chains: every wave runs as many steps as its busiest cell has records in the round; a step is 64 lane-steps
ws = s[:, 12:16]
steps = (ws >> 32).astype(np.int64); recs = (ws & 0xffffffff).astype(np.int64)
if steps.sum() > 0:
    print("chains, all blocks: %d wave-steps for %d records: lane use %.1f %%; per wave (0 = the busiest cells): " % (steps.sum(), recs.sum(), 100.0 * recs.sum() / (64.0 * steps.sum())) +
          ", ".join("w%d %d steps / %d records = %.0f %%" % (w, steps[:, w].sum(), recs[:, w].sum(), 100.0 * recs[:, w].sum() / max(64.0 * steps[:, w].sum(), 1)) for w in range(4)))
    for lo, hi in ((0, 512), (512, 2048), (2048, 8192), (8192, 1 << 30)):
        sel = (s[:, 7] >= lo) & (s[:, 7] < hi)
        if sel.any():
            print("  blocks of %d..%d records: %d blocks, %d wave-steps, lane use %.1f %%" % (lo, hi, int(sel.sum()), steps[sel].sum(), 100.0 * recs[sel].sum() / max(64.0 * steps[sel].sum(), 1)))
