"""Would splitting C4's heaviest blocks shorten the batch?  A stand-in that needs no new kernel: the same batch with the records of every
block of 256 cells CAPPED at `cap` (points of over-full blocks dropped at random, evenly over the sweeps), so that no block is heavy;
the walk's time and the call's against the records that are left.  usage: python tools/dbg/c4_cap.py"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np, torch
import oracle
from gem_amd import ElevationMap, synth

wl = synth.config_c4()
L = wl.length
ref = oracle.OracleMap(L, wl.resolution)
idx = [ref.process_points(f, c[:, 0], c[:, 1], c[:, 2])["index"] for f, c in zip(wl.frames, wl.clouds)]
def block_of(i):
    r, c = i // L, i % L
    return ((r >> 5) * ((L + 31) // 32) + (c >> 5)) * 4 + ((r & 31) >> 3)
blk = [np.where(i >= 0, block_of(np.maximum(i, 0)), -1) for i in idx]
allb = np.concatenate(blk); counts = np.bincount(allb[allb >= 0])
rng = np.random.default_rng(0)
for cap in (0, 8192, 4096, 2048):
    if cap:
        keep_p = np.ones(counts.size); over = counts > cap; keep_p[over] = cap / counts[over]
        clouds = [c[(b < 0) | (rng.random(b.size) < keep_p[np.maximum(b, 0)])] for c, b in zip(wl.clouds, blk)]
    else:
        clouds = wl.clouds
    kept = sum(int(((b >= 0) & (rng.random(b.size) < 2)).sum()) for b in blk) if not cap else None
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])])
    cat = torch.from_numpy(np.concatenate(clouds)).cuda()
    for dbg in ({"overlap": 0}, {}):
        m = ElevationMap(L, wl.resolution, debug=dbg)
        pb = m.pack_batch(wl.frames, off, wl.var_updates)
        for _ in range(6):
            m.add_batch(pb, cat)
        m.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            m.add_batch(pb, cat)
        m.synchronize()
        wall = (time.perf_counter() - t0) / 30 * 1e6
        m.set_timing(True); m.stats(reset=True)
        for _ in range(6):
            m.add_batch(pb, cat)
        st = m.stats()
        print(f"cap {cap:5d} points {int(off[-1]):8d} {'alone  ' if dbg else 'overlap'}: call {wall:7.1f} us (untimed loop)  walk {1e3 * st['ms_walk'] / max(st['launches_walk'], 1):6.1f}  sort kernels {1e3 * sum(st['ms_sort']) / max(st['launches_walk'], 1):6.1f}", flush=True)
        m.close()
