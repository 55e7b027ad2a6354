#!/usr/bin/env python3
"""Profiling aid: the time line of ONE k_frame launch of the C2 stream -- cycle stamps of thread 0 of every tile workgroup (fuse of the
previous sweep) and of every binning block (this sweep), put on a common clock per XCD (HW_REG_XCC_ID; the XCDs'
counters are not synchronised, so every XCD's earliest stamp is its zero).

    python tools/frame_phases.py [sweeps]
"""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from gem_amd import ElevationMap, synth, _lib

n_sw = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = synth.config_c4(n_sweeps=n_sw)
m = ElevationMap(wl.length, wl.resolution)
lib = _lib.load()
d = [torch.from_numpy(c).cuda() for c in wl.clouds]
for k in range(n_sw - 2):
    m.add(wl.frames[k], d[k])
m.synchronize()
assert lib.gem_debug_set(m._h, b"dbg_frame", 1) == 0
lib.gem_debug_fuse_stamps(m._h, 1, None, 0)
m.add(wl.frames[n_sw - 2], d[n_sw - 2])          # k_bin_wave alone (the synchronisation above flushed the deferred fuse)
m.add(wl.frames[n_sw - 1], d[n_sw - 1])          # k_frame: fuse of sweep n-2 + binning of sweep n-1  <- the launch whose stamps are read
ROWS = 8192
buf = np.zeros((ROWS, 16), np.uint64)
n = lib.gem_debug_fuse_stamps(m._h, 0, buf.ctypes.data_as(C.c_void_p), ROWS)
st = buf[:n].astype(np.int64)
T = ((wl.length + 15) // 16) ** 2
tiles, bins = st[:T], st[T:]
tiles = tiles[tiles[:, 15] > 0]; bins = bins[bins[:, 15] > 0]
blk_t = tiles[:, 15] - 1; blk_b = bins[:, 15] - 1
nst = (tiles[:, :12] > 0).sum(1)
# time line on s_memrealtime (100 MHz, one clock for the whole chip): 10 ns ticks
rt0 = min(tiles[:, 12].min(), bins[:, 12].min() if len(bins) else tiles[:, 12].min())
ts, te = (tiles[:, 12] - rt0) / 100.0, (tiles[:, 13] - rt0) / 100.0             # us
bs, be = (bins[:, 12] - rt0) / 100.0, (bins[:, 13] - rt0) / 100.0
xt = tiles[:, 14] - 1
print(f"k_frame: {len(tiles)} tile workgroups with stamps, {len(bins)} binning blocks; stamps per tile hist {np.bincount(nst).tolist()}")
print("workgroups per XCD (HW_REG_XCC_ID):", np.bincount(np.concatenate([xt, bins[:, 14] - 1]), minlength=8).tolist())
print(f"first start -> last tile end {te.max():.2f} us; last binning block end {be.max() if len(be) else 0:.2f} us")
step = 0.5
edges = np.arange(0, max(te.max(), be.max() if len(be) else 0) + step, step)
print("window (us)     tiles starting  tiles ending  bin blocks starting  bin blocks ending")
for a_, b_ in zip(edges[:-1], edges[1:]):
    print(f"  [{a_:4.1f},{b_:4.1f})   {((ts >= a_) & (ts < b_)).sum():8d}      {((te >= a_) & (te < b_)).sum():8d}      {((bs >= a_) & (bs < b_)).sum():8d}           {((be >= a_) & (be < b_)).sum():8d}")
dur = tiles[np.arange(len(tiles)), nst - 1] - tiles[:, 0]                        # cycles (the workgroup's own counter)
full = nst == 7

names = ["flags (round trip 1)", "descriptor words (round trip 2)", "scan + list", "records in LDS + ranks (round trip 3)", "chains", "stores"]
if full.any():
    dd = np.diff(tiles[full, :7], axis=1)
    print("tiles with records:", int(full.sum()), "| mean cycles per phase:", dict(zip(names, dd.mean(0).astype(int).tolist())))
    order = np.argsort(-dur)
    for i in order[:5]:
        print(f"  slow tile: block {blk_t[i]:5d} start {ts[i]:6.2f} us end {te[i]:6.2f} us total {dur[i]:6d} cycles, phases {np.diff(tiles[i, :nst[i]]).tolist()}")
    late = np.argsort(-te)[:5]
    for i in late:
        print(f"  last to end: block {blk_t[i]:5d} start {ts[i]:6.2f} us end {te[i]:6.2f} us total {dur[i]:6d} cycles ({dur[i] / max(1e-9, (te[i] - ts[i])) / 1000:.2f} GHz)")
print("tiles that left early (no record):", int((nst < 7).sum()), "mean life", int(dur[nst < 7].mean()) if (nst < 7).any() else 0)
if len(bins):
    print(f"binning blocks: mean life {(be - bs).mean():.2f} us, max {(be - bs).max():.2f} | first start {bs.min():.2f} us, last start {bs.max():.2f} us")
