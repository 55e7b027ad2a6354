"""Reproduces a fuzz scenario whose process_points + fuse (host arrays) step differs from the oracle, outside the scenario:
   python tools/dbg/fuse_repro.py <seed>"""
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import numpy as np
import fuzz_parity as fp
import oracle
from gem_amd import ElevationMap

seed = int(sys.argv[1])
cap = {}
orig_fuse = ElevationMap.fuse
def fuse(self, index, height, var, *rest, **kw):
    cap["last"] = (np.array(index), np.array(height), np.array(var), self.length, self.debug_knobs if hasattr(self, "debug_knobs") else None)
    return orig_fuse(self, index, height, var, *rest, **kw)
ElevationMap.fuse = fuse
try:
    print(fp.scenario(seed)); print("scenario passed"); sys.exit(0)
except AssertionError as e:
    print("scenario failed:", str(e)[:200])
idx, h, v, L, _ = cap["last"]
n = idx.size
print("captured fuse: n", n, "L", L, "valid", int(((idx >= 0) & (idx < L * L) & (h != -1)).sum()))
def trial(tag, knobs, idx, h, v):
    g = ElevationMap(L, 0.05, debug=knobs); o = oracle.OracleMap(L, 0.05)
    g.fuse(idx, h, v); o.fuse(idx, h, v)
    ge, oe = g.layer("elevation"), o.layer("elevation")
    bad = np.flatnonzero(ge.ravel() != oe.ravel())
    print(f"{tag:60s} differing cells {bad.size}", bad[:8], flush=True)
    g.close()
    return bad
bad = trial("fresh map, dense_min=0", {"dense_min": 0}, idx, h, v)
trial("fresh map, default knobs", {}, idx, h, v)
trial("fresh map, dense_min=0, sort off", {"dense_min": 0, "sort_path": 0}, idx, h, v) if False else None
if bad.size:
    rows, cols = bad // L, bad % L
    tiles = set(zip((rows >> 4).tolist(), (cols >> 4).tolist()))
    print("tiles of the differing cells (16x16):", sorted(tiles))
    for (tr, tc) in sorted(tiles)[:3]:
        r, c = idx // L, idx % L
        in_tile = (idx >= 0) & (idx < L * L) & ((r >> 4) == tr) & ((c >> 4) == tc)
        pos = np.flatnonzero(in_tile)
        units = np.unique(pos // 64)
        print(f"tile ({tr},{tc}): {pos.size} points (h == -1: {int((h[pos] == -1).sum())}), units {units.size}, unit range {units.min()}..{units.max()}, "
              f"points of the differing cells: {int(np.isin(idx[pos], bad).sum())}, their units {np.unique(pos[np.isin(idx[pos], bad)] // 64)[:10]}")
        only = np.where(in_tile, idx, -1).astype(np.int32)
        trial(f"only tile ({tr},{tc}), dense_min=0", {"dense_min": 0}, only, h, v)
        trial(f"only tile ({tr},{tc}), default", {}, only, h, v)
        # the tile's points alone, compacted to the front
        k = pos.size
        trial(f"tile ({tr},{tc}) compacted, dense_min=0", {"dense_min": 0}, idx[pos].copy(), h[pos].copy(), v[pos].copy())
    # prefixes of the cloud
    for frac in (0.25, 0.5, 0.75):
        m = int(n * frac)
        trial(f"first {m} points, dense_min=0", {"dense_min": 0}, idx[:m].copy(), h[:m].copy(), v[:m].copy())
