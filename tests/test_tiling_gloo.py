"""Multi-rank host logic on CPU: world_size-2 (and 3) gloo process groups.

The compute backend here is a CPU stand-in built on the oracle (test infrastructure) that honours a
row strip exactly like the kernels do; what is under test is the product's tiling logic
(gem_amd/tiling.py): the strip partition, that strips compose to the single-device map, and the
in-place exchange of strips through torch.distributed.
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_strip_bounds_partition():
    from gem_amd.tiling import all_strips, strip_bounds
    for L in (1, 7, 75, 600, 2400):
        for W in (1, 2, 3, 4, 8):
            s = all_strips(L, W)
            assert s[0][0] == 0 and s[-1][1] == L
            assert all(a[1] == b[0] for a, b in zip(s[:-1], s[1:]))
            assert max(r1 - r0 for r0, r1 in s) - min(r1 - r0 for r0, r1 in s) <= 1
    with pytest.raises(ValueError):
        strip_bounds(10, 2, 2)


class StripOracleMap:
    """CPU stand-in with ElevationMap's surface: fuses only the points whose storage row is in its strip."""

    def __init__(self, length, resolution, strip=(0, 0)):
        import torch
        sys.path.insert(0, str(ROOT / "oracle"))
        import oracle
        self._o = oracle.OracleMap(length, resolution)
        self.length = length
        self.row0, self.row1 = strip[0], strip[0] + (strip[1] or length)
        self._t = {}
        self._torch = torch

    def move(self, p):
        return self._o.move(p)

    def mapvar_update(self, u):
        self._o.mapvar_update(u)

    def add(self, frame, xyzi):
        c = np.ascontiguousarray(xyzi, np.float32)
        out = self._o.process_points(frame, c[:, 0], c[:, 1], c[:, 2])
        idx = out["index"].copy()
        rows = idx // self.length
        idx[(idx >= 0) & ((rows < self.row0) | (rows >= self.row1))] = -1        # what k_bin does with fc.row0 / row1
        self._o.fuse(idx, out["height"], out["var"])
        # cells outside the strip are not this rank's business: poison them so the exchange must fix them
        for name in ("elevation", "variance"):
            a = self._o.layer(name)
            a[:self.row0] = np.nan; a[self.row1:] = np.nan
            self._o.set_layer(name, a)
        self._t.clear()

    def layer_tensor(self, name):
        if name not in self._t:
            self._t[name] = self._torch.from_numpy(self._o.layer(name))
        return self._t[name]

    def layer(self, name):
        return self._t[name].numpy() if name in self._t else self._o.layer(name)


def _worker(rank, world, port, L, out_dir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
    import torch.distributed as dist
    from gem_amd import synth
    from gem_amd.tiling import TiledElevationMap
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        wl = synth.config_c1()
        tm = TiledElevationMap(L, wl.resolution, rank, world, make_map=StripOracleMap, exchange="torch")
        tm.move([0.35, -0.2, 0.0])
        for rep in range(2):
            tm.mapvar_update(1e-5)
            tm.add(wl.frames[0], wl.clouds[0])
            tm.allgather()
        np.save(Path(out_dir) / f"e{rank}.npy", tm.layer("elevation"))
        np.save(Path(out_dir) / f"v{rank}.npy", tm.layer("variance"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,L", [(2, 200), (3, 200)])
def test_tiled_map_equals_single_device(tmp_path, world, L, oracle_mod):
    import torch.multiprocessing as mp
    from gem_amd import synth
    port = free_port()
    mp.spawn(_worker, args=(world, port, L, str(tmp_path)), nprocs=world, join=True)
    wl = synth.config_c1()
    ref = oracle_mod.OracleMap(L, wl.resolution)
    ref.move([0.35, -0.2, 0.0])
    for rep in range(2):
        ref.mapvar_update(1e-5); ref.add(wl.frames[0], wl.clouds[0])
    for r in range(world):
        e, v = np.load(tmp_path / f"e{r}.npy"), np.load(tmp_path / f"v{r}.npy")
        assert np.array_equal(e, ref.layer("elevation")), f"rank {r} elevation"
        assert np.array_equal(v, ref.layer("variance")), f"rank {r} variance"
