"""Multi-GPU tiling with the POINTS sharded (SURVEY.md 8e stage B): every rank projects / bins / sorts its own contiguous
share of the batch for the whole map, the sorted records of every strip go to the strip's owner, which walks its cells through
the sources in rank order.  Rank order is input order, so the tiled map equals the single-device one bit for bit.

CPU (gloo, world 2 and 3): the product's host logic -- gem_amd/tiling.py: shard_batch, tile_strip_rows, route_sorted_records,
TiledElevationMap.add_sharded with the exchange carried by torch.distributed -- over a stand-in map built on the oracle.
GPU: the C ABI's two halves (gem_shard_sort_device / gem_shard_fuse_device) with W handles on ONE device standing for W ranks
(device pointers are valid across handles there), and the whole call incl. the RCCL exchange with a single rank."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))

from gem_amd import synth  # noqa: E402
from gem_amd.tiling import route_sorted_records, shard_batch, tile_strip_rows  # noqa: E402

F32 = np.float32


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def small_batch(n_sweeps=5, per=9000, seed=7, L=96, res=0.1):
    """A batch with sweeps of very different sizes (one empty), overlapping footprints, variance increments."""
    rng = np.random.default_rng(seed)
    sizes = [per, per // 3, 0, per * 2, per // 2][:n_sweeps]
    f0 = synth._frame_for(synth.pose_matrix(0.1, -0.2, 0.0, yaw=0.3), synth.SensorModel.velodyne())
    f1 = synth._frame_for(synth.pose_matrix(-0.3, 0.1, 0.05, yaw=-0.7), synth.SensorModel.velodyne())
    clouds = [synth.random_cloud(int(rng.integers(1 << 30)), n, 0.55 * L * res, z_sigma=0.2, dup_fraction=0.3) for n in sizes]
    frames = [f0 if k % 2 else f1 for k in range(n_sweeps)]
    off = np.concatenate([[0], np.cumsum(sizes)])
    upd = [1e-5 * (1 + k % 3) for k in range(n_sweeps)]
    return L, res, frames, clouds, off, upd


def oracle_reference(oracle_mod, L, res, frames, clouds, upd, position=None):
    ref = oracle_mod.OracleMap(L, res)
    if position is not None:
        ref.move(position)
    for k, (f, c) in enumerate(zip(frames, clouds)):
        if upd is not None:
            ref.mapvar_update(upd[k])
        ref.add(f, c)
    return ref


# ---- host logic --------------------------------------------------------------------------------------------------------------
def test_shard_batch_partitions_the_index_range():
    for off in ([0, 10, 30, 30, 45], [0, 0, 10, 10, 10, 20, 20], [0, 7], [0, 0, 0]):
        n = off[-1]
        for W in (1, 2, 3, 5, 8):
            pos = 0
            for r in range(W):
                first, local = shard_batch(off, W, r)
                assert local[0] == (n * r) // W and local[-1] == (n * (r + 1)) // W and local[0] == pos
                assert all(a <= b for a, b in zip(local[:-1], local[1:]))
                for i in range(len(local) - 1):                       # local sweep i lies inside global sweep first + i
                    assert off[first + i] <= local[i] and local[i + 1] <= off[first + i + 1]
                pos = local[-1]
            assert pos == n


def test_tile_strip_rows_and_routing_table():
    for L in (40, 75, 600, 2400):
        for W in (1, 2, 3, 8):
            rows = tile_strip_rows(L, W)
            assert rows[0] == 0 and rows[-1] == L and all(a <= b for a, b in zip(rows[:-1], rows[1:]))
            assert all(r % 32 == 0 or r == L for r in rows)
    bounds = [[0, 3, 3, 10], [0, 0, 4, 4], [0, 5, 6, 9]]
    send, recv = route_sorted_records(bounds, 1)
    assert send == [(0, 0), (0, 4), (4, 4)] and recv == [0, 4, 1]
    assert sum(route_sorted_records(bounds, r)[1][s] for r in range(3) for s in range(3)) == 10 + 4 + 9


class ShardOracleMap:
    """CPU stand-in with the surface TiledElevationMap.add_sharded drives (pack_batch, shard_sort_tensors, shard_fuse_tensors),
    built on the oracle: projection by gemo_process_points, a stable sort by cell id, fusion of the strip's cells in the order
    the sources arrive."""

    def __init__(self, length, resolution, strip=(0, 0)):
        import oracle
        self._o = oracle.OracleMap(length, resolution)
        self.length = length
        self.row0, self.row1 = strip[0], strip[0] + (strip[1] or length)
        self.tpr = (length + 31) // 32
        self.id_bits = 10 + max(1, int(np.ceil(np.log2(self.tpr * self.tpr))))

    def move(self, p):
        return self._o.move(p)

    @staticmethod
    def pack_batch(frames, offsets, var_updates=None):
        return (list(frames), [int(v) for v in offsets])

    def shard_sort_tensors(self, pb, xyzi, first, n_global, strip_rows, first_point_in_sweep=0):
        import torch
        frames, off = pb
        xyzi = np.asarray(xyzi)
        ids, hv, sw = [], [], []
        L = self.length
        for i, f in enumerate(frames):
            c = xyzi[off[i]:off[i + 1]]
            if c.shape[0] == 0:
                continue
            oi = np.arange(c.shape[0], dtype=np.int32) + (first_point_in_sweep if i == 0 else 0)      # index of the point inside its sweep
            out = self._o.process_points(f, c[:, 0], c[:, 1], c[:, 2], orig_index=oi)
            keep = (out["index"] >= 0) & (out["height"] != -1.0)
            cell = out["index"][keep]
            row, col = cell // L, cell % L
            ids.append((((row >> 5) * self.tpr + (col >> 5)) << 10) | ((row & 31) << 5) | (col & 31))
            hv.append(np.stack([out["height"][keep], out["var"][keep]], 1).astype(F32))
            sw.append(np.full(cell.shape[0], first + i, np.int64))
        if not ids:
            return np.zeros(len(strip_rows), np.int64), torch.empty((0, 2), dtype=torch.int32), torch.empty((0,), dtype=torch.int32)
        ids, hv, sw = np.concatenate(ids), np.concatenate(hv), np.concatenate(sw)
        order = np.argsort(ids, kind="stable")
        ids, hv, sw = ids[order], hv[order], sw[order]
        first_id = [((min(r, 32 * self.tpr) // 32 if r < L else self.tpr) * self.tpr) << 10 for r in strip_rows]
        bounds = np.searchsorted(ids, first_id, side="left").astype(np.int64)
        key = (ids | (sw << self.id_bits)).astype(np.uint32).view(np.int32)
        return bounds, torch.from_numpy(hv.view(np.int32).copy()), torch.from_numpy(key.copy())

    def shard_fuse_tensors(self, hv_list, key_list, n_global, var_updates=None):
        hv = np.concatenate([t.numpy().view(F32).reshape(-1, 2) for t in hv_list]) if hv_list else np.zeros((0, 2), F32)
        key = np.concatenate([t.numpy().view(np.uint32) for t in key_list]).astype(np.int64) if key_list else np.zeros(0, np.int64)
        ids, sw = key & ((1 << self.id_bits) - 1), key >> self.id_bits
        tile, cell = ids >> 10, ids & 1023
        row, col = (tile // self.tpr) * 32 + (cell >> 5), (tile % self.tpr) * 32 + (cell & 31)
        assert np.all((row >= self.row0) & (row < self.row1)), "a record outside this rank's strip was routed here"
        index = (row * self.length + col).astype(np.int32)
        for s in range(n_global):                                          # the records of a cell arrive in input order: the sweeps apart,
            if var_updates is not None:                                    # each sweep's records in arrival order
                self._o.mapvar_update(var_updates[s])
            m = sw == s
            self._o.fuse(index[m], hv[m, 0], hv[m, 1])

    def layer_tensor(self, name):
        import torch
        m = self._o._m.contents
        n = self.length * self.length
        return torch.from_numpy(np.ctypeslib.as_array(getattr(m, name), (n,)).reshape(self.length, self.length))

    def layer(self, name):
        return self._o.layer(name)


def _worker_sharded(rank, world, port, q):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import oracle
        from gem_amd.tiling import TiledElevationMap
        L, res, frames, clouds, off, upd = small_batch()
        ref = oracle_reference(oracle, L, res, frames, clouds, upd, position=[0.7, -0.4, 0.0])
        tm = TiledElevationMap(L, res, rank, world, make_map=ShardOracleMap, exchange="torch", tile_strips=True)
        tm.move([0.7, -0.4, 0.0])
        tm.add_sharded(frames, np.concatenate(clouds), off, upd)
        tm.allgather()
        ok = all(np.array_equal(tm.layer(n), ref.layer(n)) for n in ("elevation", "variance"))
        touched = int((ref.layer("elevation") != -10).sum())
        dist.destroy_process_group()
        q.put((rank, ok, touched))
    except Exception as e:      # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc() + str(e)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_routing_over_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_sharded, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info in res:
        assert ok, f"rank {rank}: {info}"
        assert info > 2000


# ---- GPU ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("use_ranges", [False, True])
@pytest.mark.parametrize("world,L,res", [(2, 96, 0.1), (3, 75, 0.2), (8, 600, 0.05), (3, 96, 0.1)])
def test_sharded_halves_with_world_handles_on_one_device(oracle_mod, world, L, res, use_ranges):
    """W handles on one device stand for W ranks: every handle sorts its share (gem_shard_sort_device), the routing table says
    which part of whose records goes where, every owner walks its strip through all W sources (gem_shard_fuse_device) -- finding
    every block's records by search, or through the sources' own block ranges (what gem_add_sharded_device exchanges).  The
    (3, 96) case uses the STEREO sensor model, whose variance depends on a point's index inside its sweep: the shards that hold
    the tail of a split sweep have to be told where it begins."""
    import torch
    from gem_amd import ElevationMap
    from gem_amd.tiling import first_point_in_sweep
    if L == 600:
        wl = synth.config_c4(n_sweeps=6)
        frames, clouds, upd = wl.frames, wl.clouds, wl.var_updates
        off = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])])
    else:
        _, _, frames, clouds, off, upd = small_batch(L=L, res=res)
        if world == 3 and L == 96:
            import copy
            frames = [copy.deepcopy(f) for f in frames]
            for f in frames:
                f.model = synth.SensorModel(2, (0.1, 0.001, 380.0, 1.0, 0.002, 0.001, 30.0), original_width=97)
            clouds = [np.where(np.abs(c[:, 2:3]) < 0.05, 0.05, c).astype(F32) if c.shape[0] else c for c in clouds]      # (z = 0: infinite disparity)
            for c in clouds:
                if c.shape[0]:
                    c[:, 2] = np.where(np.abs(c[:, 2]) < 0.05, 0.05, c[:, 2])
    pos = [0.7, -0.4, 0.0]
    ref = oracle_reference(oracle_mod, L, res, frames, clouds, upd, position=pos)
    cat = torch.from_numpy(np.concatenate(clouds)).cuda()
    rows = tile_strip_rows(L, world)
    maps = [ElevationMap(L, res, strip=(rows[r], rows[r + 1] - rows[r])) for r in range(world)]
    for rep in range(2):                                                   # the second batch fuses into the populated strips
        sorted_ = []
        for r, m in enumerate(maps):
            if rep == 0:
                m.move(pos)
            first, local = shard_batch(off, world, r)
            pb = m.pack_batch([frames[first + i] for i in range(len(local) - 1)], local, None)
            sorted_.append(m.shard_sort(pb, cat, first, len(frames), rows, first_point_in_sweep(off, first, local), with_ranges=True))
        bounds = [sr[0].tolist() for sr in sorted_]
        tpr = (L + 31) // 32
        for r, m in enumerate(maps):
            _, recv = route_sorted_records(bounds, r)
            hv = [sorted_[s][1] + 8 * bounds[s][r] if recv[s] else 0 for s in range(world)]
            key = [sorted_[s][2] + 4 * bounds[s][r] if recv[s] else 0 for s in range(world)]
            if use_ranges:
                blk0 = 4 * tpr * (rows[r] // 32)                           # first block of the owner's strip
                rng = [sorted_[s][3] + 8 * blk0 if recv[s] else 0 for s in range(world)]
                m.shard_fuse(hv, key, recv, len(frames), upd, range_ptrs=rng, bases=[bounds[s][r] for s in range(world)])
            else:
                m.shard_fuse(hv, key, recv, len(frames), upd)
        for m in maps:
            m.synchronize()
        if rep == 1:
            for k, (f, c) in enumerate(zip(frames, clouds)):
                ref.mapvar_update(upd[k]); ref.add(f, c)
        for r, m in enumerate(maps):
            for name in ("elevation", "variance"):
                g, o = m.layer(name)[rows[r]:rows[r + 1]], ref.layer(name)[rows[r]:rows[r + 1]]
                if world == 3 and L == 96:                                 # stereo: double pow / sqrt on both sides, the last ulp may differ
                    assert np.array_equal(g == -10, o == -10) and np.allclose(g, o, rtol=1e-5, atol=0), (rep, r, name)
                else:
                    assert np.array_equal(g, o), (rep, r, name)
    assert (ref.layer("elevation") != -10).sum() > 2000


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["rccl", "torch"])
def test_sharded_call_single_rank(oracle_mod, exchange):
    """gem_add_sharded_device with one rank: the whole path incl. ncclAllGather of the strip boundaries and the grouped
    ncclSend / ncclRecv (to itself); and the same through torch.distributed's all_to_all (NCCL == RCCL)."""
    import torch
    import torch.distributed as dist
    from gem_amd import ElevationMap
    from gem_amd.tiling import TiledElevationMap
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        L, res, frames, clouds, off, upd = small_batch()
        ref = oracle_reference(oracle_mod, L, res, frames, clouds, upd)
        cat = torch.from_numpy(np.concatenate(clouds)).cuda()
        tm = TiledElevationMap(L, res, 0, 1, exchange=exchange, unique_id=ElevationMap.comm_unique_id(), tile_strips=True)
        tm.add_sharded(frames, cat, off, upd)
        tm.allgather()
        torch.cuda.synchronize()
        for name in ("elevation", "variance"):
            assert np.array_equal(tm.layer(name), ref.layer(name)), name
        tm.add_sharded(frames, cat, off, None)                             # again, without increments, into the populated map
        for f, c in zip(frames, clouds):
            ref.add(f, c)
        for name in ("elevation", "variance"):
            assert np.array_equal(tm.layer(name), ref.layer(name)), name
    finally:
        dist.destroy_process_group()


def _worker_rccl(rank, world, port, q, L, res):
    try:
        import torch
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        import oracle
        from gem_amd import ElevationMap
        from gem_amd.tiling import TiledElevationMap
        if L >= 600:
            wl = synth.config_c5(n_points=1_500_000, length=L)
            frames, clouds, upd = wl.frames, wl.clouds, None
            off = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])])
        else:
            _, _, frames, clouds, off, upd = small_batch(L=L, res=res)
        ref = oracle_reference(oracle, L, res, frames, clouds, upd)
        uid = [ElevationMap.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        cat = torch.from_numpy(np.concatenate(clouds)).cuda()
        ok = True
        for tiles in (True, False):                                        # stage B (tile strips), then stage A (row strips, uneven for L = 75)
            tm = TiledElevationMap(L, res, rank, world, exchange="rccl", unique_id=uid[0], tile_strips=tiles, device=rank)
            if tiles:
                tm.add_sharded(frames, cat, off, upd)
            else:
                for k, (f, c) in enumerate(zip(frames, clouds)):
                    if upd is not None:
                        tm.mapvar_update(upd[k])
                    tm.add(f, torch.from_numpy(c).cuda())
            tm.allgather()
            tm.map.synchronize()
            ok = ok and all(np.array_equal(tm.layer(n), ref.layer(n)) for n in ("elevation", "variance"))
        dist.destroy_process_group()
        q.put((rank, ok, ""))
    except Exception as e:      # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc() + str(e)))


@pytest.mark.gpu
@pytest.mark.one_pipeline            # (the ranks are processes of their own: the fixture's knobs never reach them)
@pytest.mark.parametrize("world,L,res", [(2, 96, 0.1), (2, 75, 0.2), (8, 75, 0.2), (8, 2400, 0.05)])
def test_real_rccl_ranks_stage_a_and_b(world, L, res):
    """One process per GPU through gem_comm_init(_tiles) / gem_add_sharded_device / gem_allgather_layers: the real RCCL path,
    even and uneven strips.  Needs `world` devices: skipped on the single-GPU boxes."""
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()}")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_rccl, args=(r, world, port, q, L, res)) for r in range(world)]
    for p in procs:
        p.start()
    res_ = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info in res_:
        assert ok, f"rank {rank}: {info}"
