"""The multi-GPU path's own C++ on a ONE-GPU box.

gem_add_sharded_device's W > 1 branch -- the boundary all-gather, the host's count / offset / base arithmetic, the grouped
send / recv of records and block ranges, the walk fed from receive buffers, the deferred second half with its two sets of
buffers -- and gem_allgather_layers' W > 1 body (published copies, gather stream) only run with more than one rank.  Here W handles
of this process on one device join a LOOPBACK communicator (include/gem_hip_debug.h, csrc/gem_transport.hpp): the RCCL calls
become device-to-device copies with the same stream semantics, everything else is the product's code, each handle driven by a
thread of its own like a rank.  Every rank's all-gathered map must equal the oracle's bit for bit.

What this does not cover is RCCL itself (tests/test_sharded.py::test_real_rccl_ranks_stage_a_and_b, which needs W devices)."""
import itertools
import sys
import threading
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))

from gem_amd import synth  # noqa: E402
from gem_amd.tiling import TiledElevationMap, first_point_in_sweep, shard_batch, tile_strip_rows  # noqa: E402
from test_sharded import oracle_reference, small_batch  # noqa: E402

pytestmark = pytest.mark.gpu
_world_ids = itertools.count(1000)


def run_ranks(world, fn):
    """fn(rank) on one thread per rank (ctypes releases the GIL inside the library); the first exception is re-raised."""
    errors = [None] * world

    def body(r):
        try:
            fn(r)
        except BaseException as e:      # noqa: BLE001
            import traceback
            errors[r] = traceback.format_exc() + repr(e)

    threads = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    alive = [r for r, t in enumerate(threads) if t.is_alive()]
    assert not alive, f"ranks {alive} did not finish"
    for r, e in enumerate(errors):
        assert e is None, f"rank {r}:\n{e}"


def make_world(world, L, res, tile_strips=True, debug=None):
    wid = next(_world_ids)
    return [TiledElevationMap(L, res, r, world, exchange="loopback", tile_strips=tile_strips, world_id=wid, debug=debug) for r in range(world)]


def check_all(maps, ref, what=""):
    for r, tm in enumerate(maps):
        for name in ("elevation", "variance"):
            assert np.array_equal(tm.layer(name), ref.layer(name)), (what, r, name)


@pytest.mark.parametrize("rotate", [False, True])
@pytest.mark.parametrize("world,L,res", [(2, 96, 0.1), (3, 75, 0.2), (8, 600, 0.05), (8, 96, 0.1)])
def test_sharded_steps_through_the_loopback(oracle_mod, world, L, res, rotate):
    """Steps of gem_add_sharded_device + gem_allgather_layers on W ranks: uneven strips (L = 75: three tile rows over three ranks;
    L = 96 over eight: five ranks own NOTHING and one sweep of the batch is empty), variance increments, a moved map, two more
    steps into the populated strips without a synchronisation in between.  rotate: the shards' sorts take a pass-buffer set of
    their own (as big shards do), so the second half of every step is deferred to the next call (gem_capi_comm.cpp)."""
    import torch
    if L == 600:
        wl = synth.config_c4(n_sweeps=6)
        frames, clouds, upd = wl.frames, wl.clouds, wl.var_updates
        off = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])])
    else:
        _, _, frames, clouds, off, upd = small_batch(L=L, res=res)
    pos = [0.7, -0.4, 0.0]
    ref = oracle_reference(oracle_mod, L, res, frames, clouds, upd, position=pos)
    cat = torch.from_numpy(np.concatenate(clouds)).cuda()
    maps = make_world(world, L, res, debug={"overlap_min_points": 1} if rotate else None)

    def rank(r):
        tm = maps[r]
        tm.move(pos)
        tm.add_sharded(frames, cat, off, upd)
        if rotate and world > 1:
            assert tm.map.debug_get("step_pending") == 1               # the walk is still to come
        tm.allgather()
        tm.map.synchronize()
        assert tm.map.debug_get("step_pending") == 0
    run_ranks(world, rank)
    check_all(maps, ref, "first step")

    for k, (f, c) in enumerate(zip(frames, clouds)):                       # two more steps, no increments in the second
        ref.mapvar_update(upd[k]); ref.add(f, c)
    for f, c in zip(frames, clouds):
        ref.add(f, c)

    def rank2(r):
        tm = maps[r]
        tm.add_sharded(frames, cat, off, upd)
        tm.allgather()
        tm.add_sharded(frames, cat, off, None)
        tm.allgather()
        tm.map.synchronize()
    run_ranks(world, rank2)
    check_all(maps, ref, "third step")
    assert (ref.layer("elevation") != -10).sum() > 2000
    for tm in maps:
        tm.map.close()


@pytest.mark.parametrize("world", [2, 8])
def test_empty_shards_and_ranks_that_receive_nothing(oracle_mod, world):
    """Five points over eight ranks (three shards are empty), all of them in ONE strip (seven ranks receive nothing), then a cloud
    that covers the lower half of the map only."""
    import torch
    L, res = 256, 0.1
    f = synth._frame_for(synth.pose_matrix(0.0, 0.0, 0.0), synth.SensorModel.velodyne())
    pts = np.array([[11.0, 3.0, 0.2, 1.0], [11.05, 3.0, 0.25, 1.0], [11.0, 3.02, 0.1, 1.0], [10.9, -2.0, 0.3, 1.0], [11.0, 3.0, 0.21, 1.0]], np.float32)
    half = synth.random_cloud(5, 20000, 0.45 * L * res, z_sigma=0.2, dup_fraction=0.3)
    half = half[half[:, 0] < -2.0]                                         # geographic rows of the lower half: some strips get nothing
    maps = make_world(world, L, res)
    ref = oracle_mod.OracleMap(L, res)
    for cloud in (pts, half, pts):
        ref.add(f, cloud)
    d = [torch.from_numpy(c).cuda() for c in (pts, half)]

    def rank(r):
        tm = maps[r]
        tm.add_sharded([f], d[0], [0, d[0].shape[0]], None)
        tm.allgather()
        tm.add_sharded([f], d[1], [0, d[1].shape[0]], None)
        tm.add_sharded([f], d[0], [0, d[0].shape[0]], None)                # (two steps, one gather)
        tm.allgather()
        tm.map.synchronize()
    run_ranks(world, rank)
    check_all(maps, ref)
    for tm in maps:
        tm.map.close()


def test_steps_interleaved_with_whole_map_operations(oracle_mod):
    """add_sharded / allgather / move / mapvar_update / get_layer in a loop: every whole-map operation comes behind the
    pending step's walk and the all-gather in flight, on every rank."""
    import torch
    world, L, res = 3, 160, 0.1
    _, _, frames, clouds, off, upd = small_batch(L=L, res=res, per=6000)
    cat = torch.from_numpy(np.concatenate(clouds)).cuda()
    maps = make_world(world, L, res, debug={"overlap_min_points": 1})
    ref = oracle_mod.OracleMap(L, res)
    positions = [[0.3, 0.2, 0.0], [0.9, -0.5, 0.0], [0.9, -0.5, 0.0], [-1.2, 0.4, 0.0]]
    snapshots = [[] for _ in range(world)]

    def rank(r):
        tm = maps[r]
        for it, pos in enumerate(positions):
            tm.move(pos)
            tm.add_sharded(frames, cat, off, upd if it % 2 == 0 else None)
            tm.allgather()
            if it == 1:
                tm.mapvar_update(3e-5)                                       # queued behind the pending step's walk, folded into the next one
            if it == 2:
                tm.map.map_optmove([pos[0], pos[1]], 0.01)                 # a whole-map pass: behind the all-gather in flight
                tm.allgather()
            snapshots[r].append((tm.layer("elevation").copy(), tm.layer("variance").copy()))
    run_ranks(world, rank)
    for it, pos in enumerate(positions):
        ref.move(pos)
        for k, (f, c) in enumerate(zip(frames, clouds)):
            if it % 2 == 0:
                ref.mapvar_update(upd[k])
            ref.add(f, c)
        if it == 1:
            ref.mapvar_update(3e-5)
        if it == 2:
            ref.map_optmove([pos[0], pos[1]], 0.01)
        for r in range(world):
            assert np.array_equal(snapshots[r][it][0], ref.layer("elevation")), (it, r)
            assert np.array_equal(snapshots[r][it][1], ref.layer("variance")), (it, r)
    for tm in maps:
        tm.map.close()


def test_step_phase_time_stamps(oracle_mod):
    """bench.py --gpus N reads the device time of every phase of a multi-rank step from events on the streams the phases run on
    (gem_debug_get "step_*_ns", recorded while gem_set_timing is on): all five must be there after a synchronisation."""
    import torch
    world, L, res = 2, 160, 0.1
    _, _, frames, clouds, off, upd = small_batch(L=L, res=res, per=8000)
    cat = torch.from_numpy(np.concatenate(clouds)).cuda()
    maps = make_world(world, L, res, debug={"overlap_min_points": 1})
    got = [None] * world

    def rank(r):
        tm = maps[r]
        tm.map.set_timing(True)
        for _ in range(3):
            tm.add_sharded(frames, cat, off, upd)
            tm.allgather()
        tm.map.synchronize()
        st = tm.map.stats()
        got[r] = {k: tm.map.debug_get(f"step_{k}_ns") for k in ("exchange", "exchange_to_walk", "walk", "publish", "gather")}
        got[r]["launches_walk"] = st["launches_walk"]
        for k in ("step_exchange_bytes_out", "step_exchange_bytes_in", "gather_bytes_out", "gather_bytes_in"):
            got[r][k] = tm.map.debug_get(k)
        tm.map.set_timing(False)
    run_ranks(world, rank)
    for r in range(world):
        assert got[r]["launches_walk"] == 3, got[r]
        for k in ("exchange", "walk", "gather"):
            assert got[r][k] > 0, (r, k, got[r])
        for k in ("exchange_to_walk", "publish"):                          # (hand-overs between two hardware queues: recorded; a few us of skew either way)
            assert got[r][k] > -1_000_000, (r, k, got[r])
    # the byte counts bench.py --gpus N prices against the links: what one rank sends is what the other receives (two ranks), and the
    # all-gather moves every rank's strip of two layers to its peer
    assert got[0]["step_exchange_bytes_out"] == got[1]["step_exchange_bytes_in"] > 0 and got[1]["step_exchange_bytes_out"] == got[0]["step_exchange_bytes_in"] > 0, got
    assert got[0]["gather_bytes_out"] == got[1]["gather_bytes_in"] and got[0]["gather_bytes_out"] + got[1]["gather_bytes_out"] == 8 * L * L, got
    for tm in maps:
        tm.map.close()


@pytest.mark.parametrize("world,L,res", [(3, 75, 0.2), (2, 96, 0.1)])
def test_stage_a_row_strips_through_the_loopback(oracle_mod, world, L, res):
    """Stage A (every rank bins the whole cloud and fuses its row strip; uneven strips for L = 75) + the all-gather with attributes."""
    import torch
    _, _, frames, clouds, off, upd = small_batch(L=L, res=res)
    ref = oracle_reference(oracle_mod, L, res, frames, clouds, upd)
    maps = make_world(world, L, res, tile_strips=False)
    dc = [torch.from_numpy(c).cuda() for c in clouds]

    def rank(r):
        tm = maps[r]
        for k, (f, c) in enumerate(zip(frames, dc)):
            tm.mapvar_update(upd[k])
            tm.add(f, c)
        tm.allgather(with_attributes=True)
        tm.map.synchronize()
    run_ranks(world, rank)
    check_all(maps, ref)
    for tm in maps:
        tm.map.close()


@pytest.mark.parametrize("entry", ["add_device", "add_batch_device", "shard_fuse"])
def test_stage_a_gathers_behind_a_walk_left_to_the_next_call(oracle_mod, entry):
    """ADVICE r5 high + medium: a sorted pass over caller-owned device input leaves its walk to "the next call"
    (gem_handle::dwalk).  gem_allgather_layers IS that next call: the strip it publishes must hold the newest frame (add_device: one
    cloud of >= sort_min_points; add_batch_device: a batch of >= sort_min_points_batch, attr == 0 both).  And gem_shard_fuse_device
    behind such a pass must fuse AFTER the pass's walk (the recurrence is order dependent)."""
    import torch
    world, L, res = 2, 256, 0.1
    f = synth._frame_for(synth.pose_matrix(0.0, 0.0, 0.0), synth.SensorModel.velodyne())
    big = synth.random_cloud(11, 260000, 0.45 * L * res, z_sigma=0.2, dup_fraction=0.3)
    small = synth.random_cloud(12, 30000, 0.45 * L * res, z_sigma=0.3, dup_fraction=0.3)
    maps = make_world(world, L, res, tile_strips=(entry == "shard_fuse"))
    ref = oracle_mod.OracleMap(L, res)
    d_big, d_small = torch.from_numpy(big).cuda(), torch.from_numpy(small).cuda()
    left = [0] * world

    if entry == "add_device":
        ref.add(f, small); ref.add(f, big)
    elif entry == "add_batch_device":
        for _ in range(2):
            ref.mapvar_update(2e-5); ref.add(f, big)
    else:
        ref.add(f, big); ref.add(f, small)

    def rank(r):
        tm = maps[r]
        m = tm.map
        before = m.debug_get("walks_left")
        if entry == "add_device":
            tm.add(f, d_small)
            tm.add(f, d_big)                                                 # cell-sorted, walk left behind
        elif entry == "add_batch_device":
            off = np.array([0, big.shape[0], 2 * big.shape[0]])
            m.add_batch([f, f], torch.cat([d_big, d_big]), off, [2e-5, 2e-5])
        else:
            # `small` sorted FIRST (its records copied out of the handle's arenas), then the big pass, whose walk is left behind,
            # then the fuse of this strip's share of `small`: it must come after that walk
            pb = m.pack_batch([f], [0, small.shape[0]])
            rows = tile_strip_rows(L, world)
            bounds, hv, key = m.shard_sort_tensors(pb, d_small, 0, 1, rows)
            lo, hi = int(bounds[r]), int(bounds[r + 1])
            hv, key = hv[lo:hi].clone(), key[lo:hi].clone()
            m.add(f, d_big)
            m.shard_fuse_tensors([hv], [key], 1)
        left[r] = m.debug_get("walks_left") - before
        tm.allgather()
        m.synchronize()
    run_ranks(world, rank)
    assert all(n >= 1 for n in left), f"no pass left its walk to the next call ({left}): the test does not reach the path"
    check_all(maps, ref, entry)
    for tm in maps:
        tm.map.close()


def test_c5_over_eight_loopback_ranks_equals_the_committed_digest(pipeline):
    """BASELINE configs[4] at full size: 10^7 points -> 2400 x 2400 over eight ranks (one device), two steps; after the first, every
    rank's all-gathered map must have the committed digest of the ONE-device map (tests/golden/digests.json c5_full)."""
    if pipeline != "default":
        pytest.skip("one pipeline is enough at this size (the sharded path is block-sorted whatever the knobs say)")
    import hashlib
    import json
    import torch
    world = 8
    wl = synth.config_c5()
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    cat = torch.from_numpy(np.concatenate(wl.clouds)).cuda()
    d = json.loads((ROOT / "tests" / "golden" / "digests.json").read_text())["c5_full"]
    maps = make_world(world, wl.length, wl.resolution)
    digests = [None] * world

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    def rank(r):
        tm = maps[r]
        tm.map.reserve(int(off[-1]), len(wl.frames))
        before = tm.map.debug_get("arena_allocations")
        tm.add_sharded(wl.frames, cat, off, None)
        tm.allgather()
        tm.map.synchronize()
        digests[r] = (sha(tm.layer("elevation")), sha(tm.layer("variance")))
        for _ in range(3):                                                 # steady state: deferred second halves, rotating buffers
            tm.add_sharded(wl.frames, cat, off, None)
            tm.allgather()
        tm.map.synchronize()
        assert tm.map.debug_get("arena_allocations") == before, "a step allocated after gem_reserve"
    run_ranks(world, rank)
    for r in range(world):
        assert digests[r] == (d["elevation"], d["variance"]), r
    e0, v0 = maps[0].layer("elevation"), maps[0].layer("variance")
    for tm in maps[1:]:                                                    # after four steps the replicas still agree
        assert np.array_equal(tm.layer("elevation"), e0) and np.array_equal(tm.layer("variance"), v0)
    for tm in maps:
        tm.map.close()
