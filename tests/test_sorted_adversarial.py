"""The sorted pipeline (gem_sort.hip) on inputs chosen against its structure: maps smaller than a tile, a single cell flooded by
a whole cloud (one lane walks a chain of 200 000), a big map where nearly every record is alone in its cell, batches of more
than 64 sweeps (the walk's increment table spans several loads) with empty sweeps and sweeps that miss the map, chunk
boundaries (clouds of 4095 / 4096 / 4097 points), map sizes that are not multiples of 32, and the pass counts 2 and 3 for
each.  Exact equality with the oracle throughout."""
import numpy as np
import pytest

from gem_amd import ElevationMap, SensorModel, synth

pytestmark = pytest.mark.gpu
F32 = np.float32


def frame(x=0.0, y=0.0, yaw=0.0):
    return synth._frame_for(synth.pose_matrix(x, y, 0.0, yaw=yaw), SensorModel.velodyne())


def check(gpu, ref, what=""):
    for name in ("elevation", "variance"):
        g, o = gpu.layer(name), ref.layer(name)
        bad = np.flatnonzero(g.ravel() != o.ravel())
        assert bad.size == 0, f"{what} {name}: {bad.size} cells differ, first {bad[:4]}, gpu {g.ravel()[bad[:4]]} oracle {o.ravel()[bad[:4]]}"


@pytest.mark.parametrize("passes", [0, 2, 3])
@pytest.mark.parametrize("L,res", [(8, 0.5), (33, 0.2), (75, 0.2), (1000, 0.05)])
def test_map_sizes_and_sparse_records(oracle_mod, L, res, passes):
    import torch
    knobs = {"sort_min_points": 1, "sort_passes": passes}
    gpu, ref = ElevationMap(L, res, debug=knobs), oracle_mod.OracleMap(L, res)
    f = frame(0.1, -0.2, 0.4)
    n = 300_000 if L == 1000 else 20_000                     # L = 1000: 10^6 cells, most records alone in their cell
    c = synth.random_cloud(3 + L, n, 0.55 * L * res, z_sigma=0.2, dup_fraction=0.05)
    for m in (gpu, ref):
        m.move([0.3 * L * res * 0.1, -0.2, 0.0])
    for rep in range(2):
        gpu.mapvar_update(2e-5); ref.mapvar_update(2e-5)
        gpu.add(f, torch.from_numpy(c).cuda()); ref.add(f, c)
        check(gpu, ref, f"L={L} rep={rep}")


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4095, 4096, 4097, 8192 + 1])
def test_chunk_boundaries(oracle_mod, n):
    import torch
    gpu, ref = ElevationMap(96, 0.1, debug={"sort_min_points": 1}), oracle_mod.OracleMap(96, 0.1)
    c = synth.random_cloud(n, n, 4.0, z_sigma=0.1, dup_fraction=0.5)
    gpu.add(frame(), torch.from_numpy(c).cuda()); ref.add(frame(), c)
    check(gpu, ref, f"n={n}")
    # ... and as three sweeps of a batch, the middle one empty
    off = np.array([0, n // 2, n // 2, n])
    gpu.add_batch([frame(), frame(0.1), frame(0.0, 0.1)], torch.from_numpy(c).cuda(), off, [1e-5, 2e-5, 3e-5])
    for k, fr in enumerate([frame(), frame(0.1), frame(0.0, 0.1)]):
        ref.mapvar_update([1e-5, 2e-5, 3e-5][k]); ref.add(fr, c[off[k]:off[k + 1]])
    check(gpu, ref, f"n={n} batch")


def test_one_cell_takes_the_whole_cloud(oracle_mod):
    import torch
    rng = np.random.default_rng(12)
    n = 200_000
    c = np.zeros((n, 4), F32)
    c[:, 0] = 0.52 + rng.uniform(0, 0.04, n); c[:, 1] = -0.33 + rng.uniform(0, 0.04, n)      # one 10 cm cell
    c[:, 2] = np.where(rng.random(n) < 0.02, rng.normal(1.0, 0.5, n), rng.normal(0.2, 0.01, n))   # inliers and outliers
    c[::1000, 0] += 0.3                                                                          # a few neighbours
    gpu, ref = ElevationMap(64, 0.1, debug={"sort_min_points": 1}), oracle_mod.OracleMap(64, 0.1)
    gpu.add(frame(), torch.from_numpy(c).cuda()); ref.add(frame(), c)
    check(gpu, ref, "flood")
    assert (ref.layer("elevation") != -10).sum() <= 4


@pytest.mark.parametrize("with_updates", [True, False])
def test_more_than_64_sweeps_with_gaps(oracle_mod, with_updates):
    """100 sweeps: most tiny, some empty, some entirely outside the map, negative and zero increments among the positive ones
    (a negative one pushes variances under the floor: the per-sweep floor has to repair them at the right moment)."""
    import torch
    rng = np.random.default_rng(5)
    L, res, ns = 96, 0.1, 100
    sizes = rng.integers(0, 1500, ns); sizes[[3, 17, 64, 65, 99]] = 0; sizes[40] = 30_000
    clouds = [synth.random_cloud(700 + k, int(s), 5.0, z_sigma=0.15, dup_fraction=0.4) for k, s in enumerate(sizes)]
    clouds[50][:, 0] += 100.0                                                  # a sweep that misses the map
    frames = [frame(0.01 * k, -0.005 * k, 0.02 * k) for k in range(ns)]
    upd = (rng.uniform(0, 5e-5, ns)).astype(F32).tolist()
    upd[10], upd[11], upd[70] = -8e-5, 0.0, -2e-4
    off = np.concatenate([[0], np.cumsum(sizes)])
    cat = torch.from_numpy(np.concatenate(clouds)).cuda()
    gpu, ref = ElevationMap(L, res, debug={"sort_min_points": 1}), oracle_mod.OracleMap(L, res)
    for rep in range(2):
        gpu.add_batch(frames, cat, off, upd if with_updates else None)
        for k in range(ns):
            if with_updates:
                ref.mapvar_update(upd[k])
            ref.add(frames[k], clouds[k])
        check(gpu, ref, f"rep {rep}")
    # per-sweep statistics: distinct touched cells summed over the sweeps, accepted points
    gpu.set_counting(True)
    gpu.add_batch(frames, cat, off, upd if with_updates else None)
    st = gpu.stats()
    acc = cells = 0
    for k in range(ns):
        if with_updates:
            ref.mapvar_update(upd[k])
        ref.add(frames[k], clouds[k]); acc += ref.last_counts[0]; cells += ref.last_counts[1]
    check(gpu, ref, "counting")
    assert st["cells_touched"] == cells
    assert st["points_binned"] <= acc                                           # binned = accepted AND inside the map


def test_big_batch_on_two_streams_then_small_passes(oracle_mod):
    """Passes of very different kinds back to back on one handle: the buffers of the two pipelines and of the two streams must
    not step on each other (sorted + second stream, a single sweep through k_frame, a small batch on the tile pipeline, ...)."""
    import torch
    wl = synth.config_c4(n_sweeps=10)
    gpu, ref = ElevationMap(wl.length, wl.resolution), oracle_mod.OracleMap(wl.length, wl.resolution)
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    cat = torch.from_numpy(np.concatenate(wl.clouds)).cuda()
    d = [torch.from_numpy(c).cuda() for c in wl.clouds]
    for rnd in range(3):
        gpu.add_batch(wl.frames, cat, off, wl.var_updates)                      # 1.3 M points: sorted pipeline, two streams
        gpu.add(wl.frames[0], d[0]); gpu.add(wl.frames[1], d[1])                # single sweeps: k_frame, fusion deferred
        gpu.add_batch(wl.frames[:2], cat[:off[2]], off[:3], None)               # 262 k points in 2 sweeps: tile pipeline
        gpu.add_batch(wl.frames, cat, off, None)                                # sorted again, right behind it
        for k in range(10):
            ref.mapvar_update(wl.var_updates[k]); ref.add(wl.frames[k], wl.clouds[k])
        ref.add(wl.frames[0], wl.clouds[0]); ref.add(wl.frames[1], wl.clouds[1])
        for k in range(2):
            ref.add(wl.frames[k], wl.clouds[k])
        for k in range(10):
            ref.add(wl.frames[k], wl.clouds[k])
        check(gpu, ref, f"round {rnd}")


@pytest.mark.parametrize("period", [1, 2, 3, 5, 7, 31, 32, 33, 63, 64, 65])
def test_cell_patterns_inside_a_wave_step(oracle_mod, period):
    """The scatter ranks 64 consecutive records at a time; the lanes of a cell find each other through the LDS.  Clouds whose
    points visit `period` cells cyclically (so a wave step holds every mix from 64 records of one cell to 64 different cells),
    with every third-or-so point rejected by the height window (the records are compacted per wave before they are ranked) and
    heights that make every swap inside a cell visible (the recurrence is order dependent)."""
    import torch
    L, res = 96, 0.1
    rng = np.random.default_rng(100 + period)
    n = 3 * 4096 + 777
    k = np.arange(n)
    cell = (k % period) if period > 1 else np.zeros(n, np.int64)
    # the cells of the cycle lie on a line through several 32x32 tiles; the first digit of the id changes with every one
    cx = 0.5 * res + res * ((cell * 7) % (L - 8) - (L - 8) // 2)
    cy = 0.5 * res + res * ((cell * 3) % 11 - 5)
    z = rng.normal(0.3, 0.4, n).astype(F32)
    z[rng.random(n) < 0.3] = 50.0                            # outside the window: rejected
    c = np.stack([cx, cy, z, np.ones(n)], 1).astype(F32)
    f = frame(); f.lower, f.upper = -5.0, 5.0
    for knobs in ({"sort_min_points": 1}, {"sort_min_points": 1, "sort_passes": 3}):
        gpu, ref = ElevationMap(L, res, debug=knobs), oracle_mod.OracleMap(L, res)
        d = torch.from_numpy(c).cuda()
        for rep in range(2):
            gpu.add(f, d); ref.add(f, c)
            check(gpu, ref, f"period={period} rep={rep} {knobs}")
        off = np.array([0, 1000, 1001, 5000, n])             # the same cloud as four sweeps with increments in between
        vu = [1e-4, 0.0, 3e-5, 2e-4]
        gpu.add_batch([f] * 4, d, off, vu)
        for s in range(4):
            ref.mapvar_update(vu[s]); ref.add(f, c[off[s]:off[s + 1]])
        check(gpu, ref, f"period={period} batch {knobs}")
