"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bar (BASELINE.json north_star): cell indices bit-exact; fused height / variance within
1e-5 relative.  The kernels replay the oracle's float operations in the same order with FMA
contraction off, so most comparisons are in fact checked for exact equality.
"""
import numpy as np
import pytest

from gem_amd import ElevationMap, Frame, RejectFilter, SensorModel, synth

pytestmark = pytest.mark.gpu
F32 = np.float32
REL_TOL = 1e-5          # north_star: "fused height/variance within 1e-5 relative"


def rel_err(a, b):
    return float(np.max(np.abs(a.astype(np.float64) - b) / np.maximum(np.abs(b.astype(np.float64)), 1e-30)))


def assert_maps_match(gpu, ref, exact=True, layers=("elevation", "variance")):
    for name in layers:
        g, o = gpu.layer(name), ref.layer(name)
        if name == "elevation":
            assert np.array_equal(g == -10, o == -10), "set of non-empty cells differs"
        if exact:
            bad = np.flatnonzero(g.ravel() != o.ravel())
            assert bad.size == 0, f"{name}: {bad.size} cells differ, first {bad[:5]}, gpu {g.ravel()[bad[:5]]} oracle {o.ravel()[bad[:5]]}"
        else:
            assert rel_err(g, o) <= REL_TOL, f"{name}: rel err {rel_err(g, o)}"


def make_pair(oracle_mod, L, res, **kw):
    return ElevationMap(L, res, **kw), oracle_mod.OracleMap(L, res, **kw)


# ---- Process_points -------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", ["c1", "c2", "c2_filter"])
def test_process_points_parity(oracle_mod, cfg):
    wl = {"c1": synth.config_c1, "c2": synth.config_c2, "c2_filter": lambda: synth.config_c2(reference_filter=True)}[cfg]()
    gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
    c = wl.clouds[0]
    g = gpu.process_points(wl.frames[0], c[:, 0], c[:, 1], c[:, 2], write_back_xyz=True)
    o = ref.process_points(wl.frames[0], c[:, 0], c[:, 1], c[:, 2], write_back_xyz=True)
    assert np.array_equal(g["index"], o["index"])                   # bit-exact cell indices
    for k in ("var", "x_ts", "y_ts", "height", "x", "y", "z"):
        assert np.array_equal(g[k], o[k]), k
    frac_in = float((o["index"] >= 0).mean())
    assert 0.05 < frac_in < 1.0                                     # both branches exercised


def test_process_points_moved_map_and_odd_length(oracle_mod):
    gpu, ref = make_pair(oracle_mod, 75, 0.2)                       # kitti_demo_map.yaml geometry (odd L)
    for m in (gpu, ref):
        m.move([1.3, -2.9, 0.0])
    c = synth.random_cloud(3, 20000, 9.0)
    f = synth._frame_for(synth.pose_matrix(0.2, 0.1, 0.5, 0.3, 0.02, -0.01), SensorModel.velodyne())
    g = gpu.process_points(f, c[:, 0], c[:, 1], c[:, 2]); o = ref.process_points(f, c[:, 0], c[:, 1], c[:, 2])
    assert np.array_equal(g["index"], o["index"]) and np.array_equal(g["var"], o["var"])


def test_process_points_nonfinite_inputs(oracle_mod):
    gpu, ref = make_pair(oracle_mod, 200, 0.1)
    f = synth.config_c1().frames[0]
    f.lower, f.upper = -np.inf, np.inf
    x = np.array([np.nan, np.inf, -np.inf, 1e38, 1.0, 0.0], F32)
    y = np.array([0, 0, 0, 0, np.nan, 3e38], F32); z = np.zeros(6, F32)
    g = gpu.process_points(f, x, y, z); o = ref.process_points(f, x, y, z)
    assert np.array_equal(g["index"], o["index"]) and np.all(g["index"] == -1)


@pytest.mark.parametrize("bounds", [(0.1, 0.30000000000000004), (-0.7, 0.1 + 1e-9), (1e-45, 1e300), (-1e300, -1e-320), (-np.inf, 0.2),
                                    (0.25, 0.5), (np.nan, 1.0), (0.0, np.nan), (3.5e38, np.inf)])
def test_height_window_edges(oracle_mod, bounds):
    """GPU:397 compares (double)h with DOUBLE bounds; the kernels compare h with float bounds chosen on the host so that the
    decision is the same (gem_capi_core.cpp fill_frame): heights on either side of each bound, one float apart, on all three paths"""
    lo, hi = bounds
    gpu, ref = make_pair(oracle_mod, 64, 0.1)
    f = synth._frame_for(np.eye(4), SensorModel.velodyne()); f.lower, f.upper = lo, hi
    hs = []
    for b in (lo, hi):
        if np.isnan(b):
            continue
        with np.errstate(over="ignore"):
            c = F32(b)
        for k in range(-3, 4):
            v = c
            for _ in range(abs(k)):
                v = np.nextafter(v, F32(np.inf if k > 0 else -np.inf), dtype=F32)
            hs.append(v)
    hs += [F32(0.0), F32(-0.0), F32(0.2), F32(np.inf), F32(-np.inf), F32(np.nan), F32(3.4e38), F32(-3.4e38), F32(1e-45)]
    z = np.array(hs, F32); x = np.full(z.shape, 0.05, F32); y = np.full(z.shape, -0.05, F32)
    g = gpu.process_points(f, x, y, z); o = ref.process_points(f, x, y, z)
    assert np.array_equal(g["index"], o["index"]) and np.array_equal(g["height"], o["height"], equal_nan=True)
    keep = np.isnan(z) | (np.abs(z) < 1e3)          # (heights of 1e38 in one cell fuse to NaN on both sides: nothing to compare)
    c = np.stack([x, y, z, np.ones_like(z)], 1)[keep]
    gpu.add(f, c); ref.add(f, c)
    assert_maps_match(gpu, ref)
    import torch
    gpu.add_batch([f, f], torch.from_numpy(np.concatenate([c, c])).cuda(), np.array([0, len(c), 2 * len(c)]), None)
    ref.add(f, c); ref.add(f, c)
    assert_maps_match(gpu, ref)


@pytest.mark.parametrize("beam_c", [0.0015, 0.0])
@pytest.mark.parametrize("pose", ["identity", "shifted"])
def test_straight_line_projection_on_adversarial_points(oracle_mod, pose, beam_c):
    """Round 6: the laser-only straight-line projection (gem_device.hpp project_bin_laser_fast) takes sqrtf and the two binning
    divisions WITHOUT their rescaling preambles (sqrt_plain, div_binning).  The ranges they are proven for have edges: a sum of squares
    of exactly 0 (LiDARs report (0, 0, 0) for no return), positive but below 2^-96 (the wave falls back to sqrtf), denormal; a shift from
    the map centre of exactly 0, below 2^-60, beyond 2^60, infinite, NaN.  Every one of them through gem_add (tile pipeline, and the
    sorted forms by the pipeline fixture) on a map the fast form qualifies for; beam_c = 0 makes the variance a pure function of the
    square root, so a wrong root near zero shows."""
    L, res = 64, 0.1
    gpu, ref = make_pair(oracle_mod, L, res)
    model = SensorModel(_lib_model_laser(), (0.018, 0.0006, beam_c), 5.0, -5.0)
    T = np.eye(4)
    if pose == "shifted":
        T = synth.pose_matrix(0.25, -0.15, 0.3, 0.4, 0.01, -0.02)
        for m in (gpu, ref):
            m.move([0.25, -0.15, 0.0])                              # the sensor stands on the map centre: shifts of exactly 0
    f = synth._frame_for(T, model)
    tiny = [0.0, -0.0, 1e-45, 1e-40, 1e-30, 3e-20, 1e-19, 2.8e-15, 1e-10]
    pts = []
    for a in tiny:
        for b in (0.0, 1e-30, a):
            pts += [(a, b, 0.0), (b, a, -a), (-a, 0.0, b), (a, a, a)]
    edges = [3.15, 3.1999998, 3.2, 3.2000003, -3.1999998, -3.2, -3.2000003, 0.05, 0.049999997, 0.1, -0.1, 0.15000001]
    pts += [(x, y, 0.01 * k) for k, x in enumerate(edges) for y in (0.0, 0.05, -3.2, 3.1999998)]
    huge = [1e10, -1e10, 1.2e18, 2e19, 3e30, 3.4e38, np.inf, -np.inf, np.nan]
    pts += [(h, 0.3, 0.1) for h in huge] + [(0.3, h, 0.1) for h in huge] + [(0.3, 0.2, h) for h in huge[:4]] + [(h, h, 0.0) for h in huge]
    rng = np.random.default_rng(7)
    pts += [tuple(v) for v in rng.uniform(-3.3, 3.3, (400, 3)) * np.array([1.0, 1.0, 0.1])]
    c = np.array([p + (1.0,) for p in pts], F32)
    c = np.concatenate([c, c[::-1]])                                 # every cell twice, in the other order the second time
    with np.errstate(all="ignore"):
        for _ in range(2):
            gpu.add(f, c); ref.add(f, c)
    assert int((ref.layer("elevation") != -10).sum()) > 200
    assert_maps_match(gpu, ref)
    import torch
    with np.errstate(all="ignore"):
        gpu.add_batch([f, f], torch.from_numpy(np.concatenate([c, c])).cuda(), np.array([0, len(c), 2 * len(c)]), [1e-6, 2e-6])
        for u in (1e-6, 2e-6):
            ref.mapvar_update(u); ref.add(f, c)
    assert_maps_match(gpu, ref)


def _lib_model_laser():
    from gem_amd import _lib
    return _lib.MODEL_LASER


@pytest.mark.parametrize("model", ["structured_light", "stereo", "perfect"])
def test_other_noise_models(oracle_mod, model):
    gpu, ref = make_pair(oracle_mod, 400, 0.025)
    wl = synth.config_c3()
    f = wl.frames[0]
    if model == "structured_light":
        f.model = SensorModel.realsense_d435()
    elif model == "stereo":
        f.model = SensorModel(2, (0.1, 0.001, 380.0, 1.0, 0.002, 0.001, 30.0), original_width=640)
    else:
        f.model = SensorModel.perfect()
    c = wl.clouds[0][:50000]
    oi = wl.orig_index[:50000]
    g = gpu.process_points(f, c[:, 0], c[:, 1], c[:, 2], orig_index=oi); o = ref.process_points(f, c[:, 0], c[:, 1], c[:, 2], orig_index=oi)
    assert np.array_equal(g["index"], o["index"])
    assert rel_err(g["var"][o["index"] >= 0], o["var"][o["index"] >= 0]) <= 1e-5     # double pow()/sqrt() may differ in the last ulp


@pytest.mark.parametrize("model", ["structured_light", "stereo"])
def test_full_c3_fused_with_the_camera_noise_models(oracle_mod, model):
    """The whole depth image (307 200 points, hundreds per cell under the camera) FUSED with the structured-light and the stereo
    variance models -- not only projected (test_other_noise_models) -- through whatever pipeline the run selects, from a host array
    and from a device tensor, twice into the map.  The variances go through double pow / sqrt on both sides and may differ in the
    last ulp, so the bar is north_star's 1e-5 on every cell (and the same set of non-empty cells), not bit equality."""
    import torch
    gpu, ref = make_pair(oracle_mod, 400, 0.025)
    gpu_d = ElevationMap(400, 0.025)
    wl = synth.config_c3(structured_light=(model == "structured_light"))
    f = wl.frames[0]
    if model == "stereo":
        f.model = SensorModel(2, (0.1, 0.001, 380.0, 1.0, 0.002, 0.001, 30.0), original_width=640)
    c, oi = wl.clouds[0], wl.orig_index
    dc, doi = torch.from_numpy(c).cuda(), torch.from_numpy(oi).cuda()
    for m in (gpu, gpu_d, ref):
        m.move(wl.map_position)
    for rep in range(2):
        gpu.add(f, c, orig_index=oi); gpu_d.add(f, dc, orig_index=doi); ref.add(f, c, orig_index=oi)
        assert_maps_match(gpu, ref, exact=False)
        assert_maps_match(gpu_d, ref, exact=False)
    assert (ref.layer("elevation") != -10).sum() > 20000


# ---- the fused add path -----------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", ["c1", "c2", "c2_filter", "c3"])
def test_add_parity(oracle_mod, cfg):
    wl = {"c1": synth.config_c1, "c2": synth.config_c2, "c2_filter": lambda: synth.config_c2(reference_filter=True),
          "c3": synth.config_c3}[cfg]()
    gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
    if wl.map_position is not None:
        for m in (gpu, ref):
            m.move(wl.map_position)
    gpu.set_counting(True)
    for rep in range(2):                       # 2nd pass: non-empty cells, Kalman / Mahalanobis branches
        gpu.add(wl.frames[0], wl.clouds[0]); ref.add(wl.frames[0], wl.clouds[0])
        st = gpu.stats()
        assert st["cells_touched"] == ref.last_counts[1]
        assert_maps_match(gpu, ref)
    assert ref.last_counts[1] > 1000


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 255, 257, 4097])
def test_add_ragged_sizes(oracle_mod, n):
    gpu, ref = make_pair(oracle_mod, 64, 0.1)
    c = synth.random_cloud(n + 5, max(n, 1), 3.5)[:n]
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    gpu.add(f, c); ref.add(f, c)
    assert_maps_match(gpu, ref)
    gpu.add(f, c[::-1].copy()); ref.add(f, c[::-1].copy())
    assert_maps_match(gpu, ref)


def test_add_heavy_collisions(oracle_mod):
    # 200k points into a 40x40 map: ~125 points per cell on average, long in-order chains, several tile batches
    gpu, ref = make_pair(oracle_mod, 40, 0.1)
    c = synth.random_cloud(21, 200_000, 2.2, z_sigma=0.05)
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    gpu.add(f, c); ref.add(f, c)
    assert_maps_match(gpu, ref)


def test_add_single_cell_chain(oracle_mod):
    # every point in ONE cell: the pure sequential recurrence (order must be the input order)
    gpu, ref = make_pair(oracle_mod, 32, 0.1)
    rng = np.random.default_rng(5)
    n = 10_000
    c = np.zeros((n, 4), F32); c[:, 0] = 0.31 + rng.uniform(0, 0.05, n); c[:, 1] = -0.72 + rng.uniform(0, 0.05, n)
    c[:, 2] = rng.normal(0, 0.05, n)
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    gpu.add(f, c); ref.add(f, c)
    assert ref.last_counts[1] <= 4
    assert_maps_match(gpu, ref)


def test_add_all_rejected(oracle_mod):
    gpu, ref = make_pair(oracle_mod, 64, 0.1)
    c = synth.random_cloud(2, 1000, 3.0)
    f = synth._frame_for(np.eye(4), SensorModel.velodyne()); f.lower, f.upper = 50.0, 60.0
    gpu.add(f, c); ref.add(f, c)
    assert_maps_match(gpu, ref)                      # floor pass still ran: variance -10 -> 1e-4 everywhere
    assert np.all(gpu.layer("variance") == F32(1e-4))


def test_add_with_rgb_and_intensity(oracle_mod):
    gpu, ref = make_pair(oracle_mod, 100, 0.1)
    rng = np.random.default_rng(9)
    c = synth.random_cloud(9, 60000, 5.5)            # intensity in {0,1,2,3}
    rgb = (rng.integers(0, 3, (60000, 3)) * 100).astype(np.uint32)     # zeros are common
    packed = (rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2]
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    for _ in range(2):
        gpu.add(f, c, rgb=packed); ref.add(f, c, rgb=packed)
        assert_maps_match(gpu, ref, layers=("elevation", "variance", "intensity", "color_r", "color_g", "color_b"))
    assert (ref.layer("color_r") != 0).sum() > 100


# ---- Fuse with the reference's host arrays ----------------------------------------------------------------
def test_fuse_arrays_parity(oracle_mod):
    gpu, ref = make_pair(oracle_mod, 50, 0.1)
    rng = np.random.default_rng(13)
    n = 30000
    idx = rng.integers(-1, 2500 + 5, n).astype(np.int32)             # includes -1 and out-of-range indices
    h = rng.normal(0, 0.2, n).astype(F32); h[rng.integers(0, n, 50)] = -1.0      # the h == -1 sentinel (GPU:482)
    v = rng.uniform(1e-6, 2e-3, n).astype(F32)
    R, G, B = (rng.integers(0, 3, n).astype(np.int32) * 90 for _ in range(3))
    I = rng.integers(0, 2, n).astype(F32)
    for _ in range(2):
        gpu.fuse(idx, h, v, R, G, B, I); ref.fuse(idx, h, v, R, G, B, I)
        assert_maps_match(gpu, ref, layers=("elevation", "variance", "intensity", "color_r", "color_g", "color_b"))
    gpu.fuse(idx, h, v); ref.fuse(idx, h, v)                         # without attribute arrays
    assert_maps_match(gpu, ref)


@pytest.mark.parametrize("n", [131072 + 1, 180000, 199656, 200000, 420000])
@pytest.mark.parametrize("colours", [False, True])
def test_fuse_arrays_longer_than_one_descriptor_row(oracle_mod, n, colours):
    """Fuse (gpu_process.cu:1154-1193) with more than 131 072 points whose LATER points reach tiles the earlier ones never touch --
    a depth image handed over row by row.  The tile pipeline's descriptor rows hold 2048 units of 64 points: longer inputs are cut
    into sweeps (run_pipeline); until round 4 only clouds were, and a Fuse between 131 073 and the sorted pipeline's threshold
    (200 000) lost every point behind the first 131 072 (found by the soak once it drove gem_fuse)."""
    L = 400
    gpu, ref = make_pair(oracle_mod, L, 0.05)
    rng = np.random.default_rng(n)
    idx = np.sort(rng.integers(0, L * L, n)).astype(np.int32)              # cells in increasing order: the map fills row by row
    idx[rng.integers(0, n, 200)] = -1
    h = rng.normal(0, 0.2, n).astype(F32); v = rng.uniform(1e-6, 2e-3, n).astype(F32)
    h[rng.integers(0, n, 100)] = -1.0                                       # the "rejected" sentinel (GPU:482)
    if colours:
        R, G, B = (rng.integers(0, 3, n).astype(np.int32) * 90 for _ in range(3)); I = rng.integers(0, 2, n).astype(F32)
        gpu.fuse(idx, h, v, R, G, B, I); ref.fuse(idx, h, v, R, G, B, I)
        assert_maps_match(gpu, ref, layers=("elevation", "variance", "intensity", "color_r", "color_g", "color_b"))
    else:
        gpu.fuse(idx, h, v); ref.fuse(idx, h, v)
        assert_maps_match(gpu, ref)
    assert (ref.layer("elevation")[L - 3:] != -10).sum() > 100               # the last rows of the map did get their points


@pytest.mark.parametrize("threads", [0, 1, 4, 8])
def test_host_arrays_through_the_pinned_staging_and_through_the_runtime(oracle_mod, threads):
    """Caller-owned pageable arrays (the node's: gpu_process.cu:1096-1141, :1165-1192, :1283-1291) travel through the handle's pinned
    staging buffer with `copy_threads` threads between it and the arrays (gem_hostcopy.hpp; 0 = handed to the runtime as they are).
    A frame's calls with sizes that grow and shrink (the buffer is reallocated while an upload may still be read), odd sizes,
    sizes under the staging threshold, write-back, every array of Process_points / Fuse / Map_feature / get_layer: equal to the
    oracle whatever the route."""
    import ctypes as C
    L, res = 300, 0.1
    gpu, ref = make_pair(oracle_mod, L, res)
    gpu.debug_set("copy_threads", threads)
    assert gpu.debug_get("copy_threads") == threads
    f = synth.config_c2(reference_filter=True).frames[0]             # (the reject filter looks at a point's index in the cloud)
    for it, n in enumerate([70001, 300, 262144 + 3, 131072, 9, 524288 + 1]):
        c = synth.random_cloud(40 + it, n, 14.0)
        idx = np.arange(n, dtype=np.int32)[::-1].copy() if it % 3 == 0 else None
        g = gpu.process_points(f, c[:, 0], c[:, 1], c[:, 2], orig_index=idx, write_back_xyz=True)
        o = ref.process_points(f, c[:, 0], c[:, 1], c[:, 2], orig_index=idx, write_back_xyz=True)
        for k in ("index", "var", "x_ts", "y_ts", "height", "x", "y", "z"):
            assert np.array_equal(g[k], o[k]), (k, n)
        rng = np.random.default_rng(it)
        R, G, B = (rng.integers(0, 256, n).astype(np.int32) for _ in range(3))
        I = rng.integers(0, 2, n).astype(F32)
        gpu.mapvar_update(1e-6); ref.mapvar_update(1e-6)
        if it % 2:
            gpu.fuse(g["index"], g["height"], g["var"]); ref.fuse(o["index"], o["height"], o["var"])
        else:
            gpu.fuse(g["index"], g["height"], g["var"], R, G, B, I); ref.fuse(o["index"], o["height"], o["var"], R, G, B, I)
        gpu.add(f, c); ref.add(f, c)                                     # (XYZI host array: the fused entry point's upload)
        outs = {k: np.full(L * L, -77, np.int32 if k.startswith("color") else F32) for k in
                ("elevation", "variance", "color_r", "color_g", "color_b", "rough", "slope", "traver", "intensity")}
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        assert gpu._lib.gem_map_feature(gpu._h, *[vp(outs[k]) for k in ("elevation", "variance", "color_r", "color_g", "color_b", "rough", "slope", "traver", "intensity")]) == 0
        feat = ref.map_feature()
        for k, a in outs.items():
            assert np.array_equal(a.reshape(L, L), feat[k] if k in feat else ref.layer(k)), (k, n)
        assert_maps_match(gpu, ref, layers=("elevation", "variance", "intensity", "color_r", "color_g", "color_b"))      # gem_get_layer
    assert (gpu.debug_get("hstage_allocations") > 0) == (threads > 0)


def test_host_array_calls_from_several_threads_on_their_own_handles(oracle_mod):
    """One pool of copy threads serves the process: calls on different handles at the same time (one gets the pool, the others
    copy on their own thread) and all give the oracle's map."""
    import threading
    wl = synth.config_c2()
    ref = oracle_mod.OracleMap(wl.length, wl.resolution)
    c, f = wl.clouds[0], wl.frames[0]
    o = ref.process_points(f, c[:, 0], c[:, 1], c[:, 2]); ref.fuse(o["index"], o["height"], o["var"])
    want = {k: ref.layer(k) for k in ("elevation", "variance")}
    bad = []

    def frame(k):
        for _ in range(3):
            m = ElevationMap(wl.length, wl.resolution)
            g = m.process_points(f, c[:, 0], c[:, 1], c[:, 2]); m.fuse(g["index"], g["height"], g["var"])
            for name, w in want.items():
                if not np.array_equal(m.layer(name), w):
                    bad.append((k, name))
            m.close()

    ts = [threading.Thread(target=frame, args=(k,)) for k in range(4)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not bad, bad


@pytest.mark.parametrize("what", ["variances", "heights", "state", "increments", "cancellation"])
def test_values_outside_the_plain_range_take_the_guarded_chain_loops(oracle_mod, what):
    """The walks' plain chain loops (gem_sort.hip) drop the per-step guards when every record has |h| <= 2^28, 2^-28 <= v <= 2^28,
    every cell starts in range and the increments are in [0, 2^18]; anything else -- tiny, huge, zero, negative, infinite or NaN
    variances, huge or non-finite heights, a map that already holds such values, odd increments -- must be routed to the guarded
    loops, per block (k_fuse_block) or per pass (k_fuse_walk), and still equal the oracle bit for bit.  `cancellation`: in-range
    records whose Kalman numerator sf h + v e cancels to (almost) zero -- the one thing the plain step itself looks at."""
    import torch
    L = 96
    rng = np.random.default_rng({"variances": 1, "heights": 2, "state": 3, "increments": 4, "cancellation": 5}[what])
    n = 60000
    idx = rng.integers(0, L * L, n).astype(np.int32)
    idx[: n // 2] = rng.integers(0, 400, n // 2)                           # long chains in a few blocks, short ones elsewhere
    h = rng.normal(0, 0.2, n).astype(F32)
    v = rng.uniform(3e-4, 2e-3, n).astype(F32)
    incs = [1e-5, 2e-5]
    weird = rng.integers(0, n, 300)
    if what == "variances":
        v[weird] = rng.choice(np.array([0.0, -1e-3, 1e-40, 1e-38, 1e-30, 1e-12, 3e8, 1e30, np.inf, np.nan, 3.7e-9, 2.7e8], F32), weird.size)
    elif what == "heights":
        h[weird] = rng.choice(np.array([3e8, -3e8, 1e30, -1e30, np.inf, -np.inf, np.nan, 2.6e8, 1e-40, -0.0], F32), weird.size)
    elif what == "increments":
        incs = [-1e-5, 3e5]
    elif what == "cancellation":
        h[:] = 0.0; h[rng.integers(0, n, n // 3)] = -0.0                  # e and h both zero: N1 = +-0
    maps = {"hip": ElevationMap(L, 0.1), "oracle": oracle_mod.OracleMap(L, 0.1)}
    if what == "state":
        e0 = np.full((L, L), -10.0, F32); s0 = np.full((L, L), -10.0, F32)
        cells = rng.integers(0, L * L, 500)
        e0.ravel()[cells] = rng.choice(np.array([5e8, -1e20, np.inf, np.nan, 1.0, -0.0], F32), cells.size)
        s0.ravel()[cells] = rng.choice(np.array([5e8, 1e25, np.inf, np.nan, 1e-42, 0.0, -3.0], F32), cells.size)
        for m in maps.values():
            m.set_layer("elevation", e0); m.set_layer("variance", s0)
    for rep in range(2):
        for k, inc in enumerate(incs):
            part = slice(k * n // 2, (k + 1) * n // 2)
            for m in maps.values():
                m.mapvar_update(inc)
                m.fuse(idx[part], h[part], v[part])
        g_e, o_e = maps["hip"].layer("elevation"), maps["oracle"].layer("elevation")
        g_s, o_s = maps["hip"].layer("variance"), maps["oracle"].layer("variance")
        for name, g, o in (("elevation", g_e, o_e), ("variance", g_s, o_s)):   # NaNs must match as NaNs, -0.0 as -0.0, everything else bit for bit
            assert np.array_equal(np.isnan(g), np.isnan(o)), (what, rep, name, "NaN pattern")
            ok = ~np.isnan(o)
            bad = np.flatnonzero(g[ok].view(np.uint32) != o[ok].view(np.uint32))
            assert bad.size == 0, (what, rep, name, bad.size, g[ok][bad[:5]], o[ok][bad[:5]])
    maps["hip"].close()


@pytest.mark.parametrize("thr", [5.0, 2.5, 0.75])
def test_mahalanobis_decision_at_the_threshold(oracle_mod, thr):
    """GPU:502-504: m = |h - e| / sqrt(s) > threshold.  The device takes the decision from a fast estimate and
    replays the reference expression only inside a band around the threshold: sweep the second record of many
    cells ulp by ulp across m == threshold (both signs of h - e) and require the oracle's result everywhere."""
    L = 96
    rng = np.random.default_rng(17)
    n = L * L
    e0 = rng.uniform(-2, 2, n).astype(F32)
    s0 = (10.0 ** rng.uniform(-4, 0, n)).astype(F32)
    sf = np.maximum(s0, F32(1e-4))
    sign = np.where(np.arange(n) % 2 == 0, 1.0, -1.0).astype(F32)
    h1 = (e0 + sign * F32(thr) * np.sqrt(sf, dtype=F32)).astype(F32)
    k = (np.arange(n) % 129) - 64                                      # -64 .. 64 ulps around the estimate of the crossing
    h1 = (h1.view(np.int32) + k.astype(np.int32)).view(F32)
    v1 = (10.0 ** rng.uniform(-4, -1, n)).astype(F32)
    idx = np.arange(n, dtype=np.int32)
    gpu, ref = make_pair(oracle_mod, L, 0.1, mahalanobis_threshold=thr)
    for m in (gpu, ref):
        m.fuse(idx, e0, s0)                                            # first record of a cell: replace
        m.fuse(idx, h1, v1)                                            # the one at the threshold
    assert_maps_match(gpu, ref)
    e1 = ref.layer("elevation").ravel()
    outlier = (e1 == e0) | (e1 == h1)                                  # ignored / replaced; anything else was fused
    assert 0.2 < outlier.mean() < 0.8                                  # the sweep straddles the crossing
    for m in (gpu, ref):
        m.fuse(idx, (h1 + F32(0.01)).astype(F32), v1)                  # and once more from the state that decision left
    assert_maps_match(gpu, ref)


def test_process_then_fuse_equals_add(oracle_mod):
    wl = synth.config_c2()
    a = ElevationMap(wl.length, wl.resolution); b = ElevationMap(wl.length, wl.resolution)
    c = wl.clouds[0]
    a.add(wl.frames[0], c)
    out = b.process_points(wl.frames[0], c[:, 0], c[:, 1], c[:, 2])
    b.fuse(out["index"], out["height"], out["var"])
    assert np.array_equal(a.layer("elevation"), b.layer("elevation")) and np.array_equal(a.layer("variance"), b.layer("variance"))


# ---- Move + Mapvar_update interleavings -------------------------------------------------------------------------
def test_move_and_add_sequence(oracle_mod):
    gpu, ref = make_pair(oracle_mod, 120, 0.1)       # simple_demo_map.yaml geometry
    rng = np.random.default_rng(17)
    pos = np.zeros(3)
    for k in range(12):
        pos = pos + np.array([rng.normal(0.4, 0.3), rng.normal(-0.2, 0.3), 0.0])
        if k == 7:
            pos = pos + np.array([30.0, 0, 0])       # jump > map size: whole-map clear
        rg, ro = gpu.move(pos), ref.move(pos)
        for x, y in zip(rg, ro):
            assert np.array_equal(x, y)
        T = synth.pose_matrix(pos[0], pos[1], 0.8, 0.2 * k, 0.01, 0.0)
        c = synth.lidar_sweep(np.random.default_rng(100 + k), T, beams=16, azimuth_steps=512, max_range=30.0)
        f = synth._frame_for(T, SensorModel.velodyne())
        gpu.mapvar_update(2e-6 * k); ref.mapvar_update(2e-6 * k)
        gpu.add(f, c); ref.add(f, c)
        assert_maps_match(gpu, ref)


def test_mapvar_update_queue_semantics(oracle_mod):
    gpu, ref = make_pair(oracle_mod, 64, 0.1)
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    c = synth.random_cloud(4, 5000, 3.0)
    for m in (gpu, ref):
        m.mapvar_update(0.5)                          # before any fuse: no-op (all -10)
        m.add(f, c)
        for u in (1e-5, 2e-5, 3e-5, 4e-5, 5e-5, 6e-5):   # more than the 4-deep queue
            m.mapvar_update(u)
    assert_maps_match(gpu, ref)                       # get_layer flushes the queue
    for m in (gpu, ref):
        m.mapvar_update(-3e-4)                        # pushes variances under the floor ...
        m.add(f, c[:10])                              # ... the next Fuse repairs every cell
    assert_maps_match(gpu, ref)
    for m in (gpu, ref):
        m.mapvar_update(1e-5); m.move([0.5, 0.0, 0.0]); m.mapvar_update(2e-5)
        m.add(f, c)
    assert_maps_match(gpu, ref)


def test_set_get_layers_and_gridmap_layout(oracle_mod):
    from gem_amd import _lib
    gpu, ref = make_pair(oracle_mod, 48, 0.1)
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    c = synth.random_cloud(8, 3000, 2.0)
    gpu.add(f, c); ref.add(f, c)
    e = gpu.layer("elevation")
    # grid_map layout (EM.cpp:98-111: matrix(index_x, index_y) = flat[index_x * L + index_y], Eigen column-major, NaN for cells
    # without elevation) against the ORACLE's layers, for a float and an int layer, and the raw column-major bytes
    eo, vo = ref.layer("elevation"), ref.layer("variance")
    for name, want in (("elevation", eo), ("variance", vo)):
        gm = gpu.layer(name, layout=_lib.LAYOUT_GRIDMAP_COLMAJOR_NAN)          # viewed as [row, col]
        assert np.array_equal(np.isnan(gm), eo == -10), name
        assert np.array_equal(gm[eo != -10], want[eo != -10]), name
        assert gm.base is not None and gm.base.flags["C_CONTIGUOUS"]           # the buffer itself is column-major: base[col, row]
        assert np.array_equal(np.nan_to_num(gm.base, nan=-10.0)[3], np.where(eo == -10, F32(-10), want)[:, 3]), name
    new_e = np.where(e == -10, e, e + 1.0).astype(F32)
    gpu.set_layer("elevation", new_e); ref.set_layer("elevation", new_e)
    gpu.add(f, c); ref.add(f, c)
    assert_maps_match(gpu, ref)


# ---- batched sweeps (BASELINE config 4) -------------------------------------------------------------------------------
def test_add_batch_parity(oracle_mod):
    import torch
    wl = synth.config_c4(n_sweeps=5)
    gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
    sizes = [131072, 100000, 7, 0, 131072]
    clouds = [c[:n] for c, n in zip(wl.clouds, sizes)]
    offsets = np.concatenate([[0], np.cumsum(sizes)])
    d = torch.from_numpy(np.concatenate(clouds, 0)).to("cuda:0")
    gpu.mapvar_update(7e-6); ref.mapvar_update(7e-6)
    gpu.add_batch(wl.frames, d, offsets, wl.var_updates)
    for k in range(5):
        ref.mapvar_update(wl.var_updates[k]); ref.add(wl.frames[k], clouds[k])
    assert_maps_match(gpu, ref)
    # same thing as individual device-resident adds
    gpu2 = ElevationMap(wl.length, wl.resolution)
    gpu2.mapvar_update(7e-6)
    for k in range(5):
        gpu2.mapvar_update(wl.var_updates[k]); gpu2.add(wl.frames[k], d[offsets[k]:offsets[k + 1]])
    assert np.array_equal(gpu2.layer("elevation"), gpu.layer("elevation")) and np.array_equal(gpu2.layer("variance"), gpu.layer("variance"))


# ---- row strips (multi-GPU tiling on one device) ----------------------------------------------------------------------
def test_row_strips_compose(oracle_mod):
    wl = synth.config_c2()
    L = wl.length
    full, ref = make_pair(oracle_mod, L, wl.resolution)
    full.add(wl.frames[0], wl.clouds[0]); ref.add(wl.frames[0], wl.clouds[0])
    e = np.full((L, L), -10, F32); v = np.full((L, L), -10, F32)
    bounds = [0, 150, 301, 450, 600]
    for r0, r1 in zip(bounds[:-1], bounds[1:]):
        part = ElevationMap(L, wl.resolution, strip=(r0, r1 - r0))
        part.add(wl.frames[0], wl.clouds[0])
        e[r0:r1] = part.layer("elevation")[r0:r1]; v[r0:r1] = part.layer("variance")[r0:r1]
        pe = part.layer("elevation"); pe[r0:r1] = -10
        assert np.all(pe == -10)                     # nothing outside the owned strip is written
    assert np.array_equal(e, ref.layer("elevation")) and np.array_equal(v, ref.layer("variance"))


# ---- multi-GPU plumbing on one device --------------------------------------------------------------------------------
def test_rccl_single_rank_allgather(oracle_mod):
    from gem_amd.tiling import TiledElevationMap
    wl = synth.config_c1()
    uid = ElevationMap.comm_unique_id()
    assert len(uid) == 128
    tm = TiledElevationMap(wl.length, wl.resolution, 0, 1, exchange="rccl", unique_id=uid)
    ref = oracle_mod.OracleMap(wl.length, wl.resolution)
    tm.add(wl.frames[0], wl.clouds[0]); ref.add(wl.frames[0], wl.clouds[0])
    tm.allgather(with_attributes=True)
    assert np.array_equal(tm.layer("elevation"), ref.layer("elevation"))
    assert np.array_equal(tm.layer("variance"), ref.layer("variance"))


def test_torch_exchange_aliases_device_layers(oracle_mod):
    import os
    import torch
    import torch.distributed as dist
    from gem_amd.tiling import TiledElevationMap
    wl = synth.config_c1()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        tm = TiledElevationMap(wl.length, wl.resolution, 0, 1, exchange="torch")
        ref = oracle_mod.OracleMap(wl.length, wl.resolution)
        tm.add(wl.frames[0], wl.clouds[0]); ref.add(wl.frames[0], wl.clouds[0])
        tm.allgather()
        e, v = tm.layer_tensors()
        torch.cuda.synchronize()
        assert e.is_cuda and e.data_ptr() == tm.map.layer_device_ptr("elevation")      # zero-copy alias
        assert np.array_equal(e.cpu().numpy(), ref.layer("elevation")) and np.array_equal(v.cpu().numpy(), ref.layer("variance"))
    finally:
        dist.destroy_process_group()


# ---- kernel variants: every fuse kernel / tile size must give the same bits ---------------------------------------------
@pytest.mark.parametrize("variant,ts", [(12, 4), (12, 5), (11, 5), (10, 5)])
def test_fuse_kernel_variants(oracle_mod, monkeypatch, variant, ts):
    monkeypatch.setattr(ElevationMap, "default_debug", {"fuse_variant": variant, "tile_shift": ts})
    wl = synth.config_c4(n_sweeps=2)
    gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
    for k in range(2):                                  # LiDAR: the rank-row fast path
        gpu.mapvar_update(wl.var_updates[k]); ref.mapvar_update(wl.var_updates[k])
        gpu.add(wl.frames[k], wl.clouds[k]); ref.add(wl.frames[k], wl.clouds[k])
    assert_maps_match(gpu, ref)
    gpu, ref = make_pair(oracle_mod, 40, 0.1)           # > 7 points per cell and batch: the generic path, several batches
    c = synth.random_cloud(22, 60_000, 2.2, z_sigma=0.05)
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    gpu.add(f, c); ref.add(f, c)
    assert_maps_match(gpu, ref)


def test_mixed_fast_and_generic_batches(oracle_mod):
    # a sparse cloud (fast path) with one dense cluster (generic path) in the same tile, twice
    gpu, ref = make_pair(oracle_mod, 64, 0.1)
    rng = np.random.default_rng(9)
    sparse = synth.random_cloud(31, 3000, 3.0, dup_fraction=0.0)
    dense = np.zeros((500, 4), F32); dense[:, 0] = 0.52 + rng.uniform(0, 0.2, 500); dense[:, 1] = -0.33 + rng.uniform(0, 0.2, 500)
    dense[:, 2] = rng.normal(0, 0.03, 500)
    c = np.concatenate([sparse[:1500], dense, sparse[1500:]], 0)
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    for _ in range(2):
        gpu.add(f, c); ref.add(f, c)
        assert_maps_match(gpu, ref)


def _laser_frames():
    """Laser frames either side of every condition under which the library takes the short form of the laser variance
    (gem_capi_core.cpp: fill_frame): the map it writes must be the oracle's whichever form ran."""
    base = synth.pose_matrix(0.3, -0.2, 0.9, 0.4, 0.05, -0.03)
    out = {}
    out["plain"] = synth._frame_for(base, SensorModel.velodyne())
    f = synth._frame_for(base, SensorModel.velodyne()); f.rotation_variance = np.full((3, 3), -0.0, F32); out["minus_zero_Q"] = f
    f = synth._frame_for(base, SensorModel.velodyne()); q = np.zeros((3, 3), F32); q[0, 0] = q[1, 1] = 1e-4; q[0, 2] = q[2, 0] = 2e-5
    f.rotation_variance = q; out["rotation_variance"] = f
    f = synth._frame_for(base, SensorModel.velodyne()); f.sensor_jacobian = np.array([0.3, -0.2, 0.0], F32); out["Js2_zero"] = f
    f = synth._frame_for(base, SensorModel.velodyne()); f.sensor_jacobian = np.array([0.0, 0.0, -1.0], F32); out["Js_negative"] = f
    out["tiny_min_radius"] = synth._frame_for(base, SensorModel(SensorModel.velodyne().kind, (1e-24, 0.0006, 0.0015), 0.8, -5.0))
    out["no_beam_growth"] = synth._frame_for(base, SensorModel(SensorModel.velodyne().kind, (0.018, 0.0, 0.0), 0.8, -5.0))
    f = synth._frame_for(base, SensorModel.velodyne()); T = f.T.copy(); T[:3, :3] *= 1.03; f.T = T; out["scaled_T"] = f
    f = synth._frame_for(base, SensorModel.velodyne()); T = f.T.copy(); T[0, :3] += np.array([0.0, 0.05, 0.0], F32); f.T = T; out["sheared_T"] = f
    f = synth._frame_for(base, SensorModel.velodyne()); f.P_mul_C_BM_T = np.array([0.0, 0.0, 3e6], F32); out["huge_P"] = f
    f = synth._frame_for(base, SensorModel.velodyne()); f.lower, f.upper = -np.inf, np.inf; out["open_window"] = f
    return out


@pytest.mark.parametrize("which", ["plain", "minus_zero_Q", "rotation_variance", "Js2_zero", "Js_negative", "tiny_min_radius",
                                   "no_beam_growth", "scaled_T", "sheared_T", "huge_P", "open_window"])
def test_laser_variance_short_form_qualification(oracle_mod, which):
    f = _laser_frames()[which]
    gpu, ref = make_pair(oracle_mod, 96, 0.1)
    c = synth.random_cloud(77, 6000, 4.0)
    c[::97, 2] += 30.0                                       # some points outside the height window
    for _ in range(2):
        gpu.add(f, c); ref.add(f, c)
        assert_maps_match(gpu, ref)


def test_batches_mixing_short_form_and_generic_frames(oracle_mod):
    import torch
    fr = _laser_frames()
    frames = [fr["plain"], fr["rotation_variance"], fr["plain"], fr["scaled_T"], fr["Js_negative"]]
    gpu, ref = make_pair(oracle_mod, 96, 0.1)
    clouds = [synth.random_cloud(200 + k, 3000 + 500 * k, 4.0) for k in range(len(frames))]
    offsets = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])])
    d = torch.from_numpy(np.concatenate(clouds, 0)).to("cuda:0")
    incs = [1e-5 * (k + 1) for k in range(len(frames))]
    for rep in range(2):
        gpu.add_batch(frames, d, offsets, incs)
        for k, f in enumerate(frames):
            ref.mapvar_update(incs[k]); ref.add(f, clouds[k])
        assert_maps_match(gpu, ref)
    # a batch whose frames all qualify, next to the same one with the short form switched off
    gpu2 = ElevationMap(96, 0.1, debug={"fast_laser": 0}); gpu3 = ElevationMap(96, 0.1)
    for m in (gpu2, gpu3):
        m.add_batch([fr["plain"], fr["open_window"], fr["minus_zero_Q"]], d[:offsets[3]], offsets[:4], incs[:3])
    for name in ("elevation", "variance"):
        assert np.array_equal(gpu2.layer(name), gpu3.layer(name))


# ---- AoS ingest: the PCL point structs as they are --------------------------------------------------------
def test_add_aos_equals_add_on_unpacked_arrays(oracle_mod):
    # PointXYZRGBICT (PointXYZRGBICT.hpp:28-46): x y z pad | b g r a | covariance | intensity | travers = 32 bytes
    pt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("pad", "<f4"), ("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1"),
                   ("covariance", "<f4"), ("intensity", "<f4"), ("travers", "<f4")])
    assert pt.itemsize == 32
    rng = np.random.default_rng(31)
    n = 50_000
    c = synth.random_cloud(33, n, 5.5)
    pts = np.zeros(n, pt)
    pts["x"], pts["y"], pts["z"], pts["intensity"] = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    pts["pad"], pts["covariance"], pts["travers"] = 1.0, rng.random(n), rng.random(n)          # must be ignored
    for k in "rgb":
        pts[k] = rng.integers(0, 3, n) * 100
    pts["a"] = 255                                                                            # PCL sets alpha; not a colour
    packed = (pts["r"].astype(np.uint32) << 16) | (pts["g"].astype(np.uint32) << 8) | pts["b"].astype(np.uint32)
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    gpu, ref = make_pair(oracle_mod, 100, 0.1)
    for _ in range(2):
        gpu.add_aos(f, pts); ref.add(f, c, rgb=packed)
        assert_maps_match(gpu, ref, layers=("elevation", "variance", "intensity", "color_r", "color_g", "color_b"))
    assert (ref.layer("color_r") != 0).sum() > 100
    # XYZ-only structs (no intensity, no colour), 16-byte step
    xyz = np.zeros((n, 4), F32); xyz[:, :3] = c[:, :3]
    gpu2, ref2 = make_pair(oracle_mod, 100, 0.1)
    c0 = c.copy(); c0[:, 3] = 0
    gpu2.add_aos(f, xyz, off_intensity=-1, off_rgb=-1); ref2.add(f, c0)
    assert_maps_match(gpu2, ref2, layers=("elevation", "variance", "intensity"))
    with pytest.raises(Exception):
        gpu2.add_aos(f, xyz, off_x=2)                         # misaligned field


# ---- the HIP path against the reference's own code (no oracle in between) ---------------------------------------------
def test_hip_path_against_the_compiled_reference(ref_mod):
    """libgem_hip (through the C ABI) vs oracle/_ref/libgem_ref.so, the reference's gpu_process.cu compiled for the CPU:
    Move, then per frame Mapvar_update + Process_points + Fuse on the reference side, gem_mapvar_update + gem_add on ours
    (BASELINE config 1 geometry, the reference's hard-coded reject filter), then the traversability stage."""
    L, res = 200, 0.1
    gpu, ref = ElevationMap(L, res), ref_mod.RefMap(L, res)
    rng = np.random.default_rng(77)
    pos = np.zeros(3)
    for step in range(4):
        pos[:2] += rng.uniform(-1.5, 1.5, 2)
        pg, pr = gpu.move(pos), ref.move(pos)
        f = synth._frame_for(synth.pose_matrix(pos[0] + 0.1, pos[1], 0.55, 0.4 * step, 0.01, -0.02), SensorModel.velodyne())
        f.filter = RejectFilter.reference()
        c = synth.random_cloud(70 + step, 12_000, 9.0, z_sigma=0.08)
        gpu.mapvar_update(1e-5 * step); ref.mapvar_update(1e-5 * step)
        out = ref.process_points(f, c[:, 0], c[:, 1], c[:, 2])
        ref.fuse(out["index"], out["height"], out["var"])
        gpu.add(f, c)
        assert (out["index"] >= 0).mean() > 0.05
        for name in ("elevation", "variance"):
            g, r = gpu.layer(name), ref.layer(name)
            bad = np.flatnonzero(g.ravel() != r.ravel())
            assert bad.size == 0, f"step {step} {name}: {bad.size} cells differ: ours {g.ravel()[bad[:4]]} reference {r.ravel()[bad[:4]]}"
    fg, fr = gpu.map_feature(), ref.map_feature()
    live = ref.layer("elevation") != -10
    assert live.sum() > 2000
    assert np.array_equal(fg["rough"][live], fr["rough"][live])
    for k in ("slope", "traver"):
        assert np.max(np.abs(fg[k][live] - fr[k][live])) <= 1e-4, k          # float trigonometry: implementation-defined last bits


# ---- lowest scan points + visibility clean-up -----------------------------------------------------------
def _walls(rng, m_list, L):
    e = m_list[-1].layer("elevation").copy()
    walls = (rng.random((L, L)) < 0.08) & (e != -10)
    e[walls] += rng.uniform(0.2, 2.5, walls.sum()).astype(F32)
    t = np.where(walls, F32(0.2), F32(0.9)).astype(F32); t[rng.random((L, L)) < 0.05] = F32(0.1)
    for m in m_list:
        m.set_layer("elevation", e); m.set_layer("traver", t)
    return int((e != -10).sum())


@pytest.mark.parametrize("L,res,dense_min", [(100, 0.1, 2048), (75, 0.2, 2048), (100, 0.1, 0)])
def test_lowest_tracking_and_raytracing(oracle_mod, ref_mod, monkeypatch, L, res, dense_min):
    """gem_add with lowest tracking + gem_raytracing against the oracle AND against the reference's own code
    (Process_points + Fuse + Raytracing of the compiled gpu_process.cu); dense_min = 0 sends every tile down the dense path."""
    monkeypatch.setattr(ElevationMap, "default_debug", {"dense_min": dense_min})
    gpu, ora, ref = ElevationMap(L, res), oracle_mod.OracleMap(L, res), ref_mod.RefMap(L, res)
    gpu.set_lowest_tracking(True)
    rng = np.random.default_rng(L)
    deleted = 0
    for step in range(4):
        pos = [float(rng.uniform(-1, 1)) * (step + 1), float(rng.uniform(-1, 1)) * (step + 1), 0.45 + 0.1 * step]
        for m in (gpu, ora, ref):
            m.move(pos)
        f = synth._frame_for(synth.pose_matrix(pos[0], pos[1], pos[2], 0.7 * step, 0.01, 0.02), SensorModel.velodyne())
        f.filter = RejectFilter.reference(); f.lower, f.upper = -3.0, 3.0
        c = synth.random_cloud(200 + step, 30_000, 0.5 * L * res, z_sigma=0.25)
        if step % 2:
            gpu.add(f, c)
        else:                                                        # the reference-shaped pair of calls
            pp = gpu.process_points(f, c[:, 0], c[:, 1], c[:, 2]); gpu.fuse(pp["index"], pp["height"], pp["var"])
        ora.add(f, c)
        out = ref.process_points(f, c[:, 0], c[:, 1], c[:, 2]); ref.fuse(out["index"], out["height"], out["var"])
        for name in ("elevation", "variance", "lowest"):
            g = gpu.layer(name)
            assert np.array_equal(g, ora.layer(name)), (step, name, "oracle")
            assert np.array_equal(g, ref.layer(name)), (step, name, "reference")
        before = _walls(rng, [gpu, ora, ref], L)
        gpu.raytracing(); ora.raytracing(); ref.raytracing()
        for name in ("elevation", "variance", "lowest"):
            g = gpu.layer(name)
            assert np.array_equal(g, ora.layer(name)), (step, name, "oracle, after raytracing")
            assert np.array_equal(g, ref.layer(name)), (step, name, "reference, after raytracing")
        deleted += before - int((gpu.layer("elevation") != -10).sum())
    assert deleted > 20


@pytest.mark.parametrize("L,res,pos", [(300, 0.05, (3.37, -2.11, 0.62)), (251, 0.1, (-7.05, 4.4, 1.3)), (64, 0.1, (0.0, 0.0, 0.4)), (33, 0.2, (1.0, 1.0, 0.9)),
                                       (1201, 0.05, (11.3, -17.9, 0.8))])
def test_raytracing_walks_split_over_lanes(oracle_mod, L, res, pos):
    """k_raytracing hands one walk to several lanes, each starting in the middle of it (closed-form state of the border-distance
    merge): every lane count / look-ahead depth must leave the map the oracle's one-thread-per-cell walk leaves.  Dense random
    layers: long walks in all directions, ties on the diagonals, walks that leave through either axis."""
    rng = np.random.default_rng(L)
    gpu, ora = make_pair(oracle_mod, L, res)
    for m in (gpu, ora):
        m.move(pos)                                                       # circular-buffer start off zero, the sensor height
    for trial in range(2):
        e = rng.normal(0.0, 0.4, (L, L)).astype(F32); e[rng.random((L, L)) < 0.15] = F32(-10)
        v = rng.uniform(1e-4, 4e-3, (L, L)).astype(F32)
        t = rng.uniform(0.0, 1.0, (L, L)).astype(F32)
        t[rng.random((L, L)) < 0.3] = F32(0.1)
        low = np.full((L, L), 10.0, F32)
        k = rng.random((L, L)) < (0.5 if trial else 0.05)
        low[k] = rng.normal(-0.3 if trial else 0.2, 0.3, k.sum()).astype(F32)
        results = []
        for lanes, depth in [(0, 0), (1, 4), (1, 8), (4, 4), (4, 8), (8, 4), (8, 8), (16, 4), (16, 8)]:
            m = ora if lanes == 0 else gpu
            if lanes:
                gpu.debug_set("ray_lanes", lanes); gpu.debug_set("ray_depth", depth)
            for name, a in (("elevation", e), ("variance", v), ("traver", t), ("lowest", low)):
                m.set_layer(name, a)
            m.raytracing()
            results.append((lanes, depth, m.layer("elevation"), m.layer("lowest")))
        want = results[0]
        deleted = int(((want[2] == -10) & (e != -10)).sum())
        assert deleted > 5, "the case deletes nothing"
        assert int(((want[2] != -10) & (t < 0.7)).sum()) > L, "the case deletes everything"
        for lanes, depth, ge, gl in results[1:]:
            bad = np.flatnonzero(ge.ravel() != want[2].ravel())
            assert bad.size == 0, f"lanes {lanes} depth {depth}: {bad.size} cells differ, first {bad[:6]}"
            assert np.array_equal(gl, want[3])


def test_lowest_sees_the_minus_one_sentinel(oracle_mod):
    # a point whose height is exactly -1 is skipped by the fusion (GPU:482) but still counts as a scan point (GPU:430-439)
    L = 40
    gpu, ora = make_pair(oracle_mod, L, 0.1)
    gpu.set_lowest_tracking(True)
    f = synth._frame_for(np.eye(4), SensorModel.velodyne()); f.lower, f.upper = -5.0, 5.0
    rng = np.random.default_rng(4)
    c = synth.random_cloud(12, 20_000, 1.9, z_sigma=0.3)
    c[rng.integers(0, c.shape[0], 4000), 2] = -1.0
    for _ in range(2):
        gpu.add(f, c); ora.add(f, c)
        assert_maps_match(gpu, ora, layers=("elevation", "variance", "lowest"))


def test_lowest_tracking_batched_and_off(oracle_mod):
    import torch
    wl = synth.config_c4(n_sweeps=3)
    gpu, ora = make_pair(oracle_mod, wl.length, wl.resolution)
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    cat = torch.from_numpy(np.concatenate(wl.clouds)).cuda()
    gpu.add_batch(wl.frames, cat, off, wl.var_updates)                 # tracking off: the layer keeps its initial value
    assert np.all(gpu.layer("lowest") == 100.0)
    gpu.set_lowest_tracking(True)
    gpu.add_batch(wl.frames, cat, off, wl.var_updates)
    for rep in range(2):
        for k in range(3):
            ora.mapvar_update(wl.var_updates[k]); ora.add(wl.frames[k], wl.clouds[k])
    assert_maps_match(gpu, ora)
    lo = ora.layer("lowest")
    # the oracle tracked both rounds, the device only the second: the per-cell recurrence (min, then + 3 var when the
    # point IS the minimum) is not idempotent, so compare against an oracle that only tracked the second round
    ora2 = oracle_mod.OracleMap(wl.length, wl.resolution)
    for k in range(3):
        ora2.add(wl.frames[k], wl.clouds[k])
    assert np.array_equal(gpu.layer("lowest"), ora2.layer("lowest"))
    assert (lo != 100).mean() > 0.1


# ---- dense tiles (k_fuse_list hands the tile to its second copy, which counting-sorts the sweep's records by cell) ----------
@pytest.mark.parametrize("dense_min", [0, 300])
def test_dense_tile_path(oracle_mod, monkeypatch, dense_min):
    """The dense_min knob (gem_debug_set) lowers the records-per-(tile, sweep) threshold of the dense path, so that ordinary clouds take it:
    mixed dense / LiDAR tiles, several sweeps per tile with variance increments in between, colour attributes,
    the strip clipping, a single cell fed by a whole sweep."""
    import torch
    monkeypatch.setattr(ElevationMap, "default_debug", {"dense_min": dense_min})
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    # (a) collisions + attributes, twice (the second call fuses into non-empty cells)
    gpu, ref = make_pair(oracle_mod, 100, 0.1)
    rng = np.random.default_rng(9)
    c = synth.random_cloud(9, 60000, 3.5)
    rgb = (rng.integers(0, 3, (60000, 3)) * 100).astype(np.uint32)
    packed = (rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2]
    for _ in range(2):
        gpu.add(f, c, rgb=packed); ref.add(f, c, rgb=packed)
        assert_maps_match(gpu, ref, layers=("elevation", "variance", "intensity", "color_r", "color_g", "color_b"))
    # (b) a batch: sweeps of very different density over the same tiles, Mapvar_update between them
    wl = synth.config_c4(n_sweeps=4)
    gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
    clouds = [wl.clouds[0], wl.clouds[1][:3000], wl.clouds[2], wl.clouds[3][:40000]]
    off = np.concatenate([[0], np.cumsum([x.shape[0] for x in clouds])])
    cat = torch.from_numpy(np.concatenate(clouds)).cuda()
    for _ in range(2):
        gpu.add_batch(wl.frames, cat, off, wl.var_updates)
        for k in range(4):
            ref.mapvar_update(wl.var_updates[k]); ref.add(wl.frames[k], clouds[k])
        assert_maps_match(gpu, ref)
    # (c) one cell fed by 10 000 points, inside a 300 000-point cloud that is cut into sweeps
    gpu, ref = make_pair(oracle_mod, 64, 0.1)
    big = synth.random_cloud(41, 300_000, 3.0)
    big[100_000:110_000, 0] = 0.31 + rng.uniform(0, 0.05, 10_000).astype(F32)
    big[100_000:110_000, 1] = -0.72 + rng.uniform(0, 0.05, 10_000).astype(F32)
    gpu.add(f, big); ref.add(f, big)
    assert_maps_match(gpu, ref)
    # (d) fuse() with host arrays (records come from precomputed indices) incl. the h == -1 sentinel
    gpu, ref = make_pair(oracle_mod, 50, 0.1)
    n = 30000
    idx = rng.integers(-1, 2500 + 5, n).astype(np.int32)
    h = rng.normal(0, 0.2, n).astype(F32); h[rng.integers(0, n, 50)] = -1.0
    v = rng.uniform(1e-6, 2e-3, n).astype(F32)
    R, G, B = (rng.integers(0, 3, n).astype(np.int32) * 90 for _ in range(3))
    I = rng.integers(0, 2, n).astype(F32)
    for _ in range(2):
        gpu.fuse(idx, h, v, R, G, B, I); ref.fuse(idx, h, v, R, G, B, I)
        assert_maps_match(gpu, ref, layers=("elevation", "variance", "intensity", "color_r", "color_g", "color_b"))


def test_dense_depth_image_default_threshold(oracle_mod):
    # BASELINE config 3 (640x480 depth image) takes the dense path with the default threshold; fused twice
    wl = synth.config_c3()
    gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
    for m in (gpu, ref):
        m.move(wl.map_position)
    for _ in range(2):
        gpu.add(wl.frames[0], wl.clouds[0]); ref.add(wl.frames[0], wl.clouds[0])
        assert_maps_match(gpu, ref)
        gpu.mapvar_update(3e-5); ref.mapvar_update(3e-5)


# ---- accumulate mode: sweeps without variance increments between them share batches ---------------------------------------
@pytest.mark.parametrize("dense_min", [2048, 150])
def test_batch_without_increments_accumulates_across_sweeps(oracle_mod, monkeypatch, dense_min):
    """Sweeps of very different sizes over the same tiles: a few points (appended), a flood (does not fit: flush, then its own
    batches), tiles that turn dense in the middle of a collected batch, an empty sweep, a tail that is flushed at the end of
    the pass -- and the same call with per-sweep counting on, which switches the accumulation off."""
    import torch
    monkeypatch.setattr(ElevationMap, "default_debug", {"dense_min": dense_min})
    L, res = 96, 0.1
    rng = np.random.default_rng(21)
    f0 = synth._frame_for(synth.pose_matrix(0.1, -0.2, 0.0, yaw=0.3), SensorModel.velodyne())
    f1 = synth._frame_for(synth.pose_matrix(-0.3, 0.1, 0.05, yaw=-0.7), SensorModel.velodyne())
    sizes = [40, 300, 25, 9000, 10, 0, 700, 60, 30_000, 15, 80, 5]
    clouds = [synth.random_cloud(300 + k, n, 5.0, z_sigma=0.2, dup_fraction=0.3) for k, n in enumerate(sizes)]
    clouds[6][:, :2] = clouds[6][:, :2] * 0.1 + 1.0                       # a cluster: a few cells get hundreds of records
    frames = [f0 if k % 3 else f1 for k in range(len(sizes))]
    off = np.concatenate([[0], np.cumsum(sizes)])
    cat = torch.from_numpy(np.concatenate(clouds)).cuda()
    gpu, ref = make_pair(oracle_mod, L, res)
    gpu2 = ElevationMap(L, res)
    gpu2.set_counting(True)
    for rep in range(2):
        gpu.mapvar_update(2e-5); gpu2.mapvar_update(2e-5); ref.mapvar_update(2e-5)      # queued: applied before the first record
        gpu.add_batch(frames, cat, off, None)
        gpu2.add_batch(frames, cat, off, None)
        for k in range(len(sizes)):
            ref.add(frames[k], clouds[k])
        assert_maps_match(gpu, ref)
        assert_maps_match(gpu2, ref)


# ---- aggregated cloud into the big map (BASELINE config 5, reduced) -----------------------------------------------------
def test_c5_aggregated_batch_parity(oracle_mod):
    import torch
    wl = synth.config_c5(n_points=600_000)
    gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    d = torch.from_numpy(np.concatenate(wl.clouds, 0)).to("cuda:0")
    gpu.add_batch(wl.frames, d, off, None)
    for fr, c in zip(wl.frames, wl.clouds):
        ref.add(fr, c)
    assert_maps_match(gpu, ref)
    # one of eight row strips (rank 3 of the 8-GPU tiling) equals the same rows of the full map
    L = wl.length
    part = ElevationMap(L, wl.resolution, strip=(900, 300))
    part.add_batch(wl.frames, d, off, None)
    assert np.array_equal(part.layer("elevation")[900:1200], ref.layer("elevation")[900:1200])
    assert np.array_equal(part.layer("variance")[900:1200], ref.layer("variance")[900:1200])


def test_batch_tail_descriptor_crossing_a_batch_boundary(oracle_mod, monkeypatch):
    # regression: a tile whose last descriptor merely extends past a multiple of the batch quantum
    # (an empty trailing batch) -- met by sweep 7 of the C4 series with 2048-record batches
    import torch
    monkeypatch.setattr(ElevationMap, "default_debug", {"fuse_variant": 12, "tile_shift": 5})
    wl = synth.config_c4(n_sweeps=8)
    gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    d = torch.from_numpy(np.concatenate(wl.clouds, 0)).to("cuda:0")
    gpu.add_batch(wl.frames, d, off, wl.var_updates)
    for k in range(8):
        ref.mapvar_update(wl.var_updates[k]); ref.add(wl.frames[k], wl.clouds[k])
    assert_maps_match(gpu, ref)


def test_big_single_cloud_is_cut_into_sweeps(oracle_mod):
    # > 131072 points in ONE call: processed internally as a batch of sweeps sharing the frame; the
    # stereo model uses the point's index in the cloud (pixel row / column), which must survive the cut
    gpu, ref = make_pair(oracle_mod, 400, 0.025)
    wl = synth.config_c3()
    f = wl.frames[0]
    f.model = SensorModel(2, (0.1, 0.001, 380.0, 1.0, 0.002, 0.001, 30.0), original_width=640)
    f.model.ignore_points_above, f.model.ignore_points_below = float("inf"), float("-inf")
    c = wl.clouds[0]
    assert c.shape[0] > 2 * 131072
    gpu.move(wl.map_position); ref.move(wl.map_position)
    gpu.add(f, c); ref.add(f, c)                       # no orig_index: the index in the cloud is used
    assert_maps_match(gpu, ref, exact=False)           # stereo variance goes through double sqrt (last-ulp differences)
    gpu2, ref2 = make_pair(oracle_mod, 64, 0.1)        # laser model, 300k points in one call, several sweeps per tile
    c2 = synth.random_cloud(77, 300_000, 3.0, z_sigma=0.05)
    f2 = synth._frame_for(np.eye(4), SensorModel.velodyne())
    gpu2.add(f2, c2); ref2.add(f2, c2)
    assert_maps_match(gpu2, ref2)


# ---- the one-launch-per-frame stream (k_frame: fuse of the previous sweep + binning of the new one) -----------------------
def test_device_sweep_stream(oracle_mod):
    import torch
    wl = synth.config_c4(n_sweeps=6)
    gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
    d = [torch.from_numpy(c).to("cuda:0") for c in wl.clouds]
    gpu.set_timing(True)
    for k in range(6):                                   # add, add, ... : the fuse of frame k runs inside the launch of frame k+1
        if k == 3:
            gpu.mapvar_update(2e-5); ref.mapvar_update(2e-5)
        gpu.add(wl.frames[k], d[k]); ref.add(wl.frames[k], wl.clouds[k])
        if k in (1, 4):
            assert_maps_match(gpu, ref)                  # observing the map flushes the pending fuse
    assert_maps_match(gpu, ref)
    if not ElevationMap.base_debug:
        assert gpu.stats()["launches_frame"] >= 3        # the merged kernel is what ran
    # dense clusters through the device path: more than 7 records per cell and batch
    gpu, ref = make_pair(oracle_mod, 40, 0.1)
    c = synth.random_cloud(23, 90_000, 2.2, z_sigma=0.05)
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    for _ in range(2):
        gpu.add(f, torch.from_numpy(c).to("cuda:0")); ref.add(f, c)
    assert_maps_match(gpu, ref)


# ---- the host-pointer entry points with their uploads left in flight (round 5) ---------------------------------------------------
def test_add_batch_from_host_arrays(oracle_mod):
    """gem_add_batch: the sweeps as separate HOST arrays -- ragged, one empty, one too short for the staging buffer --
    against the oracle and against the device-resident batch."""
    import torch
    wl = synth.config_c4(n_sweeps=5)
    gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
    sizes = [131072, 100003, 7, 0, 131072]
    clouds = [np.ascontiguousarray(c[:n]) for c, n in zip(wl.clouds, sizes)]
    gpu.mapvar_update(7e-6); ref.mapvar_update(7e-6)
    gpu.add_batch_host(wl.frames, clouds, wl.var_updates)
    for k in range(5):
        ref.mapvar_update(wl.var_updates[k]); ref.add(wl.frames[k], clouds[k])
    assert_maps_match(gpu, ref)
    dev = ElevationMap(wl.length, wl.resolution)
    dev.mapvar_update(7e-6)
    dev.add_batch(wl.frames, torch.from_numpy(np.concatenate(clouds, 0)).to("cuda:0"), np.concatenate([[0], np.cumsum(sizes)]), wl.var_updates)
    assert np.array_equal(dev.layer("elevation"), gpu.layer("elevation")) and np.array_equal(dev.layer("variance"), gpu.layer("variance"))
    # twice in a row (the second call's copies go into the other half of the staging buffer while the first's DMA may still run),
    # then a single-sweep batch with its increment
    gpu.add_batch_host(wl.frames, clouds, wl.var_updates); gpu.add_batch_host(wl.frames[:1], clouds[:1], wl.var_updates[:1])
    for k in list(range(5)) + [0]:
        ref.mapvar_update(wl.var_updates[k]); ref.add(wl.frames[k], clouds[k])
    assert_maps_match(gpu, ref)


def test_stream_of_host_sweeps_with_the_uploads_in_flight(oracle_mod):
    """gem_add from host arrays, call after call without a synchronisation in between: every call returns with its DMA in flight and
    the next one fills the other half of the staging buffer; clouds of different sizes (a later, larger one makes the buffer grow;
    small ones take the runtime's path), a caller that overwrites its array right after the call, map reads in between."""
    wl = synth.config_c4(n_sweeps=8)
    gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
    sizes = [40000, 131072, 9000, 131072, 65536, 131072, 131072, 120001]
    buf = np.empty((131072, 4), np.float32)
    for k in range(8):
        n = sizes[k]
        buf[:n] = wl.clouds[k][:n]
        if k == 5:
            gpu.mapvar_update(3e-5); ref.mapvar_update(3e-5)
        gpu.add(wl.frames[k], buf[:n]); ref.add(wl.frames[k], wl.clouds[k][:n])
        buf[:n] = np.nan                                                 # the caller's array is its own again when the call returns
        if k in (2, 6):
            assert_maps_match(gpu, ref)
    assert_maps_match(gpu, ref)


# ---- the sorted pipeline's walk left to the next call (round 5: gem_handle::dwalk) ---------------------------------------------
@pytest.mark.parametrize("form", [1, 2])
def test_stream_of_sorted_passes_whose_walks_the_next_call_launches(oracle_mod, form):
    """Device clouds through the OVERLAPPED sorted pipeline call after call with nothing read in between: every call sorts, leaves its
    walk to its successor and launches its predecessor's -- clouds of different sizes, single clouds and batches, a variance increment,
    a small sweep that takes the tile pipeline, a move, and a host array (never deferred) in between; the map is compared at the end
    and once in the middle.  The same with the deferral switched off must give the same map."""
    import torch
    wl = synth.config_c4(n_sweeps=8)
    knobs = dict(ElevationMap.base_debug or {}); knobs.update({"overlap_min_points": 1})
    if "sort_form" not in knobs:
        knobs.update({"sort_min_points": 20000, "sort_form": form})
    maps = []
    for defer in (1, 0):
        gpu, ref = make_pair(oracle_mod, wl.length, wl.resolution)
        for k, v in dict(knobs, defer_walk=defer).items():
            gpu.debug_set(k, v)
        d = [torch.from_numpy(c).to("cuda:0") for c in wl.clouds]
        sizes = [131072, 60000, 131072, 25000, 131072, 9000, 131072, 100001]
        for k in range(8):
            n = sizes[k]
            if k == 2:
                gpu.mapvar_update(2e-5); ref.mapvar_update(2e-5)
            if k == 4:
                gpu.move(np.array([0.37, -0.21, 0.0])); ref.move(np.array([0.37, -0.21, 0.0]))
            if k == 6:                                                   # a batch of two sweeps with increments, device-resident
                cat = torch.cat([d[6][:n], d[7][:sizes[7]]])
                gpu.add_batch(wl.frames[6:8], cat, np.array([0, n, n + sizes[7]]), wl.var_updates[6:8])
                for j in (6, 7):
                    ref.mapvar_update(wl.var_updates[j]); ref.add(wl.frames[j], wl.clouds[j][:sizes[j]])
                break
            if k == 3:
                gpu.add(wl.frames[k], wl.clouds[k][:n])                  # host array: sorted or tiled, its walk is never left behind
            else:
                gpu.add(wl.frames[k], d[k][:n].contiguous())
            ref.add(wl.frames[k], wl.clouds[k][:n])
            if k == 1:
                assert_maps_match(gpu, ref)
        assert_maps_match(gpu, ref)
        maps.append((gpu.layer("elevation"), gpu.layer("variance"), gpu.debug_get("walks_left")))
        gpu.close()
    assert np.array_equal(maps[0][0], maps[1][0]) and np.array_equal(maps[0][1], maps[1][1])
    assert maps[0][2] >= 4 and maps[1][2] == 0                           # (walks were left to later calls; switched off: none)
