"""Traversability stage: Map_feature / G_Mapfeature + the Jacobi eigen-solver
(gpu_process.cu:549-670, 66-187, 1256-1302).  CPU tests pin the oracle with hand-derived cases; the GPU
tests compare gem_map_feature with the oracle.

Tolerances.  Every float operation of the stage is replayed in the same order on both sides (FMA
contraction off); the only library calls are sin / cos / atan2 / acos, evaluated in double and rounded to
float on both sides, so results normally agree bit for bit.  A double result within ~1e-16 of a float rounding
boundary may round differently between glibc and the device library; slope = acos(|n_z|) amplifies such a
last-bit difference near n_z = 1 (slope ~ sqrt(2 (1 - n_z))), hence the absolute bounds below.
"""
import numpy as np
import pytest

F32 = np.float32
SLOPE_ABS, ROUGH_ABS, TRAVER_ABS = 2e-3, 1e-6, 2e-3


def terrain(L, res, seed=0, amp=0.4):
    rng = np.random.default_rng(seed)
    x, y = np.meshgrid(np.arange(L) * res, np.arange(L) * res, indexing="ij")
    z = 0.6 * x + 0.15 * y + amp * np.sin(2 * np.pi * x / (9 * res)) * np.cos(2 * np.pi * y / (7 * res)) + rng.normal(0, 0.01, (L, L))
    z[rng.random((L, L)) < 0.15] = -10.0          # holes
    z[:, L // 2: L // 2 + 3] = -10.0             # a gap wider than the 5x5 window's reach on one side
    z[L // 3: L // 3 + 1, :] += 0.5              # a step
    return z.astype(F32)


# ---- oracle known answers (CPU) -----------------------------------------------------------------------------------------
def test_flat_plane_is_fully_traversable(oracle_mod):
    m = oracle_mod.OracleMap(16, 0.1)
    m.set_layer("elevation", np.full((16, 16), 0.25, F32))
    f = m.map_feature()
    assert np.all(f["slope"] == 0) and np.all(f["rough"] == 0) and np.all(f["traver"] == 1.0)
    assert np.array_equal(m.layer("traver"), f["traver"])          # map_traver is updated (GPU:658)


def test_isolated_cells_and_empty_cells(oracle_mod):
    m = oracle_mod.OracleMap(16, 0.1)
    e = np.full((16, 16), -10, F32); e[5, 5] = 1.0; e[5, 6] = 1.1
    m.set_layer("elevation", e)
    f = m.map_feature()
    assert f["traver"][5, 5] == -10 and f["slope"][5, 5] == 0 and f["rough"][5, 5] == 0      # p_n <= 7 (GPU:660-666)
    assert f["traver"][0, 0] == -10 and f["rough"][0, 0] == 0                                  # empty cell: untouched map_traver


def test_tilted_plane_slope_is_the_plane_angle(oracle_mod):
    # z = a*x on a 5x5 patch: cov_xz = a * cov_xx > 0.01 -> the Jacobi loop runs; normal = (-a, 0, 1)/sqrt(1+a^2)
    L, res, a = 32, 0.2, 0.5
    x = (np.arange(L) * res)[:, None] * np.ones((1, L))
    m = oracle_mod.OracleMap(L, res)
    m.set_layer("elevation", (a * x).astype(F32))
    f = m.map_feature()
    inner = f["slope"][4:-4, 4:-4]
    assert np.allclose(inner, np.arctan(a), atol=2e-3)             # Jacobi stops at |off-diagonal| < 0.01
    assert np.allclose(f["rough"][4:-4, 4:-4], 0.0, atol=1e-5)
    assert np.allclose(f["traver"][4:-4, 4:-4], 0.5 * (1 - inner / 0.6) + 0.5, atol=1e-6)


def test_neighbourhood_bounds_follow_the_unrolled_index(oracle_mod):
    # after Move the window is clipped at the map edge in UNROLLED coordinates while storage reads wrap
    L = 24
    m = oracle_mod.OracleMap(L, 0.1)
    m.move(np.array([0.5, -0.3, 0.0], F32))
    e = np.full((L, L), 0.1, F32)
    m.set_layer("elevation", e)
    f = m.map_feature()
    sx, sy = m.pose()[1]
    # unrolled corner cell (0, 0) sits at storage (sx, sy): it sees a 3x3 window = 9 cells > 7 -> valid
    assert f["traver"][sx % L, sy % L] != -10
    # the plane is flat in z but the storage x/y coordinates jump across the wrap: the reference uses storage
    # coordinates for the fit, which only changes the in-plane eigenvectors; slope stays 0
    assert np.all(f["slope"] == 0)


# ---- GPU parity -------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("L,res,seed", [(64, 0.1, 1), (75, 0.2, 2), (200, 0.05, 3)])
def test_map_feature_parity(oracle_mod, L, res, seed):
    from gem_amd import ElevationMap
    gpu, ref = ElevationMap(L, res), oracle_mod.OracleMap(L, res)
    if seed == 2:
        gpu.move(np.array([1.3, -0.7, 0], F32)); ref.move(np.array([1.3, -0.7, 0], F32))
    z = terrain(L, res, seed)
    gpu.set_layer("elevation", z); ref.set_layer("elevation", z)
    g, o = gpu.map_feature(), ref.map_feature()
    assert np.array_equal(g["traver"] == -10, o["traver"] == -10)
    exact = np.mean(g["slope"] == o["slope"])
    assert exact > 0.999, f"only {exact:.4f} of the slopes agree bit for bit"
    assert np.max(np.abs(g["slope"] - o["slope"])) <= SLOPE_ABS
    assert np.max(np.abs(g["rough"] - o["rough"])) <= ROUGH_ABS
    assert np.max(np.abs(g["traver"] - o["traver"])) <= TRAVER_ABS
    assert (o["slope"] > 0.05).mean() > 0.3                          # the Jacobi path is what is being compared
    # the layers are resident: traver / rough / slope can be fetched like any other layer
    assert np.array_equal(gpu.layer("traver"), g["traver"]) and np.array_equal(gpu.layer("slope"), g["slope"])
    assert np.array_equal(gpu.layer("rough"), g["rough"])


@pytest.mark.gpu
def test_map_feature_after_fusion(oracle_mod):
    from gem_amd import ElevationMap, synth
    wl = synth.config_c2()
    gpu, ref = ElevationMap(wl.length, wl.resolution), oracle_mod.OracleMap(wl.length, wl.resolution)
    gpu.add(wl.frames[0], wl.clouds[0]); ref.add(wl.frames[0], wl.clouds[0])
    g, o = gpu.map_feature(), ref.map_feature()
    for k in ("rough", "slope", "traver"):
        assert np.array_equal(g[k], o[k]), k
    assert (o["traver"] != -10).sum() > 10000


# ---- loop-closure re-anchoring: Map_optmove / Map_closeloop (gpu_process.cu:1215-1254) ---------------------------------------
def test_optmove_relabels_the_centre_and_shifts_heights(oracle_mod):
    m = oracle_mod.OracleMap(16, 0.1)
    e = np.full((16, 16), -10, F32); e[3, 4] = 0.5; e[9, 9] = -0.25
    m.set_layer("elevation", e)
    m.move(np.array([0.3, -0.2, 0.0], F32))
    c0, s0 = m.pose()
    aligned = m.map_optmove([c0[0] + 0.234, c0[1] - 0.561], 0.125)
    c1, s1 = m.pose()
    assert tuple(s1) == tuple(s0)                                           # the circular buffer is not shifted
    assert np.allclose(aligned, [c0[0] + 0.2, c0[1] - 0.6], atol=1e-6) and np.allclose(c1, aligned)
    out = m.layer("elevation")
    assert out[3, 4] == F32(0.5) + F32(0.125) and out[9, 9] == F32(-0.25) + F32(0.125) and out[0, 0] == -10


@pytest.mark.gpu
def test_optmove_and_closeloop_parity(oracle_mod):
    from gem_amd import ElevationMap
    L, res = 64, 0.1
    gpu, ref = ElevationMap(L, res), oracle_mod.OracleMap(L, res)
    z = terrain(L, res, 5)
    for m in (gpu, ref):
        m.move(np.array([1.03, -0.46, 0], F32)); m.set_layer("elevation", z)
    a_g, a_o = gpu.map_optmove([1.31, -0.77], 0.071), ref.map_optmove([1.31, -0.77], 0.071)
    assert np.array_equal(a_g, a_o)
    gpu.map_closeloop([0.52, 0.18], -0.033); ref.map_closeloop([0.52, 0.18], -0.033)
    cg, sg = gpu.pose(); co, so = ref.pose()
    assert np.array_equal(np.asarray(cg, F32), np.asarray(co, F32)) and tuple(sg) == tuple(so)
    assert np.array_equal(gpu.layer("elevation"), ref.layer("elevation"))
