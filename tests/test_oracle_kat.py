"""Known-answer tests pinning the CPU oracle.

The reference has no tests or golden vectors (PARITY UNPINNED); these hand-derivable cases are
what pins the oracle's restatement of gpu_process.cu:309-358 (binning), 384-455 (projection,
filter, variance), 477-537 (fusion), 540-547 (variance increment), 1004-1083 (move).
"""
import numpy as np
import pytest

from gem_amd.api import Frame, RejectFilter, SensorModel

F32 = np.float32


def ident_frame(**kw):
    model = kw.pop("model", SensorModel(0, (0.018, 0.0006, 0.0015)))
    return Frame(T=np.eye(4, dtype=F32), lower=kw.pop("lower", -1e9), upper=kw.pop("upper", 1e9), model=model,
                 sensor_jacobian=np.array(kw.pop("Js", [0, 0, 1]), F32), **kw)


def fuse_cells(om, L, idx, h, v, **kw):
    m = om.OracleMap(L, 0.1)
    m.fuse(np.array(idx, np.int32), np.array(h, F32), np.array(v, F32), **kw)
    return m


# ---- C.1 single point into an empty cell -------------------------------------------------------
def test_single_point_empty_cell(oracle_mod):
    m = fuse_cells(oracle_mod, 4, [5], [1.25], [4e-4])
    e, v = m.layer("elevation").ravel(), m.layer("variance").ravel()
    assert e[5] == F32(1.25) and v[5] == F32(4e-4)
    # every other cell: elevation still -10, variance floored from -10 to 1e-4 (GPU:533-534)
    assert np.all(np.delete(e, 5) == F32(-10)) and np.all(np.delete(v, 5) == F32(1e-4))


def test_single_point_small_variance_is_floored(oracle_mod):
    m = fuse_cells(oracle_mod, 4, [0], [0.5], [1e-6])
    assert m.layer("variance").ravel()[0] == F32(1e-4)


# ---- C.2 two points, same cell: 1-D Kalman fuse --------------------------------------------------
def test_two_point_kalman(oracle_mod):
    m = fuse_cells(oracle_mod, 4, [3, 3], [1.0, 1.02], [4e-4, 4e-4])
    e, v = m.layer("elevation").ravel()[3], m.layer("variance").ravel()[3]
    s, vv, h1, h2 = F32(4e-4), F32(4e-4), F32(1.0), F32(1.02)
    assert e == (s * h2 + vv * h1) / (s + vv)         # (GPU:518) in float32
    assert v == (vv * s) / (vv + s)
    assert abs(e - 1.01) < 1e-6 and abs(v - 2e-4) < 1e-9


# ---- C.3 Mahalanobis branch and order dependence ------------------------------------------------
def test_outlier_higher_replaces_lower_ignored(oracle_mod):
    hi = fuse_cells(oracle_mod, 4, [0, 0], [1.0, 1.5], [4e-4, 9e-4])      # 0.5/0.02 = 25 > 5, higher
    assert hi.layer("elevation").ravel()[0] == F32(1.5) and hi.layer("variance").ravel()[0] == F32(9e-4)
    lo = fuse_cells(oracle_mod, 4, [0, 0], [1.0, 0.5], [4e-4, 9e-4])      # lower outlier is ignored
    assert lo.layer("elevation").ravel()[0] == F32(1.0) and lo.layer("variance").ravel()[0] == F32(4e-4)


def test_order_dependence(oracle_mod):
    # 1.0,1.08 fuse (m=4) -> 1.04 / 2e-4, then 1.15 is a >5 sigma outlier and replaces: 1.15
    a = fuse_cells(oracle_mod, 4, [0, 0, 0], [1.0, 1.08, 1.15], [4e-4, 4e-4, 4e-4])
    # 1.0 -> 1.15 replaces (m=7.5), then 1.08 is within 3.5 sigma and fuses: 1.115
    b = fuse_cells(oracle_mod, 4, [0, 0, 0], [1.0, 1.15, 1.08], [4e-4, 4e-4, 4e-4])
    ea, eb = a.layer("elevation").ravel()[0], b.layer("elevation").ravel()[0]
    assert ea == F32(1.15) and abs(eb - 1.115) < 1e-6


def test_threshold_is_strict_greater(oracle_mod):
    # |h-e|/sqrt(s) == 5 exactly -> NOT an outlier (GPU:504 uses '>')
    m = fuse_cells(oracle_mod, 4, [0, 0], [1.0, 1.5], [1e-2, 1e-2])       # sqrt(0.01)=0.1 (float), 0.5/0.1 = 5
    s = F32(1e-2)
    if F32(0.5) / np.sqrt(s) > F32(5):
        pytest.skip("float rounding makes this case > 5")
    assert m.layer("elevation").ravel()[0] == (s * F32(1.5) + s * F32(1.0)) / (s + s)


# ---- C.4 five equal points: variance sequence and the floor --------------------------------------
def test_five_equal_points_floor(oracle_mod):
    seq = []
    for n in range(1, 7):
        m = fuse_cells(oracle_mod, 4, [0] * n, [1.0] * n, [4e-4] * n)
        seq.append(float(m.layer("variance").ravel()[0]))
    v = F32(4e-4)
    exp = [v]
    s = v
    for _ in range(5):
        s = max(s, F32(1e-4))
        s = (v * s) / (v + s)
        exp.append(s)
    exp = [float(max(x, F32(1e-4))) for x in exp]
    assert seq == exp
    assert abs(seq[1] - 2e-4) < 1e-9 and abs(seq[3] - 1e-4) < 2e-9
    assert seq[4] == float(F32(1e-4))            # 8e-5 lifted by the final floor
    # the 6th point starts from the floored 1e-4 (not 8e-5): 1e-4*4e-4/5e-4 = 8e-5 again -> floored
    assert seq[5] == float(F32(1e-4))


# ---- C.5 binning ----------------------------------------------------------------------------------
@pytest.mark.parametrize("x,ix", [(15.02, 0), (15.051, -1), (-14.99, 599), (-15.0, -1), (0.0, 300), (0.049, 299),
                                  (-0.001, 300), (14.999, 0)])
def test_binning_even_L(oracle_mod, x, ix):
    m = oracle_mod.OracleMap(600, 0.05)
    got = m.points_to_index(x, 0.0)
    if ix < 0:
        assert got == -1
    else:
        # index along x is the row: geo = ix*L + iy with iy for y=0 -> 300
        assert got == ix * 600 + 300
        v = F32(300.0) - F32(x) / F32(0.05)
        assert int(v) == ix


def test_binning_plus_edge_truncation_quirk(oracle_mod):
    # (float)(L/2) - s/res in (-1, 0) truncates to 0 (accepted): points up to one cell beyond +edge
    m = oracle_mod.OracleMap(600, 0.05)
    assert m.points_to_index(15.02, 15.03) == 0
    assert m.points_to_index(15.06, 0.0) == -1


@pytest.mark.parametrize("x,ix", [(0.0, 37), (0.09, 37), (0.11, 36), (-0.09, 37), (-0.11, 38), (7.45, 0),
                                  (7.55, -1), (-7.45, 74), (-7.55, -1)])
def test_binning_odd_L(oracle_mod, x, ix):
    # L = 75, res 0.2 (kitti_demo_map.yaml): ix = 37 - (int)(x/0.2 + 0.5*sign(x)), sign(0) = -1
    m = oracle_mod.OracleMap(75, 0.2)
    got = m.points_to_index(x, 0.0)
    sx = F32(x)
    exp = 37 - int(np.float64(sx / F32(0.2)) + 0.5 * (1 if sx > 0 else -1))
    assert exp == ix or ix < 0
    if ix < 0:
        assert got == -1
    else:
        assert got == ix * 75 + 37


def test_binning_nonfinite_is_outside(oracle_mod):
    m = oracle_mod.OracleMap(600, 0.05)
    for bad in (np.inf, -np.inf, np.nan, 1e38, -1e38):
        assert m.points_to_map_index(bad, 0.0) == -1
        assert m.points_to_map_index(0.0, bad) == -1
    m75 = oracle_mod.OracleMap(75, 0.2)
    for bad in (np.inf, -np.inf, np.nan):
        assert m75.points_to_map_index(bad, 0.0) == -1


# ---- C.6 circular buffer / move -----------------------------------------------------------------------
def test_move_keeps_world_to_storage_mapping(oracle_mod):
    m = oracle_mod.OracleMap(20, 0.5)
    world = [(1.3, -2.2), (-3.1, 0.4), (0.26, 0.26)]
    before = [m.points_to_map_index(*w) for w in world]
    c, s, a = m.move([1.5, -1.0, 0.7])            # +3 cells in x, -2 cells in y
    assert list(a) == [1.5, -1.0] and list(c) == [1.5, -1.0]
    assert list(s) == [(0 - 3) % 20, (0 + 2) % 20]
    after = [m.points_to_map_index(*w) for w in world]
    assert before == after                        # same world point -> same storage cell


def test_move_clears_vacated_rows_and_cols(oracle_mod):
    L = 10
    m = oracle_mod.OracleMap(L, 1.0)
    m.set_layer("elevation", np.arange(L * L, dtype=F32).reshape(L, L))
    m.set_layer("variance", np.full((L, L), 0.5, F32))
    m.set_layer("traver", np.full((L, L), 0.25, F32))
    c, s, a = m.move([3.0, -2.0, 0.0])
    e = m.layer("elevation")
    # x shift +3: rows [wrap(0-3), 0) = 7,8,9 cleared; y shift -2: cols [0, 2) cleared
    cleared = np.zeros((L, L), bool); cleared[7:10, :] = True; cleared[:, 0:2] = True
    assert np.all(e[cleared] == -10) and np.all(m.layer("variance")[cleared] == -10)
    assert np.all(e[~cleared] == np.arange(L * L, dtype=F32).reshape(L, L)[~cleared])
    assert np.all(m.layer("traver") == 0.25)      # G_Clear_map leaves traver alone (GPU:255-276)
    assert list(s) == [7, 2]


def test_move_wrap_split_clear(oracle_mod):
    L = 10
    m = oracle_mod.OracleMap(L, 1.0)
    m.move([0.0, 2.0, 0.0])                       # start_y = 8
    m.set_layer("elevation", np.ones((L, L), F32))
    launches = m._l.gemo_move(m._m, (oracle_mod.c_float * 3)(0.0, -2.0, 0.0), (oracle_mod.c_float * 2)(),
                              (oracle_mod.c_int * 2)(), (oracle_mod.c_float * 2)())
    # y shift -4 from start 8: clears cols [8, 12) -> split [8,10) + [0,2)
    assert launches == 2
    e = m.layer("elevation")
    assert np.all(e[:, [8, 9, 0, 1]] == -10) and np.all(e[:, 2:8] == 1)


def test_move_small_motion_is_noop(oracle_mod):
    m = oracle_mod.OracleMap(20, 0.5)
    c, s, a = m.move([0.2, -0.2, 0.0])            # < half a cell
    assert list(s) == [0, 0] and list(c) == [0.0, 0.0] and list(a) == [0.0, 0.0]


def test_move_full_map_clear(oracle_mod):
    m = oracle_mod.OracleMap(8, 1.0)
    m.set_layer("elevation", np.ones((8, 8), F32)); m.set_layer("traver", np.ones((8, 8), F32))
    m.move([20.0, 0.0, 0.0])
    assert np.all(m.layer("elevation") == -10) and np.all(m.layer("traver") == -10)   # G_Clear_allmap resets traver


# ---- C.7 variance model -------------------------------------------------------------------------------
def test_variance_level_sensor(oracle_mod):
    m = oracle_mod.OracleMap(200, 0.1)
    out = m.process_points(ident_frame(), [1.0, 3.0], [2.0, -4.0], [0.5, 0.1])
    assert np.all(out["var"] == F32(0.018) * F32(0.018))      # Js = (0,0,1): var = min_r^2 exactly


def test_variance_pitched_sensor(oracle_mod):
    m = oracle_mod.OracleMap(200, 0.1)
    x, y, z = F32(1.0), F32(2.0), F32(0.5)
    out = m.process_points(ident_frame(Js=[1, 0, 0]), [x], [y], [z])
    d = np.sqrt(x * x + (y * y + z * z))
    t = F32(0.0015) + F32(0.0006) * d
    assert out["var"][0] == t * t


def test_rotation_variance_term(oracle_mod):
    m = oracle_mod.OracleMap(200, 0.1)
    Q = np.diag([1e-4, 2e-4, 3e-4]).astype(F32)
    out = m.process_points(ident_frame(rotation_variance=Q), [1.0], [2.0], [0.5])
    # Jq = e_z^T skew(p) = (-y, x, 0) -> var = y^2 q0 + x^2 q1 + min_r^2
    exp = 4.0 * 1e-4 + 1.0 * 2e-4 + 0.018 ** 2
    assert abs(out["var"][0] - exp) < 1e-9


# ---- C.8 reject filter ------------------------------------------------------------------------------------
def test_reject_filter_reference(oracle_mod):
    m = oracle_mod.OracleMap(200, 0.1)
    f = ident_frame(filter=RejectFilter.reference())
    pts = np.array([[0, -2, 0], [0, -0.5, 0], [0, 2, 0], [1, -1.2, 0], [2, -1.2, 0], [2, -1.0, 0]], F32)   # last: band test is strict (y > -1)
    out = m.process_points(f, pts[:, 0], pts[:, 1], pts[:, 2], write_back_xyz=True)
    assert list(out["index"] >= 0) == [True, False, False, False, True, True]
    # rejected points: all outputs -1 and x,y,z overwritten with -1 (GPU:441-451)
    for k in ("var", "x_ts", "y_ts", "height", "x", "y", "z"):
        assert np.all(out[k][[1, 2, 3]] == -1)


def test_height_window_is_strict(oracle_mod):
    m = oracle_mod.OracleMap(200, 0.1)
    out = m.process_points(ident_frame(lower=0.0, upper=1.0), [1, 1, 1, 1], [1, 1, 1, 1], [0.0, 1.0, 0.5, np.nan])
    assert list(out["index"] >= 0) == [False, False, True, False]


# ---- C.9 colour rule ------------------------------------------------------------------------------------------
def test_colour_rule(oracle_mod):
    m = oracle_mod.OracleMap(4, 0.1)
    m.fuse(np.array([0, 0, 1], np.int32), np.array([1.0, 1.001, 2.0], F32), np.array([4e-4] * 3, F32),
           R=[10, 20, 5], G=[11, 0, 6], B=[12, 22, 7], intensity=[3.0, 4.0, 0.0])
    assert m.layer("color_r").ravel()[0] == 10 and m.layer("intensity").ravel()[0] == 3.0   # 2nd point has G=0
    assert m.layer("color_r").ravel()[1] == 0 and m.layer("intensity").ravel()[1] == 0.0    # intensity 0 -> untouched
    assert m.layer("elevation").ravel()[0] != F32(1.0)                                      # ...but height fused


# ---- C.10 mapvar_update -----------------------------------------------------------------------------------------
def test_mapvar_update_noop_before_first_fuse(oracle_mod):
    m = oracle_mod.OracleMap(4, 0.1)
    m.mapvar_update(0.5)
    assert np.all(m.layer("variance") == -10)
    m.fuse(np.array([2], np.int32), np.array([1.0], F32), np.array([4e-4], F32))
    m.mapvar_update(0.5)
    v = m.layer("variance").ravel()
    assert v[2] == F32(4e-4) + F32(0.5) and np.all(np.delete(v, 2) == F32(1e-4) + F32(0.5))


# ---- sentinel and equivalence -----------------------------------------------------------------------------------
def test_height_minus_one_sentinel_skipped(oracle_mod):
    m = fuse_cells(oracle_mod, 4, [0, 1], [-1.0, -1.5], [4e-4, 4e-4])
    e = m.layer("elevation").ravel()
    assert e[0] == -10 and e[1] == F32(-1.5)          # GPU:482


def test_linear_restatement_equals_literal_per_cell_scan(oracle_mod):
    rng = np.random.default_rng(7)
    L, n = 12, 3000
    idx = rng.integers(-1, L * L, n).astype(np.int32)
    h = rng.normal(0, 0.2, n).astype(F32); v = rng.uniform(1e-5, 2e-3, n).astype(F32)
    R, G, B = (rng.integers(0, 3, n).astype(np.int32) for _ in range(3))
    I = rng.integers(0, 2, n).astype(F32)
    a, b = oracle_mod.OracleMap(L, 0.1), oracle_mod.OracleMap(L, 0.1)
    for _ in range(2):
        a.fuse(idx, h, v, R, G, B, I); b.fuse_literal(idx, h, v, R, G, B, I)
        a.mapvar_update(3e-5); b.mapvar_update(3e-5)
    for name in ("elevation", "variance", "intensity", "color_r", "color_g", "color_b"):
        assert np.array_equal(a.layer(name), b.layer(name)), name


def test_add_equals_process_then_fuse(oracle_mod):
    from gem_amd import synth
    wl = synth.config_c1()
    a, b = oracle_mod.OracleMap(wl.length, wl.resolution), oracle_mod.OracleMap(wl.length, wl.resolution)
    c = wl.clouds[0]
    a.add(wl.frames[0], c)
    out = b.process_points(wl.frames[0], c[:, 0], c[:, 1], c[:, 2])
    b.fuse(out["index"], out["height"], out["var"])
    assert np.array_equal(a.layer("elevation"), b.layer("elevation"))
    assert np.array_equal(a.layer("variance"), b.layer("variance"))
    assert a.last_counts[0] == out["accepted"] and a.last_counts[1] > 5000


def test_splitting_a_fuse_call_is_exact(oracle_mod):
    # Fuse(A ++ B) == Fuse(A); Fuse(B): what lets the product process a cloud in several passes
    rng = np.random.default_rng(11)
    L, n = 10, 4000
    idx = rng.integers(0, L * L, n).astype(np.int32)
    h = rng.normal(0, 0.3, n).astype(F32); v = rng.uniform(1e-6, 2e-3, n).astype(F32)
    a, b = oracle_mod.OracleMap(L, 0.1), oracle_mod.OracleMap(L, 0.1)
    a.fuse(idx, h, v)
    for lo, hi in ((0, 1000), (1000, 1001), (1001, 4000)):
        b.fuse(idx[lo:hi], h[lo:hi], v[lo:hi])
    assert np.array_equal(a.layer("elevation"), b.layer("elevation"))
    assert np.array_equal(a.layer("variance"), b.layer("variance"))


def test_all_core_oracle_equals_the_sequential_one(oracle_mod):
    """gemo_add_batch_mt (cells partitioned into row strips, every thread scans the index array in order) against the
    sequential mapvar_update + add calls: every layer bit for bit, for thread counts that do and do not divide L."""
    from gem_amd import synth
    wl = synth.config_c4(n_sweeps=3)
    clouds = [c[:40_000] for c in wl.clouds]
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])])
    seq = oracle_mod.OracleMap(wl.length, wl.resolution)
    seq.move([0.4, -0.3, 0.0])
    acc = 0
    for k in range(3):
        seq.mapvar_update(wl.var_updates[k]); seq.add(wl.frames[k], clouds[k]); acc += seq.last_counts[0]
    for nt in (1, 7, 16):
        mt = oracle_mod.OracleMap(wl.length, wl.resolution)
        mt.move([0.4, -0.3, 0.0])
        assert mt.add_batch_mt(wl.frames, np.concatenate(clouds), off, wl.var_updates, nt) == acc
        for name in ("elevation", "variance", "lowest", "intensity"):
            assert np.array_equal(mt.layer(name), seq.layer(name)), (nt, name)
    mt = oracle_mod.OracleMap(wl.length, wl.resolution)          # without increments
    seq = oracle_mod.OracleMap(wl.length, wl.resolution)
    mt.add_batch_mt(wl.frames, np.concatenate(clouds), off, None, 5)
    for k in range(3):
        seq.add(wl.frames[k], clouds[k])
    assert np.array_equal(mt.layer("variance"), seq.layer("variance")) and np.array_equal(mt.layer("elevation"), seq.layer("elevation"))


# ---- the reference's CPU noise models (SL.cpp:121-153, Stereo.cpp:72-104, Perfect.cpp:74-102): closed forms ------------------
# The reference never calls these on its GPU path and nothing in its tree pins their outputs.  tests/test_reference_sensor_models.py
# compares the oracle with the reference's own files compiled against stand-ins for Eigen / kindr / PCL / ROS; these hand-derived
# cases are the independent check (of the oracle's restatement and, in tests/test_parity_gpu.py, of the kernel's): with a level sensor (J_s = e_z, no rotation variance) the height variance IS the normal variance of the
# model, with the sensor's x axis along map z it is the lateral one -- each a closed formula of the point, evaluated here in
# plain Python from the reference's source lines.
def model_frame(kind, params, Js=(0, 0, 1), width=0):
    f = ident_frame(Js=list(Js))
    f.model = SensorModel(kind, tuple(params), float("inf"), float("-inf"), original_width=width)
    return f


def test_structured_light_normal_and_lateral_variance(oracle_mod):
    a, b, c, d, e, k = 0.000611, 0.003587, 0.3515, 0.0007, 2.0, 0.01576       # realsense_d435.yaml shape, d / e made non-trivial
    m = oracle_mod.OracleMap(200, 0.1)
    pts = [(0.4, -0.2, 0.8), (1.5, 0.7, 2.25), (0.1, 0.1, 0.3515)]
    x, y, z = (np.array(v, F32) for v in zip(*pts))
    out = m.process_points(model_frame(1, (a, b, c, d, e, k)), x, y, z)
    for i, zi in enumerate(z):
        zd = float(zi)                                                       # SL.cpp:128 measurementDistance = point.z (float)
        dev_n = F32(a + b * (zd - c) * (zd - c) + d * zd ** e)               # SL.cpp:130-133: double expression, stored in a float
        assert out["var"][i] == dev_n * dev_n, i                             # SL.cpp:134, :147 with J_s = (0, 0, 1)
    out = m.process_points(model_frame(1, (a, b, c, d, e, k), Js=(1, 0, 0)), x, y, z)
    for i, zi in enumerate(z):
        dev_l = F32(k * float(zi))                                           # SL.cpp:135
        assert out["var"][i] == dev_l * dev_l, i                             # SL.cpp:136, J_s = (1, 0, 0) picks varianceLateral


def test_stereo_normal_and_lateral_variance(oracle_mod):
    p1, p2, p3, p4, p5, lat, f = 0.1, 0.001, 380.0, 1.0, 0.002, 0.001, 30.0
    W = 640
    m = oracle_mod.OracleMap(200, 0.1)
    pts = [(0.4, -0.2, 0.8), (1.5, 0.7, 2.25), (-0.3, 0.2, 5.0)]
    orig = np.array([0, 17 * W + 5, 239 * W + 639], np.int32)               # pixel (I, J) = (index / width, index % width), Stereo.cpp:108-116
    x, y, z = (np.array(v, F32) for v in zip(*pts))
    out = m.process_points(model_frame(2, (p1, p2, p3, p4, p5, lat, f), width=W), x, y, z, orig_index=orig)
    for i, zi in enumerate(z):
        I, J = int(orig[i]) // W, int(orig[i]) % W
        disp = f / float(zi)                                                 # Stereo.cpp:78
        vn = (f / disp ** 2) ** 2 * ((p5 * disp + p2) * np.sqrt((p3 * disp + p4 - J) ** 2 + (240 - I) ** 2) + p1)   # :88-91
        assert abs(out["var"][i] - F32(vn)) <= 2e-7 * abs(vn), i             # double pow / sqrt, then one float rounding
    out = m.process_points(model_frame(2, (p1, p2, p3, p4, p5, lat, f), Js=(1, 0, 0), width=W), x, y, z, orig_index=orig)
    for i in range(3):
        dist = np.sqrt(x[i] * x[i] + (y[i] * y[i] + z[i] * z[i]))           # Stereo.cpp:85 pointVector.norm() (float)
        vl = F32((lat * float(dist)) ** 2)                                   # :92 pow(lateral_factor * distance, 2)
        assert abs(out["var"][i] - vl) <= 2e-7 * vl, i


def test_perfect_sensor_has_only_the_rotation_term(oracle_mod):
    m = oracle_mod.OracleMap(200, 0.1)
    f = model_frame(3, ())
    out = m.process_points(f, [1.0], [2.0], [0.5])
    assert out["var"][0] == 0.0                                              # Perfect.cpp:86-88: zero sensor variance
    f.rotation_variance = np.diag([1e-4, 2e-4, 3e-4]).astype(F32)
    out = m.process_points(f, [1.0], [2.0], [0.5])
    assert abs(out["var"][0] - (4.0 * 1e-4 + 1.0 * 2e-4)) < 1e-9            # Jq = e_z^T skew(p) = (-y, x, 0)
