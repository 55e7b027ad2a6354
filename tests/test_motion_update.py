"""RobotMotionMapUpdater (RMU.cpp:42-145): the product's host-side implementation (numpy, doubles)
against the C oracle restatement and hand-derived cases.  One float32 reaches the device."""
import numpy as np

from gem_amd import RobotMotionMapUpdater, synth


def rand_cov(rng):
    a = rng.normal(size=(6, 6)) * 1e-2
    return a @ a.T


def test_zero_covariance_gives_zero_update(oracle_mod):
    # what the reference actually feeds today (EMg.cpp:944-945: zero 6x6 covariance)
    u = RobotMotionMapUpdater().compute([1, 2, 0.5], synth.rot_zyx(0.3, 0.02, -0.01), np.zeros((6, 6)))
    assert u == 0.0


def test_level_robot_z_variance_passes_through():
    cov = np.zeros((6, 6)); cov[2, 2] = 4e-5
    u = RobotMotionMapUpdater().compute([0.1, 0, 0], np.eye(3), cov)
    assert abs(u - 4e-5) < 4e-5 * 1e-6            # one float32 rounding (RMU.cpp:69 .cast<float>())
    # scale factor (robot_motion_map_update/covariance_scale, RMU.cpp:38)
    u2 = RobotMotionMapUpdater(covariance_scale=2.5).compute([0.1, 0, 0], np.eye(3), cov)
    assert abs(u2 - 1e-4) < 1e-4 * 1e-6


def test_matches_oracle_over_a_trajectory(oracle_mod):
    rng = np.random.default_rng(3)
    mine, ref = RobotMotionMapUpdater(1.3), oracle_mod.OracleMotion(1.3)
    pos = np.zeros(3)
    for k in range(25):
        pos = pos + rng.normal(0, 0.2, 3)
        R = synth.rot_zyx(0.1 * k, rng.normal(0, 0.05), rng.normal(0, 0.05))
        cov = rand_cov(rng) * (1 + 0.1 * k)
        a, b = mine.compute(pos, R, cov), ref.compute(pos, R, cov)
        assert abs(a - b) <= 1e-6 * max(abs(b), 1e-9) + 1e-12, (k, a, b)


def test_relative_covariance_is_a_difference():
    # same covariance twice, no motion: relative covariance = reduced - F prev F^T = 0
    cov = np.diag([1e-4, 2e-4, 3e-4, 1e-5, 1e-5, 1e-5])
    m = RobotMotionMapUpdater()
    first = m.compute([0, 0, 0], np.eye(3), cov)
    second = m.compute([0, 0, 0], np.eye(3), cov)
    assert abs(first - 3e-4) < 3e-4 * 1e-6 and abs(second) < 1e-12


# ---- hand-derived cases with a pitched robot (RMU.cpp:92-145) ------------------------------------------------------------------
# The reference's updater needs kindr + ROS and cannot be compiled here; the product's numpy implementation and the oracle's C
# restatement must not be each other's only witness.  Two closed forms derived from the reference's source by hand:
def Ry(p):
    return np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])


def test_pitched_robot_first_update_mixes_x_into_z(oracle_mod):
    # previousReducedCovariance_ = 0, previous pose = identity, position at the origin: F = I (RMU.cpp:121-129 with v = 0), so the
    # relative covariance is the reduced one (:140-142) and its position block the pose's.  J_r = -(R_IB^T R_IM)^T = -R_IB for a map
    # frame aligned with the world (:62-66), whose third row is (sin p, 0, -cos p):  var = a sin^2 p + c cos^2 p   (:69, :80)
    p, a, b, c = 0.3, 4e-4, 9e-4, 1e-4
    cov = np.zeros((6, 6)); cov[0, 0], cov[1, 1], cov[2, 2] = a, b, c
    want = a * np.sin(p) ** 2 + c * np.cos(p) ** 2
    for upd in (RobotMotionMapUpdater(), oracle_mod.OracleMotion()):
        got = upd.compute([0, 0, 0], Ry(p), cov)
        assert abs(got - want) <= 2e-7 * want, type(upd).__name__


def test_tan_pitch_jacobian_couples_yaw_variance_into_height(oracle_mod):
    # Rotation covariance diag(sa, sb, sc), no position covariance, R = Ry(p) in both steps (yaw 0: the Euler ZYX angles of Ry(p) are
    # (0, p, 0), its rotation vector (0, p, 0) has no z part, so R_I_tilde_B = I).  yawJacobian = (tan p, 0, 1) (:97-100), hence
    # reduced[3][3] = tan^2 p sa + sc =: s, every other entry 0 (:107).
    # Step 1 at the origin: v = 0, F = I, relative = reduced, position block 0: update 0.
    # Step 2 at (0, d, 0): v = R^T (0, d, 0) = (0, d, 0) (:121-123); F's last column = e_z x v = (-d, 0, 0) (:129); the position block
    # of reduced - F prev F^T is -s d^2 e_x e_x^T (:140-142, G = I); with the third row of J_r = -R, (sin p, 0, -cos p):
    #     var = -s d^2 sin^2 p
    p, d, sa, sb, sc = 0.25, 0.4, 3e-4, 5e-4, 2e-4
    cov = np.zeros((6, 6)); cov[3, 3], cov[4, 4], cov[5, 5] = sa, sb, sc
    s = np.tan(p) ** 2 * sa + sc
    want = -s * d * d * np.sin(p) ** 2
    for upd in (RobotMotionMapUpdater(1.0), oracle_mod.OracleMotion(1.0)):
        first = upd.compute([0, 0, 0], Ry(p), cov)
        second = upd.compute([0, d, 0], Ry(p), cov)
        assert abs(first) < 1e-15, type(upd).__name__
        assert abs(second - want) <= 2e-7 * abs(want), (type(upd).__name__, second, want)
    # the covariance scale multiplies the pose covariance, hence the update (RMU.cpp:46)
    upd = RobotMotionMapUpdater(2.0)
    upd.compute([0, 0, 0], Ry(p), cov)
    assert abs(upd.compute([0, d, 0], Ry(p), cov) - 2.0 * want) <= 4e-7 * abs(want)


def test_against_the_references_own_code_over_random_trajectories(oracle_mod):
    """RobotMotionMapUpdater.cpp itself -- its Jacobians (A.4, A.5), the F matrix (A.8), the relative covariance (A.13) and the
    projection into the map (RMU.cpp:58-69) are the reference's text, compiled where it lies against stand-ins for Eigen / kindr /
    ROS (oracle/ref_build/motion) -- against the oracle's restatement AND the product's host implementation: random walks with
    pitched and rolled robots, full covariances, a rotated map, several covariance scales."""
    import ref
    if ref.motion_lib() is None:
        import pytest
        pytest.skip("no /root/reference to build from and no prebuilt oracle/_ref/libgem_ref_motion.so")
    worst = 0.0
    for seed in range(12):
        rng = np.random.default_rng(100 + seed)
        scale = [1.0, 0.5, 2.5, 1.3][seed % 4]
        theirs, ours, mine = ref.RefMotion(scale), oracle_mod.OracleMotion(scale), RobotMotionMapUpdater(scale)
        map_R = synth.rot_zyx(rng.normal(0, 0.4), 0.0, 0.0) if seed % 3 == 0 else None
        pos = np.zeros(3)
        for k in range(20):
            pos = pos + rng.normal(0, 0.3, 3)
            R = synth.rot_zyx(rng.uniform(-3.0, 3.0), rng.normal(0, 0.25), rng.normal(0, 0.25))
            cov = rand_cov(rng) * (1 + 0.2 * k)
            t, o, m = theirs.compute(pos, R, cov, map_R), ours.compute(pos, R, cov, map_R), mine.compute(pos, R, cov, map_R)
            tol = 2e-6 * max(abs(t), 1e-9) + 1e-12      # one float32 ulp: the doubles' summation order differs between the three
            assert abs(o - t) <= tol and abs(m - t) <= tol, (seed, k, t, o, m)
            worst = max(worst, abs(o - t) / max(abs(t), 1e-12))
    assert worst < 2e-6
