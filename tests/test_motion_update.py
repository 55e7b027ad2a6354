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
