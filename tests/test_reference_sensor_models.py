"""The CPU noise models against the REFERENCE'S OWN TEXT: Perfect / Stereo / StructuredLight
SensorProcessor.cpp compiled where they lie against stand-ins for Eigen / kindr / PCL / ROS / TF
(oracle/ref_build/build_ref.py: build_sensors -> oracle/_ref/libgem_ref_sensors.so).

What is compared is the height variance computeVariances writes per point with what the oracle (and so the product, which is held
to the oracle on the GPU) computes for the same frame, the frame built by the product's host code (gem_amd.SensorProcessor, rows
a3 / a4) from the same rotations:
* sensor-model part alone -- no rotation covariance, the sensor Jacobian a unit vector: the result IS the model's normal or
  lateral variance -- bit for bit;
* general poses with a rotation covariance: the reference's CPU code folds its two quadratic forms with Eigen's three-term
  reduction a0 + (a1 + a2), the GPU path the product follows (gpu_process.cu:293-298, cuda_computer) left to right
  (a0 + a1) + a2 -- a rounding apart in each of the two forms: at most 2 float ulps measured, 3 allowed.
"""
import numpy as np
import pytest

from gem_amd import SensorModel, SensorProcessor, synth

F32 = np.float32
SL = (0.000611, 0.003587, 0.3515, 0.0007, 2.3, 0.01576)            # realsense_d435.yaml shape, d / e made non-trivial
STEREO = (0.1, 0.001, 380.0, 1.0, 0.002, 0.001, 30.0)
W = 640


@pytest.fixture(scope="module")
def sensors():
    import ref
    if ref.sensor_lib() is None:
        pytest.skip("no /root/reference to build from and no prebuilt oracle/_ref/libgem_ref_sensors.so")
    return ref


def cloud(rng, n):
    x = rng.uniform(-2.0, 2.0, n).astype(F32); y = rng.uniform(-2.0, 2.0, n).astype(F32); z = rng.uniform(0.3, 6.0, n).astype(F32)
    return x, y, z


def oracle_var(oracle_mod, model, R_mb, R_bs, t_bs, Q, x, y, z, orig=None):
    sp = SensorProcessor(model, rotation_variance=np.asarray(Q, F32))
    T_bs = np.eye(4); T_bs[:3, :3] = R_bs; T_bs[:3, 3] = t_bs
    T_mb = np.eye(4); T_mb[:3, :3] = R_mb
    sp.update_transformations(T_mb @ T_bs, T_bs, T_mb)
    f = sp.frame(); f.lower, f.upper = -1e9, 1e9
    out = oracle_mod.OracleMap(64, 0.1).process_points(f, x, y, z, orig_index=orig)
    assert (out["var"] >= 0).all()
    return out["var"]


def ulps(a, b):
    a, b = np.asarray(a, F32), np.asarray(b, F32)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


# rotationBaseToSensor_ choices that make e_z^T C_BM^T C_SB^T a unit vector: e_z (level) and e_x (the sensor's x axis along map z)
LEVEL = np.eye(3)
X_UP = np.array([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [-1.0, 0.0, 0.0]]).T     # C_SB with C_SB^T's last row = (1, 0, 0)


@pytest.mark.parametrize("R_bs,what", [(LEVEL, "normal"), (X_UP, "lateral")])
def test_structured_light_model_is_the_references(sensors, oracle_mod, R_bs, what):
    rng = np.random.default_rng(11)
    x, y, z = cloud(rng, 4000)
    ref = sensors.sensor_variances(1, SL, np.eye(3), R_bs, np.zeros(3), np.zeros((6, 6)), x, y, z)
    mine = oracle_var(oracle_mod, SensorModel(1, SL, float("inf"), float("-inf")), np.eye(3), R_bs, np.zeros(3), np.zeros((3, 3)), x, y, z)
    assert np.array_equal(ref, mine), (what, int((ref != mine).sum()))
    assert ref.min() > 0


@pytest.mark.parametrize("R_bs,what", [(LEVEL, "normal"), (X_UP, "lateral")])
def test_stereo_model_is_the_references(sensors, oracle_mod, R_bs, what):
    rng = np.random.default_rng(12)
    x, y, z = cloud(rng, 4000)
    orig = rng.integers(0, W * 480, x.size).astype(np.int32)            # the pixel a point came from (Stereo.cpp:108-116)
    ref = sensors.sensor_variances(2, STEREO, np.eye(3), R_bs, np.zeros(3), np.zeros((6, 6)), x, y, z, original_width=W, indices=orig)
    mine = oracle_var(oracle_mod, SensorModel(2, STEREO, float("inf"), float("-inf"), original_width=W), np.eye(3), R_bs, np.zeros(3),
                      np.zeros((3, 3)), x, y, z, orig)
    assert np.array_equal(ref, mine), (what, int((ref != mine).sum()), ulps(ref, mine).max())
    assert ref.min() > 0


@pytest.mark.parametrize("kind,params", [(1, SL), (2, STEREO), (3, ())])
def test_general_poses_with_a_rotation_covariance(sensors, oracle_mod, kind, params):
    rng = np.random.default_rng(20 + kind)
    worst = 0
    for trial in range(6):
        R_mb = synth.rot_zyx(*rng.uniform(-0.4, 0.4, 3)); R_bs = synth.rot_zyx(*rng.uniform(-1.0, 1.0, 3))
        t_bs = rng.uniform(-0.3, 0.3, 3)
        a = rng.normal(size=(6, 6)) * 2e-2; cov = a @ a.T
        x, y, z = cloud(rng, 1500)
        orig = rng.integers(0, W * 480, x.size).astype(np.int32)
        ref = sensors.sensor_variances(kind, params, R_mb, R_bs, t_bs, cov, x, y, z, original_width=W, indices=orig)
        model = SensorModel(kind, tuple(params), float("inf"), float("-inf"), original_width=W if kind == 2 else 0)
        mine = oracle_var(oracle_mod, model, R_mb, R_bs, t_bs, cov[3:, 3:], x, y, z, orig if kind == 2 else None)
        worst = max(worst, int(ulps(ref, mine).max()))
        assert ulps(ref, mine).max() <= 3, (trial, int(ulps(ref, mine).max()))      # measured: 2 at most, on a quarter of the points
        assert (ref > 0).all()
    assert worst <= 3


def test_perfect_sensor_without_rotation_covariance_is_exactly_zero(sensors, oracle_mod):
    rng = np.random.default_rng(5)
    x, y, z = cloud(rng, 100)
    R = synth.rot_zyx(0.3, -0.2, 0.1)
    ref = sensors.sensor_variances(3, (), R, R.T, [0.1, 0.2, 0.3], np.zeros((6, 6)), x, y, z)
    assert (ref == 0).all()
