"""ctypes binding of oracle/_ref/libgem_ref.so -- the REFERENCE's own gpu_process.cu compiled for the CPU
(oracle/ref_build/build_ref.py).  TEST INFRASTRUCTURE ONLY: used by tests/ to pin the oracle's restatement against the
reference's code; never imported by gem_amd.

The reference keeps ONE map in file-scope state, so there is one RefMap per process at a time (a new RefMap re-runs
Init_GPU_elevationmap).  Only the laser sensor model and the hard-coded reject filter / thresholds of the reference
exist here (gpu_process.cu:393, 500-504): compare with the oracle configured the same way.
"""
from __future__ import annotations

import ctypes as C
import sys
from ctypes import POINTER, c_double, c_float, c_int, c_void_p
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE / "ref_build"))
_lib = None


def lib():
    """Load (building when /root/reference is present) the compiled reference; None when neither exists."""
    global _lib
    if _lib is None:
        import build_ref
        path = build_ref.build()
        if path is None:
            return None
        l = C.CDLL(str(path))
        l.gemref_init.argtypes = [c_int, c_float, c_float, c_float]
        l.gemref_process_points.restype = c_int
        l.gemref_process_points.argtypes = [c_void_p] * 9 + [c_int, c_double, c_double, c_float, c_float, c_float] + [c_void_p] * 5
        l.gemref_fuse.argtypes = [c_int, c_int] + [c_void_p] * 7
        l.gemref_mapvar_update.argtypes = [c_int, c_float]
        l.gemref_move.argtypes = [POINTER(c_float), c_float, c_int, POINTER(c_float), POINTER(c_int), POINTER(c_float)]
        l.gemref_map_optmove.argtypes = [POINTER(c_float), c_float, c_float, c_int, POINTER(c_float)]
        l.gemref_map_closeloop.argtypes = [POINTER(c_float), c_float, c_int, c_float]
        l.gemref_map_feature.argtypes = [c_int] + [c_void_p] * 9
        l.gemref_raytracing.argtypes = [c_int]
        l.gemref_get_layer.argtypes = [c_int, c_void_p]; l.gemref_get_layer.restype = c_int
        l.gemref_set_layer.argtypes = [c_int, c_void_p]; l.gemref_set_layer.restype = c_int
        l.gemref_get_pose.argtypes = [POINTER(c_float), POINTER(c_int)]
        _lib = l
    return _lib


def _vp(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class RefMap:
    """The surface of oracle.OracleMap that the reference implements."""

    FLOAT = {"elevation": 0, "variance": 1, "intensity": 2, "traver": 3, "lowest": 4}
    INT = {"color_r": 5, "color_g": 6, "color_b": 7}

    def __init__(self, length: int, resolution: float, mahalanobis_threshold: float = 5.0, obstacle_threshold: float = 0.7):
        self._l = lib()
        if self._l is None:
            raise RuntimeError("oracle/_ref/libgem_ref.so is missing and /root/reference is not there to build it from")
        self.length, self.resolution = int(length), float(resolution)
        self._l.gemref_init(self.length, self.resolution, float(mahalanobis_threshold), float(obstacle_threshold))

    def move(self, position):
        pos = (c_float * 3)(*[float(v) for v in position])
        c = (c_float * 2)(); s = (c_int * 2)(); a = (c_float * 2)()
        self._l.gemref_move(pos, self.resolution, self.length, c, s, a)
        return np.array(c[:], np.float32), np.array(s[:], np.int32), np.array(a[:], np.float32)

    def pose(self):
        c = (c_float * 2)(); s = (c_int * 2)()
        self._l.gemref_get_pose(c, s)
        return np.array(c[:], np.float32), np.array(s[:], np.int32)

    def process_points(self, frame, x, y, z):
        n = int(np.asarray(x).size)
        xa, ya, za = (np.array(v, np.float32, copy=True).reshape(-1) for v in (x, y, z))
        out = {"index": np.empty(n, np.int32), "var": np.empty(n, np.float32), "x_ts": np.empty(n, np.float32),
               "y_ts": np.empty(n, np.float32), "height": np.empty(n, np.float32)}
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        T, sj, rv, cs, pm, bs = (f32(frame.T), f32(frame.sensor_jacobian), f32(frame.rotation_variance), f32(frame.C_SB_T),
                                 f32(frame.P_mul_C_BM_T), f32(frame.B_r_BS_skew))
        p = [float(v) for v in frame.model.params]
        self._l.gemref_process_points(_vp(out["index"]), _vp(xa), _vp(ya), _vp(za), _vp(out["var"]), _vp(out["x_ts"]), _vp(out["y_ts"]),
                                      _vp(out["height"]), _vp(T), n, float(frame.lower), float(frame.upper), p[0], p[1], p[2],
                                      _vp(sj), _vp(rv), _vp(cs), _vp(pm), _vp(bs))
        out["x"], out["y"], out["z"] = xa, ya, za                 # the reference overwrites rejected inputs with -1 (GPU:443-446)
        return out

    def fuse(self, index, height, var, R=None, G=None, B=None, intensity=None):
        n = int(np.asarray(index).size)
        i, h, v = np.ascontiguousarray(index, np.int32), np.ascontiguousarray(height, np.float32), np.ascontiguousarray(var, np.float32)
        z = lambda a, t: np.zeros(n, t) if a is None else np.ascontiguousarray(a, t)       # the reference always takes the arrays
        r, g, b, I = z(R, np.int32), z(G, np.int32), z(B, np.int32), z(intensity, np.float32)
        self._l.gemref_fuse(self.length, n, _vp(i), _vp(r), _vp(g), _vp(b), _vp(I), _vp(h), _vp(v))

    def mapvar_update(self, u: float):
        self._l.gemref_mapvar_update(self.length, float(u))

    def map_optmove(self, opt_xy, height_update: float):
        p = (c_float * 2)(float(opt_xy[0]), float(opt_xy[1])); out = (c_float * 2)()
        self._l.gemref_map_optmove(p, float(height_update), self.resolution, self.length, out)
        return np.array([out[0], out[1]], np.float32)

    def map_closeloop(self, xy, height_update: float):
        p = (c_float * 2)(float(xy[0]), float(xy[1]))
        self._l.gemref_map_closeloop(p, float(height_update), self.length, self.resolution)

    def raytracing(self):
        self._l.gemref_raytracing(self.length)

    def map_feature(self):
        n = self.length * self.length
        f = {k: np.zeros(n, np.float32) for k in ("elevation", "var", "rough", "slope", "traver", "intensity")}
        c = {k: np.zeros(n, np.int32) for k in "RGB"}
        self._l.gemref_map_feature(self.length, _vp(f["elevation"]), _vp(f["var"]), _vp(c["R"]), _vp(c["G"]), _vp(c["B"]),
                                   _vp(f["rough"]), _vp(f["slope"]), _vp(f["traver"]), _vp(f["intensity"]))
        return {k: f[k].reshape(self.length, self.length) for k in ("rough", "slope", "traver")}

    def layer(self, name: str) -> np.ndarray:
        n = self.length * self.length
        if name in self.INT:
            out = np.empty(n, np.int32); self._l.gemref_get_layer(self.INT[name], _vp(out))
        else:
            out = np.empty(n, np.float32); self._l.gemref_get_layer(self.FLOAT[name], _vp(out))
        return out.reshape(self.length, self.length)

    def set_layer(self, name: str, values):
        if name in self.INT:
            a = np.ascontiguousarray(values, np.int32).reshape(-1); self._l.gemref_set_layer(self.INT[name], _vp(a))
        else:
            a = np.ascontiguousarray(values, np.float32).reshape(-1); self._l.gemref_set_layer(self.FLOAT[name], _vp(a))


# ---- the reference's RobotMotionMapUpdater.cpp compiled against stand-ins for Eigen / kindr / ROS (build_ref.build_motion) ----------
_motion_lib = None


def motion_lib():
    global _motion_lib
    if _motion_lib is None:
        import build_ref
        path = build_ref.build_motion()
        if path is None:
            return None
        l = C.CDLL(str(path))
        l.gemref_motion_create.restype = c_void_p; l.gemref_motion_create.argtypes = [c_double]
        l.gemref_motion_destroy.argtypes = [c_void_p]
        l.gemref_motion_update.restype = c_int
        l.gemref_motion_update.argtypes = [c_void_p, POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double), c_int, c_double, POINTER(c_float)]
        _motion_lib = l
    return _motion_lib


_sensor_lib = None


def sensor_lib():
    """oracle/_ref/libgem_ref_sensors.so: the reference's Perfect / Stereo / StructuredLight SensorProcessor.cpp (build_ref.build_sensors)."""
    global _sensor_lib
    if _sensor_lib is None:
        import build_ref
        path = build_ref.build_sensors()
        if path is None:
            return None
        l = C.CDLL(str(path))
        l.gemref_sensor_variances.restype = c_int
        l.gemref_sensor_variances.argtypes = [c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double), POINTER(c_double),
                                              c_int, POINTER(c_float), POINTER(c_float), POINTER(c_float), c_int, POINTER(c_int), POINTER(c_float)]
        _sensor_lib = l
    return _sensor_lib


def sensor_variances(model: int, params, rotation_map_to_base, rotation_base_to_sensor, translation_base_to_sensor, pose_covariance,
                     x, y, z, original_width: int = 0, indices=None) -> np.ndarray:
    """What the reference's own computeVariances writes for a cloud (model: 1 structured light, 2 stereo, 3 perfect).  The rotations
    are what SensorProcessorBase::updateTransformations stores (SPB.cpp:110-117); pose_covariance is the 6x6 robot pose covariance."""
    l = sensor_lib()
    dp = lambda a, n: np.ascontiguousarray(np.asarray(a, np.float64).reshape(-1)[:n] if n else np.asarray(a, np.float64).reshape(-1))
    prm = np.zeros(8, np.float64); prm[:len(params)] = params
    cbm, csb, t, cov = dp(rotation_map_to_base, 9), dp(rotation_base_to_sensor, 9), dp(translation_base_to_sensor, 3), dp(pose_covariance, 36)
    xs, ys, zs = (np.ascontiguousarray(v, np.float32) for v in (x, y, z))
    n = xs.size
    idx = np.ascontiguousarray(np.arange(n) if indices is None else indices, np.int32)
    out = np.empty(n, np.float32)
    p = lambda a, t_: a.ctypes.data_as(POINTER(t_))
    rc = l.gemref_sensor_variances(int(model), p(prm, c_double), p(cbm, c_double), p(csb, c_double), p(t, c_double), p(cov, c_double), n,
                                   p(xs, c_float), p(ys, c_float), p(zs, c_float), int(original_width), p(idx, c_int), p(out, c_float))
    if rc != n:
        raise RuntimeError(f"gemref_sensor_variances returned {rc}")
    return out


def quaternion_from_matrix(R) -> np.ndarray:
    """Hamilton unit quaternion (w, x, y, z), w >= 0, of a rotation matrix."""
    R = np.asarray(R, np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2; q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R))); j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0] * 4
        q[0] = (R[k, j] - R[j, k]) / s; q[1 + i] = 0.25 * s; q[1 + j] = (R[j, i] + R[i, j]) / s; q[1 + k] = (R[k, i] + R[i, k]) / s
    q = np.array(q); q /= np.linalg.norm(q)
    return q if q[0] >= 0 else -q


class RefMotion:
    """RobotMotionMapUpdater of the reference: update() with a pose (position, rotation matrix R_IB), the 6 x 6 pose covariance and
    the map's rotation; returns the float it hands to Mapvar_update (RMU.cpp:80-81)."""

    def __init__(self, covariance_scale: float = 1.0, length: int = 600):
        self._l = motion_lib()
        self._u = self._l.gemref_motion_create(covariance_scale)
        self._t, self._len = 0.0, length

    def __del__(self):
        if getattr(self, "_u", None):
            self._l.gemref_motion_destroy(self._u); self._u = None

    def compute(self, position, R_IB, cov6x6, map_rotation=None) -> float:
        d = lambda a, n: (c_double * n)(*np.asarray(a, np.float64).reshape(-1).tolist())
        self._t += 1.0
        out = c_float()
        q, mq = quaternion_from_matrix(R_IB), quaternion_from_matrix(np.eye(3) if map_rotation is None else map_rotation)
        ok = self._l.gemref_motion_update(self._u, d(position, 3), d(q, 4), d(cov6x6, 36), d(mq, 4), self._len, self._t, C.byref(out))
        assert ok == 1, "the reference skipped the update"
        return float(out.value)
