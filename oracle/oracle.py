"""ctypes binding of the CPU ORACLE (oracle/gem_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under gem_amd/ imports this module.  Pinned against the reference's own
gpu_process.cu compiled for the CPU (oracle/ref.py, tests/test_reference_compiled.py) and by tests/test_oracle_kat.py.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from ctypes import POINTER, c_double, c_float, c_int, c_longlong, c_void_p
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "libgem_oracle.so"


class OFrame(C.Structure):
    _fields_ = [("T", c_float * 16), ("lower", c_double), ("upper", c_double), ("sensor_model", c_int),
                ("sp", c_double * 8), ("sensor_jacobian", c_float * 3), ("rotation_variance", c_float * 9),
                ("C_SB_T", c_float * 9), ("P_mul_C_BM_T", c_float * 3), ("B_r_BS_skew", c_float * 9),
                ("filter_on", c_int), ("filter_box_x", c_float), ("filter_box_y", c_float),
                ("filter_band_y", c_float), ("filter_plane_y", c_float), ("original_width", c_int)]


class OMap(C.Structure):
    _fields_ = [("L", c_int), ("res", c_float), ("mahal", c_float), ("var_floor", c_float),
                ("elevation", POINTER(c_float)), ("variance", POINTER(c_float)), ("intensity", POINTER(c_float)),
                ("traver", POINTER(c_float)), ("lowest", POINTER(c_float)),
                ("colorR", POINTER(c_int)), ("colorG", POINTER(c_int)), ("colorB", POINTER(c_int)),
                ("center", c_float * 2), ("start", c_int * 2), ("sensor_z", c_float), ("obstacle_threshold", c_float)]


class OMotion(C.Structure):
    _fields_ = [("prev_reduced_cov", c_double * 16), ("prev_pos", c_double * 3), ("prev_R", c_double * 9),
                ("covariance_scale", c_double)]


def build(force: bool = False) -> Path:
    srcs = [HERE / "gem_oracle.c", HERE / "gem_oracle_motion.c", HERE / "gem_oracle_feature.c", HERE / "gem_oracle_raytrace.c",
            HERE / "gem_oracle_mt.c", HERE / "gem_oracle_show.c", HERE / "gem_oracle.h"]
    if force or not LIB.exists() or any(s.stat().st_mtime > LIB.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(HERE), "-B" if force else "-s", "libgem_oracle.so"], check=True,
                       capture_output=True)
    return LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        l = C.CDLL(str(build()))
        l.gemo_create.restype = POINTER(OMap); l.gemo_create.argtypes = [c_int, c_float, c_float, c_float]
        l.gemo_destroy.argtypes = [POINTER(OMap)]
        l.gemo_move.restype = c_int
        l.gemo_move.argtypes = [POINTER(OMap), POINTER(c_float), POINTER(c_float), POINTER(c_int), POINTER(c_float)]
        l.gemo_points_to_index.restype = c_int; l.gemo_points_to_index.argtypes = [POINTER(OMap), c_float, c_float]
        l.gemo_points_to_map_index.restype = c_int; l.gemo_points_to_map_index.argtypes = [POINTER(OMap), c_float, c_float]
        l.gemo_process_points.restype = c_int
        l.gemo_process_points.argtypes = [POINTER(OMap), POINTER(OFrame), c_int] + [c_void_p] * 9
        l.gemo_fuse.argtypes = [POINTER(OMap), c_int] + [c_void_p] * 7
        l.gemo_fuse_literal.argtypes = [POINTER(OMap), c_int] + [c_void_p] * 7
        l.gemo_mapvar_update.argtypes = [POINTER(OMap), c_float]
        l.gemo_map_feature.argtypes = [POINTER(OMap), c_void_p, c_void_p, c_void_p]
        l.gemo_map_optmove.argtypes = [POINTER(OMap), POINTER(c_float), c_float, POINTER(c_float)]
        l.gemo_map_closeloop.argtypes = [POINTER(OMap), POINTER(c_float), c_float]
        l.gemo_raytracing.argtypes = [POINTER(OMap)]
        l.gemo_set_obstacle_threshold.argtypes = [POINTER(OMap), c_float]
        l.gemo_add.restype = c_int
        l.gemo_add.argtypes = [POINTER(OMap), POINTER(OFrame), c_int, c_void_p, c_void_p, c_void_p, POINTER(c_longlong)]
        l.gemo_add_batch_mt.restype = c_longlong
        l.gemo_add_batch_mt.argtypes = [POINTER(OMap), c_int, POINTER(OFrame), c_void_p, POINTER(c_longlong), c_void_p, c_int]
        l.gemo_show.restype = c_int
        l.gemo_show.argtypes = [POINTER(OMap), c_void_p, c_void_p, c_double, c_double, POINTER(c_double), c_void_p, c_void_p, c_void_p, c_void_p]
        l.gemo_lidar_to_image.argtypes = [c_void_p, c_void_p, c_void_p]
        l.gemo_colorize.restype = c_int
        l.gemo_colorize.argtypes = [c_void_p, c_int, c_int, c_void_p, C.c_size_t, c_int, c_void_p, c_void_p]
        l.gemo_motion_init.argtypes = [POINTER(OMotion), c_double]
        l.gemo_motion_update.restype = c_double
        l.gemo_motion_update.argtypes = [POINTER(OMotion)] + [POINTER(c_double)] * 4
        _lib = l
    return _lib


def _vp(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class OracleMap:
    """Same call surface as gem_amd.ElevationMap, computed by the C oracle on the CPU."""

    LAYERS = ("elevation", "variance", "intensity", "traver", "lowest")
    INT_LAYERS = {"color_r": "colorR", "color_g": "colorG", "color_b": "colorB"}

    def __init__(self, length: int, resolution: float, mahalanobis_threshold: float = 5.0, variance_floor: float = 1e-4):
        self._l = lib()
        self._m = self._l.gemo_create(int(length), float(resolution), float(mahalanobis_threshold), float(variance_floor))
        self.length, self.resolution = int(length), float(resolution)
        self.last_counts = (0, 0)

    def close(self):
        if getattr(self, "_m", None):
            self._l.gemo_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def move(self, position):
        pos = (c_float * 3)(*[float(v) for v in position])
        c = (c_float * 2)(); s = (c_int * 2)(); a = (c_float * 2)()
        self._l.gemo_move(self._m, pos, c, s, a)
        return np.array(c[:], np.float32), np.array(s[:], np.int32), np.array(a[:], np.float32)

    def pose(self):
        m = self._m.contents
        return np.array(m.center[:], np.float32), np.array(m.start[:], np.int32)

    def points_to_index(self, x, y) -> int:
        return self._l.gemo_points_to_index(self._m, c_float(x), c_float(y))

    def points_to_map_index(self, x, y) -> int:
        return self._l.gemo_points_to_map_index(self._m, c_float(x), c_float(y))

    def process_points(self, frame, x, y, z, orig_index=None, write_back_xyz: bool = False):
        n = int(np.asarray(x).size)
        xa = np.array(x, np.float32, copy=True).reshape(-1); ya = np.array(y, np.float32, copy=True).reshape(-1)
        za = np.array(z, np.float32, copy=True).reshape(-1)
        oi = None if orig_index is None else np.ascontiguousarray(orig_index, np.int32)
        out = {"index": np.empty(n, np.int32), "var": np.empty(n, np.float32), "x_ts": np.empty(n, np.float32),
               "y_ts": np.empty(n, np.float32), "height": np.empty(n, np.float32)}
        p = frame.to_struct(OFrame)
        out["accepted"] = self._l.gemo_process_points(self._m, C.byref(p), n, _vp(xa), _vp(ya), _vp(za), _vp(oi),
                                                      _vp(out["index"]), _vp(out["var"]), _vp(out["x_ts"]),
                                                      _vp(out["y_ts"]), _vp(out["height"]))
        if write_back_xyz:
            out["x"], out["y"], out["z"] = xa, ya, za
        return out

    def _fuse(self, fn, index, height, var, R, G, B, intensity):
        n = int(np.asarray(index).size)
        c = lambda a, t: None if a is None else np.ascontiguousarray(a, t)
        i, h, v = c(index, np.int32), c(height, np.float32), c(var, np.float32)
        r, g, b, I = c(R, np.int32), c(G, np.int32), c(B, np.int32), c(intensity, np.float32)
        fn(self._m, n, _vp(i), _vp(r), _vp(g), _vp(b), _vp(I), _vp(h), _vp(v))

    def fuse(self, index, height, var, R=None, G=None, B=None, intensity=None):
        self._fuse(self._l.gemo_fuse, index, height, var, R, G, B, intensity)

    def fuse_literal(self, index, height, var, R=None, G=None, B=None, intensity=None):
        self._fuse(self._l.gemo_fuse_literal, index, height, var, R, G, B, intensity)

    def add(self, frame, xyzi, rgb=None, orig_index=None):
        a = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        r = None if rgb is None else np.ascontiguousarray(rgb, np.uint32)
        o = None if orig_index is None else np.ascontiguousarray(orig_index, np.int32)
        counts = (c_longlong * 2)()
        p = frame.to_struct(OFrame)
        self._l.gemo_add(self._m, C.byref(p), a.shape[0], _vp(a), _vp(r), _vp(o), counts)
        self.last_counts = (int(counts[0]), int(counts[1]))

    def add_batch_mt(self, frames, xyzi, offsets, var_updates=None, nthreads: int = 0) -> int:
        """All-core form (gem_oracle_mt.c): for s: mapvar_update(var_updates[s]); add(frames[s], xyzi[offsets[s]:offsets[s+1]]),
        cells partitioned into row strips over `nthreads` threads (0 = all host cores).  Same bits as the sequential calls."""
        import os
        ns = len(frames)
        fr = (OFrame * ns)(*[f.to_struct(OFrame) for f in frames])
        a = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        off = (c_longlong * (ns + 1))(*[int(v) for v in offsets])
        vu = None if var_updates is None else np.ascontiguousarray(var_updates, np.float32)
        nt = int(nthreads) if nthreads and nthreads > 0 else (os.cpu_count() or 1)
        acc = self._l.gemo_add_batch_mt(self._m, ns, fr, _vp(a), off, _vp(vu), nt)
        if acc < 0:
            raise RuntimeError("gemo_add_batch_mt failed (threads / memory)")
        return int(acc)

    def mapvar_update(self, u: float):
        self._l.gemo_mapvar_update(self._m, c_float(u))

    def map_optmove(self, opt_xy, height_update: float):
        """Map_optmove (gpu_process.cu:1215-1233): returns the aligned centre."""
        p = (c_float * 2)(float(opt_xy[0]), float(opt_xy[1])); out = (c_float * 2)()
        self._l.gemo_map_optmove(self._m, p, c_float(height_update), out)
        return np.array([out[0], out[1]], np.float32)

    def map_closeloop(self, xy, height_update: float):
        p = (c_float * 2)(float(xy[0]), float(xy[1]))
        self._l.gemo_map_closeloop(self._m, p, c_float(height_update))

    def raytracing(self):
        """Raytracing (gpu_process.cu:1304-1318): visibility clean-up, then lowest = 10 everywhere."""
        self._l.gemo_raytracing(self._m)

    def set_obstacle_threshold(self, t: float):
        self._l.gemo_set_obstacle_threshold(self._m, c_float(t))

    def map_feature(self):
        """G_Mapfeature (gpu_process.cu:549-670): returns dict(rough, slope, traver) and updates the traver layer."""
        n = self.length * self.length
        out = {k: np.zeros(n, np.float32) for k in ("rough", "slope", "traver")}
        self._l.gemo_map_feature(self._m, _vp(out["rough"]), _vp(out["slope"]), _vp(out["traver"]))
        return {k: v.reshape(self.length, self.length) for k, v in out.items()}

    def show(self, rough=None, slope=None, map_length=None, resolution=None, position=None):
        """ElevationMap::show's cell loop (ElevationMap.cpp:85-149): dict(visual [9, L, L] column-major layers with NaN, points_xyz,
        points_rgb, image_bgr).  rough / slope: the arrays Map_feature returned (default zeros)."""
        L = self.length
        res = float(np.float32(self.resolution)) if resolution is None else float(resolution)
        length = L * res if map_length is None else float(map_length)
        c = self.pose()[0] if position is None else position
        pos = (c_double * 2)(float(c[0]), float(c[1]))
        visual = np.empty((9, L * L), np.float32); xyz = np.empty((L * L, 3), np.float32); rgb = np.empty((L * L, 3), np.uint8)
        img = np.empty((L, L, 3), np.uint8)
        r = None if rough is None else np.ascontiguousarray(rough, np.float32)
        s = None if slope is None else np.ascontiguousarray(slope, np.float32)
        n = self._l.gemo_show(self._m, _vp(r), _vp(s), length, res, pos, _vp(visual), _vp(xyz), _vp(rgb), _vp(img))
        return {"visual": visual.reshape(9, L, L), "points_xyz": xyz[:n].copy(), "points_rgb": rgb[:n].copy(), "image_bgr": img, "count": n}

    def layer(self, name: str) -> np.ndarray:
        m = self._m.contents
        n = self.length * self.length
        if name == "lowest_scan_point":
            name = "lowest"
        if name in self.INT_LAYERS:
            return np.ctypeslib.as_array(getattr(m, self.INT_LAYERS[name]), (n,)).reshape(self.length, self.length).copy()
        return np.ctypeslib.as_array(getattr(m, name), (n,)).reshape(self.length, self.length).copy()

    def set_layer(self, name: str, values):
        m = self._m.contents
        n = self.length * self.length
        if name in self.INT_LAYERS:
            np.ctypeslib.as_array(getattr(m, self.INT_LAYERS[name]), (n,))[:] = np.asarray(values, np.int32).reshape(-1)
        else:
            np.ctypeslib.as_array(getattr(m, name), (n,))[:] = np.asarray(values, np.float32).reshape(-1)


def lidar_to_image(tcamera, tlidar) -> np.ndarray:
    """P_lidar2img = T.camera (3x4) * T.lidar (4x4), EMg.cpp:343."""
    a = np.ascontiguousarray(tcamera, np.float64).reshape(3, 4); b = np.ascontiguousarray(tlidar, np.float64).reshape(4, 4)
    out = np.empty((3, 4), np.float64)
    lib().gemo_lidar_to_image(_vp(a), _vp(b), _vp(out))
    return out


def colorize(P, image_bgr, xyzi):
    """The colourisation loop of ElevationMapping::Callback (EMg.cpp:349-381), point by point.  Returns dict(rgb uint32 [n]
    0x00RRGGBB, xyzi with intensity zeroed outside the image, image = the drawn-on copy, count)."""
    P = np.ascontiguousarray(P, np.float64).reshape(3, 4)
    img = np.ascontiguousarray(image_bgr, np.uint8).copy()
    assert img.ndim == 3 and img.shape[2] == 3
    pts = np.ascontiguousarray(xyzi, np.float32).copy()
    rgb = np.zeros(pts.shape[0], np.uint32)
    n = lib().gemo_colorize(_vp(P), img.shape[1], img.shape[0], _vp(img), img.strides[0], pts.shape[0], _vp(pts), _vp(rgb))
    return {"rgb": rgb, "xyzi": pts, "image": img, "count": n}


class OracleMotion:
    def __init__(self, covariance_scale: float = 1.0):
        self._l = lib()
        self._s = OMotion()
        self._l.gemo_motion_init(C.byref(self._s), covariance_scale)

    def compute(self, position, R_IB, cov6x6, map_rotation=None) -> float:
        d = lambda a, n: (c_double * n)(*np.asarray(a, np.float64).reshape(-1).tolist())
        Rm = np.eye(3) if map_rotation is None else map_rotation
        return float(self._l.gemo_motion_update(C.byref(self._s), d(position, 3), d(R_IB, 9), d(cov6x6, 36), d(Rm, 9)))
