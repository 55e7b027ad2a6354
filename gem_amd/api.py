"""Host-side mirror of the reference's interface for the hot path, on top of the C ABI.

The reference's callers are ElevationMapping::processpoints / processmapcells / updateMapLocation
(elevation_mapping/src/ElevationMapping.cpp:254-300, 1001-1044), SensorProcessorBase::process /
GPUPointCloudprocess / readcomputerparam (src/sensor_processors/SensorProcessorBase.cpp:66-94,
126-211, 270-290) and RobotMotionMapUpdater::update (src/RobotMotionMapUpdater.cpp:42-90).
The classes below keep those names and argument meanings so the parity tests read like the
reference's call sites; the C++ twin for the unmodified ROS node is include/gem/gem.hpp.

All compute happens in libgem_hip.so on the GPU; there is no CPU path here.
"""
from __future__ import annotations

import atexit
import ctypes as C
import sys
import weakref
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _lib

# GridMap layer names of the reference (ElevationMap.cpp:43-44) -> device layers (gpu_process.cu:20-28)
LAYER_BY_NAME = {
    "elevation": _lib.LAYER_ELEVATION, "variance": _lib.LAYER_VARIANCE, "intensity": _lib.LAYER_INTENSITY,
    "traver": _lib.LAYER_TRAVER, "lowest_scan_point": _lib.LAYER_LOWEST, "lowest": _lib.LAYER_LOWEST,
    "color_r": _lib.LAYER_COLOR_R, "color_g": _lib.LAYER_COLOR_G, "color_b": _lib.LAYER_COLOR_B,
    "rough": _lib.LAYER_ROUGH, "slope": _lib.LAYER_SLOPE,
}
_INT_LAYERS = {_lib.LAYER_COLOR_R, _lib.LAYER_COLOR_G, _lib.LAYER_COLOR_B}


class GemError(RuntimeError):
    pass


# handles still alive at interpreter exit are destroyed before the HIP runtime is torn down
_live_maps: "weakref.WeakSet" = weakref.WeakSet()


@atexit.register
def _close_live_maps() -> None:
    for m in list(_live_maps):
        try:
            m.close()
        except Exception:
            pass


def _is_device_tensor(x) -> bool:
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def _host_ptr(a: Optional[np.ndarray], dtype, n: Optional[int] = None):
    """contiguous numpy array of `dtype` -> (keepalive, void*)"""
    if a is None:
        return None, None
    arr = np.ascontiguousarray(a, dtype=dtype)
    if n is not None and arr.size != n:
        raise ValueError(f"expected {n} elements, got {arr.size}")
    return arr, arr.ctypes.data_as(C.c_void_p)


def skew(v) -> np.ndarray:
    """kindr::getSkewMatrixFromVector (same matrix as gpu_process.cu:302-307)."""
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


@dataclass
class SensorModel:
    """sensor_processor/* parameters (config/sensor_processors/*.yaml)."""
    kind: int = _lib.MODEL_LASER
    params: Sequence[float] = (0.018, 0.0006, 0.0015)       # velodyne.yaml: min_radius, beam_angle, beam_constant
    ignore_points_above: float = float("inf")               # SensorProcessorBase.cpp:61
    ignore_points_below: float = float("-inf")              # SensorProcessorBase.cpp:62
    original_width: int = 0

    @staticmethod
    def velodyne() -> "SensorModel":                         # velodyne.yaml:4-9
        return SensorModel(_lib.MODEL_LASER, (0.018, 0.0006, 0.0015), 0.8, -5.0)

    @staticmethod
    def realsense_d435() -> "SensorModel":                   # realsense_d435.yaml:4-11
        return SensorModel(_lib.MODEL_STRUCTURED_LIGHT, (0.000611, 0.003587, 0.3515, 0.0, 1.0, 0.01576))

    @staticmethod
    def perfect() -> "SensorModel":
        return SensorModel(_lib.MODEL_PERFECT, ())


@dataclass
class RejectFilter:
    """The hard-coded sensor-frame filter of gpu_process.cu:393; reference() reproduces it."""
    enabled: bool = False
    box_x: float = 1.5
    box_y: float = 1.5
    band_y: float = 1.0
    plane_y: float = 0.0

    @staticmethod
    def reference() -> "RejectFilter":
        return RejectFilter(True)


@dataclass
class Frame:
    """Per-frame constants as plain numpy data (what GPUPointCloudprocess derives, SPB.cpp:171-208)."""
    T: np.ndarray                                            # 4x4 float32, sensor -> map
    lower: float
    upper: float
    model: SensorModel
    sensor_jacobian: np.ndarray
    rotation_variance: np.ndarray = field(default_factory=lambda: np.zeros((3, 3), np.float32))
    C_SB_T: np.ndarray = field(default_factory=lambda: np.eye(3, dtype=np.float32))
    P_mul_C_BM_T: np.ndarray = field(default_factory=lambda: np.array([0, 0, 1], np.float32))
    B_r_BS_skew: np.ndarray = field(default_factory=lambda: np.zeros((3, 3), np.float32))
    filter: RejectFilter = field(default_factory=RejectFilter)

    def __setattr__(self, name, value):
        object.__setattr__(self, name, value)
        if name != "_cache":
            object.__setattr__(self, "_cache", {})

    def to_struct(self, cls=_lib.FrameParams):
        """Fill a ctypes struct with the gem_frame_params field layout (also used, with the oracle's
        own struct class, by the tests).  Cached until a field is reassigned."""
        cache = self.__dict__.setdefault("_cache", {})
        if cls in cache:
            return cache[cls]
        p = cache[cls] = cls()
        p.T[:] = np.asarray(self.T, np.float32).reshape(16).tolist()
        p.lower, p.upper = float(self.lower), float(self.upper)
        p.sensor_model = int(self.model.kind)
        sp = list(self.model.params) + [0.0] * (8 - len(self.model.params))
        if hasattr(p, "sensor_params"):
            p.sensor_params[:] = sp
        else:
            p.sp[:] = sp
        p.sensor_jacobian[:] = np.asarray(self.sensor_jacobian, np.float32).reshape(3).tolist()
        p.rotation_variance[:] = np.asarray(self.rotation_variance, np.float32).reshape(9).tolist()
        p.C_SB_T[:] = np.asarray(self.C_SB_T, np.float32).reshape(9).tolist()
        p.P_mul_C_BM_T[:] = np.asarray(self.P_mul_C_BM_T, np.float32).reshape(3).tolist()
        p.B_r_BS_skew[:] = np.asarray(self.B_r_BS_skew, np.float32).reshape(9).tolist()
        if hasattr(p, "filter"):
            p.filter.enabled = int(self.filter.enabled)
            p.filter.box_x, p.filter.box_y = self.filter.box_x, self.filter.box_y
            p.filter.band_y, p.filter.plane_y = self.filter.band_y, self.filter.plane_y
        else:
            p.filter_on = int(self.filter.enabled)
            p.filter_box_x, p.filter_box_y = self.filter.box_x, self.filter.box_y
            p.filter_band_y, p.filter_plane_y = self.filter.band_y, self.filter.plane_y
        p.original_width = int(self.model.original_width)
        return p


class SensorProcessor:
    """SensorProcessorBase (SensorProcessorBase.hpp:52-180) for the GPU path.

    update_transformations() takes what the three TF lookups of SPB.cpp:97-124 return:
    T_map_sensor (map <- sensor), T_base_sensor (base <- sensor), T_map_base (map <- base), 4x4 doubles.
    """

    def __init__(self, model: SensorModel, reject_filter: Optional[RejectFilter] = None,
                 rotation_variance: Optional[np.ndarray] = None):
        self.model = model
        self.filter = reject_filter or RejectFilter()
        self.rotation_variance = np.zeros((3, 3), np.float32) if rotation_variance is None else np.asarray(rotation_variance, np.float32)
        self.update_transformations(np.eye(4), np.eye(4), np.eye(4))

    def update_transformations(self, T_map_sensor, T_base_sensor, T_map_base) -> None:
        self.T_map_sensor = np.asarray(T_map_sensor, np.float64)
        Tbs = np.asarray(T_base_sensor, np.float64)
        Tmb = np.asarray(T_map_base, np.float64)
        self.rotation_base_to_sensor = Tbs[:3, :3].copy()           # SPB.cpp:110
        self.translation_base_to_sensor = Tbs[:3, 3].copy()         # SPB.cpp:111
        self.rotation_map_to_base = Tmb[:3, :3].copy()              # SPB.cpp:116
        self.translation_map_to_base = Tmb[:3, 3].copy()            # SPB.cpp:117

    def frame(self) -> Frame:
        """readcomputerparam (SPB.cpp:270-290) + the casts of GPUPointCloudprocess (SPB.cpp:171-184)."""
        C_BM_T = self.rotation_map_to_base.T
        C_SB_T = self.rotation_base_to_sensor.T
        sensor_jacobian = (C_BM_T @ C_SB_T).astype(np.float32)[2, :]          # :275 e_z^T (double product, float cast)
        C_BM_T_f = C_BM_T.astype(np.float32)
        P_mul = C_BM_T_f[2, :]                                                 # :281-282
        z_base = float(self.translation_map_to_base[2])
        return Frame(
            T=self.T_map_sensor.astype(np.float32),                           # :175-179
            lower=z_base + self.model.ignore_points_below,                    # :183
            upper=z_base + self.model.ignore_points_above,                    # :184
            model=self.model,
            sensor_jacobian=sensor_jacobian,
            rotation_variance=self.rotation_variance,
            C_SB_T=C_SB_T.astype(np.float32),                                 # :283
            P_mul_C_BM_T=P_mul,
            B_r_BS_skew=skew(self.translation_base_to_sensor.astype(np.float32)).astype(np.float32),   # :284
            filter=self.filter,
        )

    def process(self, elevation_map: "ElevationMap", x, y, z, orig_index=None):
        """SensorProcessorBase::process -> Process_points (SPB.cpp:66-94, 208): returns the
        per-point arrays the reference hands to Fuse."""
        return elevation_map.process_points(self.frame(), x, y, z, orig_index)


class PackedBatch:
    """ctypes arrays of one gem_add_batch_device call (ElevationMap.pack_batch)."""
    __slots__ = ("n", "frames", "offsets", "var_updates")


class ElevationMap:
    """The robot-centric map (device-resident; one libgem_hip handle).

    The reference's ElevationMap::add/fuse were deleted from the tree; the map now lives in
    gpu_process.cu's globals and is driven through Init_GPU_elevationmap / Move / Process_points /
    Fuse / Mapvar_update.  This class exposes exactly those operations plus the fused add().
    """

    # knobs applied to every new map through gem_debug_set (include/gem_hip_debug.h); the tests set this to drive all code paths
    default_debug: dict = {}
    base_debug: dict = {}              # applied before default_debug (the GPU tests run every case on both pipelines through this)

    def __init__(self, length: int, resolution: float, mahalanobis_threshold: float = 5.0,
                 variance_floor: float = 1e-4, strip: tuple = (0, 0), device: int = -1, obstacle_threshold: float = 0.7,
                 debug: Optional[dict] = None):
        self._lib = _lib.load()
        cfg = _lib.MapConfig(int(length), float(resolution), float(mahalanobis_threshold), float(variance_floor),
                             float(obstacle_threshold), int(strip[0]), int(strip[1]), int(device))
        h = C.c_void_p()
        rc = self._lib.gem_create(C.byref(cfg), C.byref(h))
        if rc != _lib.GEM_OK:
            raise GemError(f"gem_create failed ({rc}): {self._lib.gem_last_error(None).decode()}")
        self._h = h
        self.length = int(length)
        self.resolution = float(resolution)
        _live_maps.add(self)
        for k, v in {**type(self).base_debug, **type(self).default_debug, **(debug or {})}.items():
            self.debug_set(k, v)

    # -- plumbing ------------------------------------------------------------------------------
    def _check(self, rc: int, what: str) -> None:
        if rc != _lib.GEM_OK:
            raise GemError(f"{what} failed ({rc}): {self._lib.gem_last_error(self._h).decode()}")

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.gem_destroy(self._h)
            self._h = None

    def __del__(self):
        if sys.is_finalizing():          # too late to talk to the HIP runtime
            return
        try:
            self.close()
        except Exception:
            pass

    def debug_set(self, key: str, value: int) -> None:
        """Tuning / test knob (gem_debug_set, include/gem_hip_debug.h): selects between code paths that produce the same map."""
        self._check(self._lib.gem_debug_set(self._h, key.encode(), int(value)), f"gem_debug_set({key})")

    def debug_get(self, key: str) -> int:
        v = C.c_longlong()
        self._check(self._lib.gem_debug_get(self._h, key.encode(), C.byref(v)), f"gem_debug_get({key})")
        return int(v.value)

    def set_stream(self, hip_stream: Optional[int]) -> None:
        self._check(self._lib.gem_set_stream(self._h, C.c_void_p(hip_stream) if hip_stream else None), "gem_set_stream")

    def _hold(self, *tensors) -> None:
        """gem_add_device / gem_add_batch_device only ENQUEUE: the caller's device buffers must stay untouched until gem_synchronize
        (include/gem_hip.h).  A torch tensor that goes out of scope returns its memory to the caching allocator, which hands it to the
        next `.cuda()` at once -- on torch's stream, which knows nothing of the handle's streams.  So the Python twin keeps a reference
        to every device input until the handle is synchronised (tools/fuzz_parity.py passes temporaries; round 6 found that scenario)."""
        held = getattr(self, "_held", None)
        if not isinstance(held, dict):
            held = self._held = {}
        for t in tensors:
            if t is not None:
                held[id(t)] = t                      # (a stream that cycles through the same few tensors holds each once)
        if len(held) > 256:
            self.synchronize()

    def synchronize(self) -> None:
        self._check(self._lib.gem_synchronize(self._h), "gem_synchronize")
        self._held = {}                              # (the device inputs of the calls so far have been read)

    def wait_event(self, hip_event) -> None:
        """Everything enqueued from now on waits (on the device) for this event -- a hipEvent_t handle, or a torch.cuda.Event that
        has been recorded on the stream producing the next call's device buffers (gem_wait_event)."""
        h = getattr(hip_event, "cuda_event", hip_event)
        self._check(self._lib.gem_wait_event(self._h, C.c_void_p(int(h))), "gem_wait_event")

    # -- Move (ElevationMapping::updateMapLocation -> Move, EMg.cpp:1032) ------------------------
    def move(self, position):
        pos = (C.c_float * 3)(*[float(v) for v in position])
        c = (C.c_float * 2)(); s = (C.c_int * 2)(); a = (C.c_float * 2)()
        self._check(self._lib.gem_move(self._h, pos, c, s, a), "gem_move")
        return np.array(c[:], np.float32), np.array(s[:], np.int32), np.array(a[:], np.float32)

    def pose(self):
        c = (C.c_float * 2)(); s = (C.c_int * 2)()
        self._check(self._lib.gem_get_pose(self._h, c, s), "gem_get_pose")
        return np.array(c[:], np.float32), np.array(s[:], np.int32)

    # -- Process_points (SPB.cpp:208) ---------------------------------------------------------------
    def process_points(self, frame: Frame, x, y, z, orig_index=None, write_back_xyz: bool = False):
        n = int(np.asarray(x).size)
        xa = np.array(x, np.float32, copy=True).reshape(-1); ya = np.array(y, np.float32, copy=True).reshape(-1)
        za = np.array(z, np.float32, copy=True).reshape(-1)
        ok, okp = _host_ptr(orig_index, np.int32, n)
        out = {"index": np.empty(n, np.int32), "var": np.empty(n, np.float32), "x_ts": np.empty(n, np.float32),
               "y_ts": np.empty(n, np.float32), "height": np.empty(n, np.float32)}
        p = frame.to_struct()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self._lib.gem_process_points(self._h, C.byref(p), n, vp(xa), vp(ya), vp(za), okp, int(write_back_xyz),
                                                 vp(out["index"]), vp(out["var"]), vp(out["x_ts"]), vp(out["y_ts"]),
                                                 vp(out["height"])), "gem_process_points")
        if write_back_xyz:
            out["x"], out["y"], out["z"] = xa, ya, za
        return out

    # -- Fuse (EMg.cpp:280) --------------------------------------------------------------------------
    def fuse(self, index, height, var, R=None, G=None, B=None, intensity=None) -> None:
        n = int(np.asarray(index).size)
        ki, pi = _host_ptr(index, np.int32, n); kh, ph = _host_ptr(height, np.float32, n); kv, pv = _host_ptr(var, np.float32, n)
        kr, pr = _host_ptr(R, np.int32, n); kg, pg = _host_ptr(G, np.int32, n); kb, pb = _host_ptr(B, np.int32, n)
        kI, pI = _host_ptr(intensity, np.float32, n)
        self._check(self._lib.gem_fuse(self._h, n, pi, pr, pg, pb, pI, ph, pv), "gem_fuse")

    # -- the fused path: process + Fuse in one call (ElevationMapping::processpoints, EMg.cpp:254-283) --
    def add(self, frame: Frame, xyzi, rgb=None, orig_index=None) -> None:
        p = frame.to_struct()
        if _is_device_tensor(xyzi):
            n = int(xyzi.shape[0])
            if not xyzi.is_contiguous() or xyzi.element_size() != 4 or xyzi.numel() != 4 * n:
                raise ValueError("xyzi must be a contiguous float32 [N,4] device tensor")
            dp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
            self._check(self._lib.gem_add_device(self._h, C.byref(p), n, dp(xyzi), dp(rgb), dp(orig_index)), "gem_add_device")
            self._hold(xyzi, rgb, orig_index)
            return
        a = np.ascontiguousarray(xyzi, np.float32).reshape(-1, 4)
        n = a.shape[0]
        kr, pr = _host_ptr(rgb, np.uint32, n); ko, po = _host_ptr(orig_index, np.int32, n)
        self._check(self._lib.gem_add(self._h, C.byref(p), n, a.ctypes.data_as(C.c_void_p), pr, po), "gem_add")

    def add_aos(self, frame: Frame, points: np.ndarray, off_x: int = 0, off_y: int = 4, off_z: int = 8, off_intensity: int = 24,
                off_rgb: int = 16) -> None:
        """The cloud as an array of point structs (numpy structured array or [n, step] bytes), default offsets =
        PointXYZRGBICT (PointXYZRGBICT.hpp:28-46); unpacked on the device (gem_add_aos)."""
        a = np.ascontiguousarray(points)
        n, step = a.shape[0], a.dtype.itemsize * (int(np.prod(a.shape[1:])) if a.ndim > 1 else 1)
        p = frame.to_struct()
        self._check(self._lib.gem_add_aos(self._h, C.byref(p), n, a.ctypes.data_as(C.c_void_p), step, off_x, off_y, off_z,
                                          off_intensity, off_rgb), "gem_add_aos")

    @staticmethod
    def pack_batch(frames: Sequence[Frame], offsets, var_updates=None) -> "PackedBatch":
        """The C-ABI arrays of a batched call, built once (32 frames cost ~0.3 ms of ctypes conversion per call otherwise)."""
        ns = len(frames)
        pb = PackedBatch()
        pb.n = ns
        pb.frames = (_lib.FrameParams * ns)(*[f.to_struct() for f in frames])
        pb.offsets = (C.c_longlong * (ns + 1))(*[int(v) for v in offsets])
        pb.var_updates = (C.c_float * ns)(*[float(v) for v in var_updates]) if var_updates is not None else None
        return pb

    def add_batch(self, frames, xyzi_device, offsets=None, var_updates=None) -> None:
        """BASELINE config 4: for each sweep s: Mapvar_update(var_updates[s]); add(frames[s], cloud s).
        `frames` is a sequence of Frame (with `offsets` [, `var_updates`]) or a PackedBatch from pack_batch()."""
        pb = frames if isinstance(frames, PackedBatch) else self.pack_batch(frames, offsets, var_updates)
        if not _is_device_tensor(xyzi_device):
            raise ValueError("add_batch takes a device-resident float32 [N,4] tensor")
        self._check(self._lib.gem_add_batch_device(self._h, pb.n, pb.frames, C.c_void_p(xyzi_device.data_ptr()), pb.offsets,
                                                   pb.var_updates), "gem_add_batch_device")
        self._hold(xyzi_device)

    def add_batch_host(self, frames, clouds, var_updates=None) -> None:
        """The same from HOST memory (gem_add_batch): `clouds` is a sequence of float32 [n_s, 4] numpy arrays, one per
        sweep; `frames` a sequence of Frame or a PackedBatch (whose offsets are not used: the arrays carry their own lengths)."""
        arrs = [np.ascontiguousarray(c, np.float32).reshape(-1, 4) for c in clouds]
        ns = len(arrs)
        if isinstance(frames, PackedBatch):
            if frames.n != ns:
                raise ValueError("one cloud per frame")
            fr, vu = frames.frames, frames.var_updates
        else:
            if len(frames) != ns:
                raise ValueError("one cloud per frame")
            fr = (_lib.FrameParams * ns)(*[f.to_struct() for f in frames])
            vu = (C.c_float * ns)(*[float(v) for v in var_updates]) if var_updates is not None else None
        ptrs = (C.c_void_p * ns)(*[a.ctypes.data for a in arrs])
        counts = (C.c_int * ns)(*[a.shape[0] for a in arrs])
        self._check(self._lib.gem_add_batch(self._h, ns, fr, ptrs, counts, vu), "gem_add_batch")

    # -- Mapvar_update (RMU.cpp:81) ------------------------------------------------------------------
    def reserve(self, max_points: int, max_sweeps: int = 1, with_colours: bool = False) -> None:
        """Pre-size the device arenas for the largest pass to come (gem_reserve): no pass within these bounds allocates afterwards."""
        self._check(self._lib.gem_reserve(self._h, int(max_points), int(max_sweeps), int(with_colours)), "gem_reserve")

    def mapvar_update(self, var_update: float) -> None:
        self._check(self._lib.gem_mapvar_update(self._h, float(var_update)), "gem_mapvar_update")

    # -- loop-closure re-anchoring (Map_optmove, EMg.cpp:1020; Map_closeloop) -------------------------------------
    def map_optmove(self, opt_xy, height_update: float) -> np.ndarray:
        p = (C.c_float * 2)(float(opt_xy[0]), float(opt_xy[1])); out = (C.c_float * 2)()
        self._check(self._lib.gem_map_optmove(self._h, p, float(height_update), out), "gem_map_optmove")
        return np.array([out[0], out[1]], np.float32)

    def map_closeloop(self, xy, height_update: float) -> None:
        p = (C.c_float * 2)(float(xy[0]), float(xy[1]))
        self._check(self._lib.gem_map_closeloop(self._h, p, float(height_update)), "gem_map_closeloop")

    # -- Map_feature (EMg.cpp:410): traversability stage on the fused map --------------------------------------
    # -- visibility clean-up (Raytracing, EMg.cpp:421) --------------------------------------------------------------
    def set_lowest_tracking(self, on: bool) -> None:
        """Maintain the lowest-scan-point layer (gpu_process.cu:430-439) in the fuse kernels; needed by raytracing()."""
        self._check(self._lib.gem_set_lowest_tracking(self._h, int(bool(on))), "gem_set_lowest_tracking")

    def raytracing(self) -> None:
        self._check(self._lib.gem_raytracing(self._h), "gem_raytracing")

    def map_feature(self, fetch: bool = True):
        """Computes the rough / slope / traver layers on the device (gem_map_feature).  With fetch=True returns
        dict(rough, slope, traver) as host arrays; with fetch=False it only enqueues the kernel."""
        L = self.length
        if not fetch:
            self._check(self._lib.gem_map_feature(self._h, *([None] * 9)), "gem_map_feature")
            return None
        out = {k: np.empty((L, L), np.float32) for k in ("rough", "slope", "traver")}
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self._lib.gem_map_feature(self._h, None, None, None, None, None, p(out["rough"]), p(out["slope"]),
                                              p(out["traver"]), None), "gem_map_feature")
        return out

    # -- the feed of ElevationMap::show (ElevationMap.cpp:85-149) ------------------------------------------------------------------
    def show(self, map_length: float = 0.0, resolution: float = 0.0, position=None):
        """visualMap_'s nine layers ([9, L, L], grid_map's column-major layout, NaN for cells without elevation / traversability),
        the coloured point cloud (device-compacted, reference order) and the orthomosaic, from the resident layers (gem_show)."""
        L = self.length
        visual = np.empty((9, L * L), np.float32); xyz = np.empty((L * L, 3), np.float32); rgb = np.empty((L * L, 3), np.uint8)
        img = np.empty((L, L, 3), np.uint8)
        n = C.c_int()
        pos = None if position is None else (C.c_double * 2)(float(position[0]), float(position[1]))
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self._lib.gem_show(self._h, float(map_length), float(resolution), pos, vp(visual), vp(xyz), vp(rgb), C.byref(n), vp(img)), "gem_show")
        k = int(n.value)
        return {"visual": visual.reshape(9, L, L), "points_xyz": xyz[:k].copy(), "points_rgb": rgb[:k].copy(), "image_bgr": img, "count": k}

    # -- the step in front of the path: input colourisation (EMg.cpp:349-381) -------------------------------
    @staticmethod
    def lidar_to_image(tcamera, tlidar) -> np.ndarray:
        """P_lidar2img = T.camera (3x4) * T.lidar (4x4) in double, sums over k = 0..3 in order (EMg.cpp:343)."""
        a = np.asarray(tcamera, np.float64).reshape(3, 4); b = np.asarray(tlidar, np.float64).reshape(4, 4)
        out = np.empty((3, 4), np.float64)
        for r in range(3):
            for c in range(4):
                acc = a[r, 0] * b[0, c]
                for k in range(1, 4):
                    acc = acc + a[r, k] * b[k, c]
                out[r, c] = acc
        return out

    def colorize(self, lidar_to_image, image_bgr, xyzi):
        """Colours of a cloud from a BGR8 camera image, with the reference's draw-while-sampling order dependence (gem_colorize).
        Host arrays: returns (rgb uint32 [n] 0x00RRGGBB, xyzi copy with intensity zeroed outside the image).  Device tensors
        (image uint8 [H, W, 3], xyzi float32 [n, 4], both on the handle's device): xyzi is updated IN PLACE and the returned rgb is
        an int32 tensor holding the same words; only enqueues."""
        cam = _lib.Camera()
        P = np.asarray(lidar_to_image, np.float64).reshape(12)
        for k in range(12):
            cam.lidar_to_image[k] = float(P[k])
        if hasattr(image_bgr, "is_cuda"):
            import torch
            if not (image_bgr.is_cuda and xyzi.is_cuda and image_bgr.dtype == torch.uint8 and xyzi.dtype == torch.float32):
                raise ValueError("colorize: device tensors must be uint8 [H, W, 3] and float32 [n, 4]")
            if not (xyzi.is_contiguous() and image_bgr.stride(2) == 1 and image_bgr.stride(1) == 3):
                raise ValueError("colorize: xyzi must be contiguous, image rows packed BGR")
            cam.height, cam.width = int(image_bgr.shape[0]), int(image_bgr.shape[1])
            rgb = torch.empty(xyzi.shape[0], dtype=torch.int32, device=xyzi.device)
            self._check(self._lib.gem_colorize_device(self._h, C.byref(cam), int(xyzi.shape[0]), C.c_void_p(xyzi.data_ptr()),
                                                      C.c_void_p(image_bgr.data_ptr()), int(image_bgr.stride(0)), C.c_void_p(rgb.data_ptr())),
                        "gem_colorize_device")
            return rgb, xyzi
        img = np.ascontiguousarray(image_bgr, np.uint8)
        pts = np.ascontiguousarray(xyzi, np.float32).copy()
        cam.height, cam.width = int(img.shape[0]), int(img.shape[1])
        rgb = np.zeros(pts.shape[0], np.uint32)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self._lib.gem_colorize(self._h, C.byref(cam), int(pts.shape[0]), vp(pts), vp(img), int(img.strides[0]), vp(rgb)), "gem_colorize")
        return rgb, pts

    # -- layers ----------------------------------------------------------------------------------------
    def layer(self, name_or_id, layout: int = _lib.LAYOUT_STORAGE_ROWMAJOR) -> np.ndarray:
        lid = LAYER_BY_NAME[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
        is_int = lid in _INT_LAYERS and layout == _lib.LAYOUT_STORAGE_ROWMAJOR
        out = np.empty((self.length, self.length), np.int32 if is_int else np.float32)
        self._check(self._lib.gem_get_layer(self._h, lid, layout, out.ctypes.data_as(C.c_void_p)), "gem_get_layer")
        self._held = {}                              # (a call that returns map data has everything before it behind it)
        if layout == _lib.LAYOUT_GRIDMAP_COLMAJOR_NAN:
            return out.T          # buffer holds column-major data: view it as [row, col]
        return out

    def set_layer(self, name_or_id, values) -> None:
        lid = LAYER_BY_NAME[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
        a = np.ascontiguousarray(values, np.int32 if lid in _INT_LAYERS else np.float32)
        if a.size != self.length * self.length:
            raise ValueError("layer size mismatch")
        self._check(self._lib.gem_set_layer(self._h, lid, a.ctypes.data_as(C.c_void_p)), "gem_set_layer")

    def layer_device_ptr(self, name_or_id) -> int:
        lid = LAYER_BY_NAME[name_or_id] if isinstance(name_or_id, str) else int(name_or_id)
        out = C.c_void_p()
        self._check(self._lib.gem_layer_device_ptr(self._h, lid, C.byref(out)), "gem_layer_device_ptr")
        return int(out.value)

    # -- stats --------------------------------------------------------------------------------------------
    def set_timing(self, on: bool) -> None:
        self._check(self._lib.gem_set_timing(self._h, int(on)), "gem_set_timing")

    def set_counting(self, on: bool) -> None:
        self._check(self._lib.gem_set_counting(self._h, int(on)), "gem_set_counting")

    def stats(self, reset: bool = False) -> dict:
        s = _lib.Stats()
        self._check(self._lib.gem_get_stats(self._h, C.byref(s), int(reset)), "gem_get_stats")
        return {k: (list(getattr(s, k)) if k == "ms_sort" else getattr(s, k)) for k, _ in _lib.Stats._fields_}

    # -- multi-GPU ------------------------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = _lib.load().gem_comm_unique_id(buf)
        if rc != _lib.GEM_OK:
            raise GemError(f"gem_comm_unique_id failed ({rc})")
        return buf.raw

    def comm_init(self, unique_id: bytes, nranks: int, rank: int) -> None:
        buf = C.create_string_buffer(unique_id, 128)
        self._check(self._lib.gem_comm_init(self._h, buf, int(nranks), int(rank)), "gem_comm_init")

    def allgather_layers(self, with_attributes: bool = False) -> None:
        self._check(self._lib.gem_allgather_layers(self._h, int(with_attributes)), "gem_allgather_layers")

    def comm_init_tiles(self, unique_id: bytes, nranks: int, rank: int) -> None:
        """Like comm_init, with strips made of whole rows of 32 x 32-cell tiles (what the sharded path needs)."""
        buf = C.create_string_buffer(unique_id, 128)
        self._check(self._lib.gem_comm_init_tiles(self._h, buf, int(nranks), int(rank)), "gem_comm_init_tiles")

    def comm_init_loopback(self, world_id: int, nranks: int, rank: int, tile_strips: bool = True) -> None:
        """Join a LOOPBACK communicator (include/gem_hip_debug.h): `nranks` handles of this process on one device, each driven by
        a thread of its own, run the multi-GPU path's code with device-to-device copies in place of the RCCL calls."""
        self._check(self._lib.gem_comm_init_loopback(self._h, int(world_id), int(nranks), int(rank), int(bool(tile_strips))), "gem_comm_init_loopback")

    def strip(self):
        r0, r1 = C.c_int(), C.c_int()
        self._check(self._lib.gem_get_strip(self._h, C.byref(r0), C.byref(r1)), "gem_get_strip")
        return int(r0.value), int(r1.value)

    # -- multi-GPU with the points sharded (SURVEY 8e stage B) --------------------------------------------------------------
    def add_sharded(self, pb: "PackedBatch", xyzi_device, first_global_sweep: int, n_global_sweeps: int, var_updates_global=None,
                    first_point_in_sweep: int = 0) -> None:
        """This rank's contiguous share of a batch (local sweeps of `pb`, numbered first_global_sweep.. globally; its first point is
        point first_point_in_sweep of its sweep): sort, RCCL exchange of the sorted records to the strip owners, walk in rank order
        (gem_add_sharded_device)."""
        vu = None if var_updates_global is None else (C.c_float * n_global_sweeps)(*[float(v) for v in var_updates_global])
        ptr = C.c_void_p(xyzi_device.data_ptr()) if pb.n else None
        self._check(self._lib.gem_add_sharded_device(self._h, pb.n, pb.frames, ptr, pb.offsets, int(first_global_sweep),
                                                     int(n_global_sweeps), int(first_point_in_sweep), vu), "gem_add_sharded_device")

    def shard_sort(self, pb: "PackedBatch", xyzi_device, first_global_sweep: int, n_global_sweeps: int, strip_rows, first_point_in_sweep: int = 0,
                   with_ranges: bool = False):
        """First half: returns (bounds [nstrips + 1], device pointer of the sorted {h, var} records, of their keys[, of the block ranges])."""
        ns = len(strip_rows) - 1
        rows = (C.c_int * (ns + 1))(*[int(v) for v in strip_rows])
        bounds = (C.c_uint32 * (ns + 1))()
        phv, pkey, prng = C.c_void_p(), C.c_void_p(), C.c_void_p()
        ptr = C.c_void_p(xyzi_device.data_ptr()) if pb.n else None
        self._check(self._lib.gem_shard_sort_device(self._h, pb.n, pb.frames, ptr, pb.offsets, int(first_global_sweep), int(n_global_sweeps),
                                                    int(first_point_in_sweep), ns, rows, bounds, C.byref(phv), C.byref(pkey), C.byref(prng)),
                    "gem_shard_sort_device")
        out = (np.array(bounds[:], np.int64), phv.value or 0, pkey.value or 0)
        return out + (prng.value or 0,) if with_ranges else out

    def shard_sort_tensors(self, pb, xyzi_device, first_global_sweep, n_global_sweeps, strip_rows, first_point_in_sweep: int = 0):
        """shard_sort with the sorted records as torch tensors aliasing the handle's arenas ([M, 2] int32 {h, var} bits, [M] int32
        keys); they stay valid until the next pass of this handle."""
        import torch
        from .tiling import _DeviceArray
        bounds, phv, pkey = self.shard_sort(pb, xyzi_device, first_global_sweep, n_global_sweeps, strip_rows, first_point_in_sweep)
        m = int(bounds[-1])
        if m == 0:
            dev = xyzi_device.device
            return bounds, torch.empty((0, 2), dtype=torch.int32, device=dev), torch.empty((0,), dtype=torch.int32, device=dev)
        hv = torch.as_tensor(_DeviceArray(phv, (m, 2), "<i4"), device="cuda")
        key = torch.as_tensor(_DeviceArray(pkey, (m,), "<i4"), device="cuda")
        return bounds, hv, key

    def shard_fuse_tensors(self, hv_list, key_list, n_global_sweeps, var_updates_global=None) -> None:
        import torch
        torch.cuda.synchronize()                     # the exchange ran on torch's stream, the walk runs on the handle's
        self._keep = (hv_list, key_list)             # until the walk has read them
        self.shard_fuse([t.data_ptr() if t.numel() else 0 for t in hv_list], [t.data_ptr() if t.numel() else 0 for t in key_list],
                        [int(t.shape[0]) for t in key_list], n_global_sweeps, var_updates_global)

    def shard_fuse(self, hv_ptrs, key_ptrs, counts, n_global_sweeps: int, var_updates_global=None, range_ptrs=None, bases=None) -> None:
        """Second half: walk this handle's strip through the sources (device pointers + record counts) in the order given; with
        range_ptrs / bases the sources' own block ranges replace the search (gem_hip.h)."""
        n = len(counts)
        a_hv = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in hv_ptrs])
        a_key = (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in key_ptrs])
        a_cnt = (C.c_uint32 * n)(*[int(c) for c in counts])
        a_rng = None if range_ptrs is None else (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in range_ptrs])
        a_base = None if bases is None else (C.c_uint32 * n)(*[int(b) for b in bases])
        vu = None if var_updates_global is None else (C.c_float * n_global_sweeps)(*[float(v) for v in var_updates_global])
        self._check(self._lib.gem_shard_fuse_device(self._h, n, a_hv, a_key, a_cnt, a_rng, a_base, int(n_global_sweeps), vu), "gem_shard_fuse_device")


class RobotMotionMapUpdater:
    """RobotMotionMapUpdater (RobotMotionMapUpdater.cpp:42-145): pose covariance -> scalar variance
    increment -> Mapvar_update.  Host-side doubles, exactly one float reaches the device."""

    def __init__(self, covariance_scale: float = 1.0):
        self.covariance_scale = float(covariance_scale)                 # RMU.cpp:25,38
        self.previous_reduced_covariance = np.zeros((4, 4))            # RMU.cpp:27
        self.previous_position = np.zeros(3)
        self.previous_rotation = np.eye(3)

    @staticmethod
    def _yaw_pitch(R):
        return np.arctan2(R[1, 0], R[0, 0]), np.arctan2(-R[2, 0], np.hypot(R[0, 0], R[1, 0]))

    @staticmethod
    def _rotation_vector_z(R):
        c = min(1.0, max(-1.0, 0.5 * (np.trace(R) - 1.0)))
        angle = np.arccos(c)
        wz = 0.5 * (R[1, 0] - R[0, 1])
        return wz if angle < 1e-12 else wz * angle / np.sin(angle)

    def compute(self, position, R_IB, covariance6x6, map_rotation=None) -> float:
        """Returns var_update (the float handed to Mapvar_update, RMU.cpp:80) and advances the state."""
        R = np.asarray(R_IB, np.float64); p = np.asarray(position, np.float64)
        cov = self.covariance_scale * np.asarray(covariance6x6, np.float64)
        Rm = np.eye(3) if map_rotation is None else np.asarray(map_rotation, np.float64)
        yaw, pitch = self._yaw_pitch(R)
        J = np.zeros((4, 6)); J[:3, :3] = np.eye(3)
        J[3, 3:] = [np.cos(yaw) * np.tan(pitch), np.sin(yaw) * np.tan(pitch), 1.0]
        reduced = J @ cov @ J.T                                         # RMU.cpp:107
        rz = self._rotation_vector_z(R)
        Rt = np.array([[np.cos(rz), -np.sin(rz), 0], [np.sin(rz), np.cos(rz), 0], [0, 0, 1.0]])
        v = self.previous_rotation.T @ (p - self.previous_position)    # RMU.cpp:121-123
        F = np.eye(4); F[:3, 3] = skew([0, 0, 1.0]) @ Rt @ v           # RMU.cpp:126-129
        G = np.zeros((4, 4)); G[3, 3] = 1.0; Gt = G.copy()
        G[:3, :3] = Rt.T; Gt[:3, :3] = Rt                              # RMU.cpp:132-137
        relative = G @ (reduced - F @ self.previous_reduced_covariance @ F.T) @ Gt   # RMU.cpp:140-142
        R_BM = R.T @ Rm                                                # RMU.cpp:62-63
        Jr = -R_BM.T                                                   # RMU.cpp:66
        upd = np.float32((Jr @ relative[:3, :3] @ Jr.T)[2, 2])         # RMU.cpp:69,80
        self.previous_reduced_covariance = reduced
        self.previous_position, self.previous_rotation = p.copy(), R.copy()
        return float(upd)

    def update(self, elevation_map: ElevationMap, position, R_IB, covariance6x6, map_rotation=None) -> float:
        u = self.compute(position, R_IB, covariance6x6, map_rotation)
        elevation_map.mapvar_update(u)                                  # RMU.cpp:81
        return u
