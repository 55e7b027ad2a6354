"""ctypes binding of libgem_hip.so (the C ABI declared in include/gem_hip.h).

There is no CPU fallback: if the library is missing it is built with hipcc; if it cannot be
loaded, or no HIP device is present at gem_create time, the error is raised to the caller.
"""
from __future__ import annotations

import ctypes as C
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_longlong, c_uint32, c_void_p

from . import build as _build

GEM_OK = 0

# layers / layouts / models (include/gem_hip.h)
LAYER_ELEVATION, LAYER_VARIANCE, LAYER_INTENSITY, LAYER_TRAVER, LAYER_LOWEST, \
    LAYER_COLOR_R, LAYER_COLOR_G, LAYER_COLOR_B, LAYER_ROUGH, LAYER_SLOPE = range(10)
LAYOUT_STORAGE_ROWMAJOR, LAYOUT_GRIDMAP_COLMAJOR_NAN = 0, 1
MODEL_LASER, MODEL_STRUCTURED_LIGHT, MODEL_STEREO, MODEL_PERFECT = range(4)


class MapConfig(C.Structure):
    _fields_ = [("length", c_int), ("resolution", c_float), ("mahalanobis_threshold", c_float),
                ("variance_floor", c_float), ("obstacle_threshold", c_float),
                ("strip_row0", c_int), ("strip_rows", c_int), ("device", c_int)]


class RejectFilter(C.Structure):
    _fields_ = [("enabled", c_int), ("box_x", c_float), ("box_y", c_float), ("band_y", c_float), ("plane_y", c_float)]


class FrameParams(C.Structure):
    _fields_ = [("T", c_float * 16), ("lower", c_double), ("upper", c_double), ("sensor_model", c_int),
                ("sensor_params", c_double * 8), ("sensor_jacobian", c_float * 3),
                ("rotation_variance", c_float * 9), ("C_SB_T", c_float * 9), ("P_mul_C_BM_T", c_float * 3),
                ("B_r_BS_skew", c_float * 9), ("filter", RejectFilter), ("original_width", c_int)]


class Camera(C.Structure):
    _fields_ = [("lidar_to_image", c_double * 12), ("width", c_int), ("height", c_int)]


class Stats(C.Structure):
    _fields_ = [("points_in", c_longlong), ("points_binned", c_longlong), ("cells_touched", c_longlong),
                ("ms_bin", c_float), ("ms_fuse", c_float), ("launches_bin", c_int), ("launches_fuse", c_int),
                ("ms_frame", c_float), ("launches_frame", c_int),
                ("ms_sort", c_float * 6), ("launches_sort", c_int), ("ms_walk", c_float), ("launches_walk", c_int)]


# every symbol include/gem_hip.h declares: (restype, argtypes)
SIGNATURES = {
    "gem_abi_version": (c_int, []),
    "gem_create": (c_int, [POINTER(MapConfig), POINTER(c_void_p)]),
    "gem_destroy": (None, [c_void_p]),
    "gem_last_error": (c_char_p, [c_void_p]),
    "gem_set_stream": (c_int, [c_void_p, c_void_p]),
    "gem_synchronize": (c_int, [c_void_p]),
    "gem_wait_event": (c_int, [c_void_p, c_void_p]),
    "gem_move": (c_int, [c_void_p, POINTER(c_float), POINTER(c_float), POINTER(c_int), POINTER(c_float)]),
    "gem_get_pose": (c_int, [c_void_p, POINTER(c_float), POINTER(c_int)]),
    "gem_process_points": (c_int, [c_void_p, POINTER(FrameParams), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gem_fuse": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gem_add": (c_int, [c_void_p, POINTER(FrameParams), c_int, c_void_p, c_void_p, c_void_p]),
    "gem_add_device": (c_int, [c_void_p, POINTER(FrameParams), c_int, c_void_p, c_void_p, c_void_p]),
    "gem_add_batch_device": (c_int, [c_void_p, c_int, POINTER(FrameParams), c_void_p, POINTER(c_longlong), POINTER(c_float)]),
    "gem_add_batch": (c_int, [c_void_p, c_int, POINTER(FrameParams), POINTER(c_void_p), POINTER(c_int), POINTER(c_float)]),
    "gem_mapvar_update": (c_int, [c_void_p, c_float]),
    "gem_get_layer": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "gem_set_layer": (c_int, [c_void_p, c_int, c_void_p]),
    "gem_layer_device_ptr": (c_int, [c_void_p, c_int, POINTER(c_void_p)]),
    "gem_map_feature": (c_int, [c_void_p] + [c_void_p] * 9),
    "gem_map_optmove": (c_int, [c_void_p, POINTER(c_float), c_float, POINTER(c_float)]),
    "gem_map_closeloop": (c_int, [c_void_p, POINTER(c_float), c_float]),
    "gem_set_lowest_tracking": (c_int, [c_void_p, c_int]),
    "gem_add_aos": (c_int, [c_void_p, POINTER(FrameParams), c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]),
    "gem_raytracing": (c_int, [c_void_p]),
    "gem_set_timing": (c_int, [c_void_p, c_int]),
    "gem_set_counting": (c_int, [c_void_p, c_int]),
    "gem_get_stats": (c_int, [c_void_p, POINTER(Stats), c_int]),
    "gem_comm_unique_id": (c_int, [c_void_p]),
    "gem_comm_init": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "gem_allgather_layers": (c_int, [c_void_p, c_int]),
    "gem_colorize": (c_int, [c_void_p, POINTER(Camera), c_int, c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "gem_colorize_device": (c_int, [c_void_p, POINTER(Camera), c_int, c_void_p, c_void_p, C.c_size_t, c_void_p]),
    "gem_show": (c_int, [c_void_p, c_double, c_double, POINTER(c_double), c_void_p, c_void_p, c_void_p, POINTER(c_int), c_void_p]),
    "gem_comm_init_tiles": (c_int, [c_void_p, c_void_p, c_int, c_int]),
    "gem_reserve": (c_int, [c_void_p, c_longlong, c_int, c_int]),
    "gem_get_strip": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "gem_add_sharded_device": (c_int, [c_void_p, c_int, POINTER(FrameParams), c_void_p, POINTER(c_longlong), c_int, c_int, c_int, POINTER(c_float)]),
    "gem_shard_sort_device": (c_int, [c_void_p, c_int, POINTER(FrameParams), c_void_p, POINTER(c_longlong), c_int, c_int, c_int, c_int, POINTER(c_int),
                                      POINTER(c_uint32), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    "gem_shard_fuse_device": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_uint32), POINTER(c_void_p), POINTER(c_uint32),
                                      c_int, POINTER(c_float)]),
}
# include/gem_hip_debug.h (tuning knobs / profiling aids, not part of the drop-in surface)
DEBUG_SIGNATURES = {
    "gem_debug_set": (c_int, [c_void_p, c_char_p, c_longlong]),
    "gem_debug_get": (c_int, [c_void_p, c_char_p, POINTER(c_longlong)]),
    "gem_debug_fuse_stamps": (c_int, [c_void_p, c_int, c_void_p, c_int]),
    "gem_comm_init_loopback": (c_int, [c_void_p, c_longlong, c_int, c_int, c_int]),
}

_lib = None


def load(rebuild_if_stale: bool = True) -> C.CDLL:
    """Load (building first if necessary) libgem_hip.so and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.build() if rebuild_if_stale else _build.LIB
    # A process that also uses PyTorch must end up with ONE HIP runtime / RCCL: torch ships its own
    # copies (torch/lib), and whichever libamdhip64 is mapped first serves both.  Import torch first
    # so device pointers, streams and events are interchangeable between torch and libgem_hip.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(str(path))
    for name, (res, args) in {**SIGNATURES, **DEBUG_SIGNATURES}.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def library_path() -> str:
    return str(_build.LIB)
