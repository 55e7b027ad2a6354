"""gem_amd -- MI355X (gfx950) implementation of GEM's point-cloud -> elevation-grid hot path.

The product is libgem_hip.so (hand-written HIP kernels behind the C ABI of include/gem_hip.h);
this package holds its sources (csrc/), the in-tree build (build.py), the ctypes binding (_lib.py)
and a Python mirror of the reference's host interface for that path (api.py).
"""
from .api import (ElevationMap, Frame, GemError, RejectFilter, RobotMotionMapUpdater, SensorModel,  # noqa: F401
                  SensorProcessor)

__all__ = ["ElevationMap", "Frame", "GemError", "RejectFilter", "RobotMotionMapUpdater", "SensorModel",
           "SensorProcessor"]
