"""Spatial tiling of one elevation map across the GPUs of a node.

The map is split into row strips in STORAGE coordinates (rank r owns rows [L*r/W, L*(r+1)/W)), so
Move() -- which only rotates the circular buffer's start index -- never migrates data between
devices.  Every rank bins the whole cloud but fuses only the cells of its strip (cells are
independent given their ordered point lists, so the result is exactly the single-device one);
the fused strips are then exchanged with an all-gather:

  exchange="rccl"   gem_allgather_layers(): RCCL ncclAllGather / grouped ncclBroadcast over xGMI,
                    issued by the C ABI on the handle's stream (the product path)
  exchange="torch"  torch.distributed collectives on tensors aliasing the layers (NCCL == RCCL on
                    ROCm; gloo on CPU, which is how the host logic is tested without GPUs)
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import numpy as np


def strip_bounds(length: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [row0, row1) owned by `rank`; identical to gem_comm_init() in csrc/gem_capi.cpp."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (length * rank) // world, (length * (rank + 1)) // world


def all_strips(length: int, world: int):
    return [strip_bounds(length, world, r) for r in range(world)]


class _DeviceArray:
    """Zero-copy view of device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def exchange_strips_torch(tensors: Sequence, length: int, world: int, rank: int, group=None) -> None:
    """In-place all-gather of row strips: tensors are [L, L] (row-major) and rank r's rows are valid."""
    import torch.distributed as dist
    even = length % world == 0
    backend = dist.get_backend(group)
    for t in tensors:
        if even and backend == "nccl":
            r0, r1 = strip_bounds(length, world, rank)
            dist.all_gather_into_tensor(t, t[r0:r1], group=group)          # in place: input is output's own slice
        else:
            for src in range(world):                                        # uneven strips / gloo: one broadcast per owner
                r0, r1 = strip_bounds(length, world, src)
                if r1 > r0:
                    dist.broadcast(t[r0:r1], src=src, group=group)


class TiledElevationMap:
    """One rank's share of a map tiled over `world` ranks.  `make_map(length, resolution, strip=(row0, rows))`
    builds the local map (gem_amd.ElevationMap in production; tests inject a CPU stand-in)."""

    def __init__(self, length: int, resolution: float, rank: int, world: int, make_map: Optional[Callable] = None,
                 exchange: str = "rccl", unique_id: Optional[bytes] = None, **map_kwargs):
        self.length, self.resolution, self.rank, self.world = int(length), float(resolution), int(rank), int(world)
        self.row0, self.row1 = strip_bounds(length, world, rank)
        if make_map is None:
            from .api import ElevationMap
            make_map = ElevationMap
        self.map = make_map(length, resolution, strip=(self.row0, self.row1 - self.row0), **map_kwargs)
        self.exchange = exchange
        self._tensors = None
        if exchange == "rccl":
            if unique_id is None:
                raise ValueError("exchange='rccl' needs the ncclUniqueId created by rank 0 (ElevationMap.comm_unique_id())")
            self.map.comm_init(unique_id, world, rank)
        elif exchange != "torch":
            raise ValueError("exchange must be 'rccl' or 'torch'")

    # the map operations every rank performs identically
    def move(self, position):
        return self.map.move(position)

    def mapvar_update(self, u: float):
        self.map.mapvar_update(u)

    def add(self, frame, xyzi, **kw):
        self.map.add(frame, xyzi, **kw)          # the kernels drop points outside [row0, row1)

    def layer_tensors(self, names=("elevation", "variance")):
        """torch tensors aliasing the local map's layers (device memory, or host arrays for CPU stand-ins)."""
        import torch
        out = []
        for n in names:
            if hasattr(self.map, "layer_tensor"):
                out.append(self.map.layer_tensor(n))
            else:
                ptr = self.map.layer_device_ptr(n)
                typestr = "<i4" if n.startswith("color") else "<f4"
                out.append(torch.as_tensor(_DeviceArray(ptr, (self.length, self.length), typestr), device="cuda"))
        return out

    def allgather(self, with_attributes: bool = False, group=None) -> None:
        """Make every rank's copy of the fused layers complete."""
        if self.exchange == "rccl":
            self.map.allgather_layers(with_attributes)
            return
        names = ("elevation", "variance") + (("intensity", "color_r", "color_g", "color_b") if with_attributes else ())
        if hasattr(self.map, "synchronize"):
            self.map.synchronize()               # torch's stream does not know about the handle's stream
        exchange_strips_torch(self.layer_tensors(names), self.length, self.world, self.rank, group)

    def layer(self, name):
        return self.map.layer(name)
