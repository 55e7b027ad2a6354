"""Spatial tiling of one elevation map across the GPUs of a node.

The map is split into row strips in STORAGE coordinates (rank r owns rows [L*r/W, L*(r+1)/W)), so
Move() -- which only rotates the circular buffer's start index -- never migrates data between
devices.  Every rank bins the whole cloud but fuses only the cells of its strip (cells are
independent given their ordered point lists, so the result is exactly the single-device one);
the fused strips are then exchanged with an all-gather:

  exchange="rccl"   gem_allgather_layers(): grouped ncclSend / ncclRecv over xGMI, issued by the C ABI on the handle's
                    gather stream (the product path)
  exchange="loopback"  the same C code with the RCCL calls replaced by device-to-device copies: `world` handles of ONE
                    process on ONE device, one thread per handle (include/gem_hip_debug.h; how the GPU suite covers W > 1)
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


def tile_strip_rows(length: int, world: int):
    """Strip boundaries [world + 1] made of whole rows of 32 x 32-cell tiles; identical to gem_comm_init_tiles()."""
    tile_rows = (length + 31) // 32
    return [min(length, 32 * ((tile_rows * k) // world)) for k in range(world + 1)]


def shard_batch(offsets, world: int, rank: int):
    """Rank `rank`'s contiguous share [N r / W, N (r+1) / W) of a batch of sweeps (offsets[n_sweeps + 1] into the
    concatenated cloud): returns (first_global_sweep, local_offsets) -- the local sweeps are first_global_sweep ..
    first_global_sweep + len(local_offsets) - 2, the first / last possibly a part of a sweep that a neighbour also holds.
    Ranks hold ascending index ranges: rank order is input order."""
    off = [int(v) for v in offsets]
    n, ns = off[-1], len(off) - 1
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    if lo >= hi:
        return 0, [lo]
    first = next(s for s in range(ns) if off[s + 1] > lo)
    last = max(s for s in range(ns) if off[s] < hi)
    # (empty sweeps inside the range stay in the list: local sweep i IS global sweep first + i)
    return first, [max(off[first], lo)] + [min(max(off[s + 1], lo), hi) for s in range(first, last + 1)]


def first_point_in_sweep(offsets, first_global_sweep: int, local_offsets) -> int:
    """Index, inside its sweep, of a shard's first point (0 unless the shard begins in the middle of a sweep whose head the
    rank before it holds): the camera sensor models take the pixel row / column from it."""
    return int(local_offsets[0]) - int(offsets[first_global_sweep]) if len(local_offsets) > 1 else 0


def route_sorted_records(bounds_by_rank, rank: int):
    """Stage-B routing table.  bounds_by_rank[s][k] = first record of strip k in rank s's sorted records.  Returns
    (send, recv): send[d] = (begin, end) of what this rank sends to strip owner d; recv[s] = number of records it gets
    from rank s -- to be concatenated in ascending s, which is the input order of the points."""
    mine = bounds_by_rank[rank]
    world = len(bounds_by_rank)
    send = [(int(mine[d]), int(mine[d + 1])) for d in range(world)]
    recv = [int(bounds_by_rank[s][rank + 1]) - int(bounds_by_rank[s][rank]) for s in range(world)]
    return send, recv


class _DeviceArray:
    """Zero-copy view of device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def exchange_strips_torch(tensors: Sequence, length: int, world: int, rank: int, group=None, strip_rows=None) -> None:
    """In-place all-gather of row strips: tensors are [L, L] (row-major) and rank r's rows are valid.  strip_rows[world + 1]
    gives the strip boundaries (default: the even split of gem_comm_init)."""
    import torch.distributed as dist
    rows = list(strip_rows) if strip_rows is not None else [strip_bounds(length, world, r)[0] for r in range(world)] + [length]
    even = all(rows[k + 1] - rows[k] == rows[1] - rows[0] for k in range(world))
    backend = dist.get_backend(group)
    for t in tensors:
        if even and backend == "nccl":
            dist.all_gather_into_tensor(t, t[rows[rank]:rows[rank + 1]], group=group)      # in place: input is output's own slice
        else:
            for src in range(world):                                        # uneven strips / gloo: one broadcast per owner
                if rows[src + 1] > rows[src]:
                    dist.broadcast(t[rows[src]:rows[src + 1]], src=src, group=group)


class TiledElevationMap:
    """One rank's share of a map tiled over `world` ranks.  `make_map(length, resolution, strip=(row0, rows))`
    builds the local map (gem_amd.ElevationMap in production; tests inject a CPU stand-in).

    tile_strips=True makes the strips whole rows of 32 x 32-cell tiles, which add_sharded() needs (SURVEY 8e stage B: every
    rank bins only ITS share of the points; the sorted records travel to the strip owners)."""

    def __init__(self, length: int, resolution: float, rank: int, world: int, make_map: Optional[Callable] = None,
                 exchange: str = "rccl", unique_id: Optional[bytes] = None, tile_strips: bool = False, world_id: int = 0, **map_kwargs):
        self.length, self.resolution, self.rank, self.world = int(length), float(resolution), int(rank), int(world)
        self.tile_strips = bool(tile_strips)
        self.strip_rows = tile_strip_rows(length, world) if tile_strips else [strip_bounds(length, world, r)[0] for r in range(world)] + [length]
        self.row0, self.row1 = self.strip_rows[rank], self.strip_rows[rank + 1]
        if make_map is None:
            from .api import ElevationMap
            make_map = ElevationMap
        self.map = make_map(length, resolution, strip=(self.row0, self.row1 - self.row0), **map_kwargs)
        self.exchange = exchange
        self._tensors = None
        if exchange == "rccl":
            if unique_id is None:
                raise ValueError("exchange='rccl' needs the ncclUniqueId created by rank 0 (ElevationMap.comm_unique_id())")
            (self.map.comm_init_tiles if tile_strips else self.map.comm_init)(unique_id, world, rank)
        elif exchange == "loopback":
            self.map.comm_init_loopback(world_id, world, rank, tile_strips)
        elif exchange != "torch":
            raise ValueError("exchange must be 'rccl', 'loopback' or 'torch'")

    # -- stage B: the points sharded, the sorted records routed to the strip owners -----------------------------------------------
    def add_sharded(self, frames, xyzi, offsets, var_updates=None, group=None) -> None:
        """for s: Mapvar_update(var_updates[s]); add(frames[s], cloud s) on the tiled map, every rank working on its own
        contiguous share of the points (`xyzi` is the whole concatenated batch, or at least this rank's share, on this
        rank's device; `frames` / `offsets` / `var_updates` describe the whole batch and are identical on all ranks)."""
        if not self.tile_strips:
            raise ValueError("add_sharded needs tile_strips=True")
        n_global = len(frames)
        first, local = shard_batch(offsets, self.world, self.rank)
        local_frames = [frames[first + i] for i in range(len(local) - 1)]
        pb = self.map.pack_batch(local_frames, local, None)
        fp = first_point_in_sweep(offsets, first, local)
        if self.exchange in ("rccl", "loopback"):
            self.map.add_sharded(pb, xyzi, first, n_global, var_updates, fp)
            return
        # the exchange carried by torch.distributed (gloo on CPU stand-ins, NCCL == RCCL on devices)
        import torch
        import torch.distributed as dist
        bounds, hv, key = self.map.shard_sort_tensors(pb, xyzi, first, n_global, self.strip_rows, fp)
        gathered = [torch.zeros(self.world + 1, dtype=torch.int64, device=hv.device) for _ in range(self.world)]    # (NCCL carries device tensors only)
        dist.all_gather(gathered, torch.as_tensor(bounds, dtype=torch.int64).to(hv.device), group=group)
        send, recv = route_sorted_records([g.tolist() for g in gathered], self.rank)
        hv_in = [torch.empty((c, 2), dtype=hv.dtype, device=hv.device) for c in recv]
        key_in = [torch.empty((c,), dtype=key.dtype, device=key.device) for c in recv]
        hv_out = [hv[a:b].contiguous() for a, b in send]
        key_out = [key[a:b].contiguous() for a, b in send]
        if dist.get_backend(group) == "gloo":                      # gloo has no all_to_all: W rounds of pairwise exchanges
            for shift in range(self.world):
                dst, src = (self.rank + shift) % self.world, (self.rank - shift) % self.world
                if shift == 0:
                    hv_in[self.rank].copy_(hv_out[self.rank]); key_in[self.rank].copy_(key_out[self.rank])
                    continue
                ops = [dist.P2POp(dist.isend, hv_out[dst], dst, group), dist.P2POp(dist.isend, key_out[dst], dst, group),
                       dist.P2POp(dist.irecv, hv_in[src], src, group), dist.P2POp(dist.irecv, key_in[src], src, group)]
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
        else:
            dist.all_to_all(hv_in, hv_out, group=group)
            dist.all_to_all(key_in, key_out, group=group)
        self.map.shard_fuse_tensors(hv_in, key_in, n_global, var_updates)      # ascending source rank = input order

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
        if self.exchange in ("rccl", "loopback"):
            self.map.allgather_layers(with_attributes)
            return
        names = ("elevation", "variance") + (("intensity", "color_r", "color_g", "color_b") if with_attributes else ())
        if hasattr(self.map, "synchronize"):
            self.map.synchronize()               # torch's stream does not know about the handle's stream
        exchange_strips_torch(self.layer_tensors(names), self.length, self.world, self.rank, group, self.strip_rows)

    def layer(self, name):
        return self.map.layer(name)
