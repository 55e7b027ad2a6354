"""Seeded synthetic clouds for the BASELINE.json configurations.

All clouds are float32 XYZI in the SENSOR frame, generated on the host with
numpy.random.default_rng(seed); every generator also returns the frame constants.
The reference ships no data (its two demo bags are external downloads), so these are the inputs
of both the parity tests and the bench.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from .api import Frame, RejectFilter, SensorModel, SensorProcessor


def rot_zyx(yaw: float, pitch: float, roll: float) -> np.ndarray:
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    return Rz @ Ry @ Rx


def pose_matrix(x, y, z, yaw=0.0, pitch=0.0, roll=0.0) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = rot_zyx(yaw, pitch, roll)
    T[:3, 3] = [x, y, z]
    return T


@dataclass
class Workload:
    name: str
    length: int
    resolution: float
    clouds: List[np.ndarray]                 # one float32 [N,4] XYZI array per sweep (sensor frame)
    frames: List[Frame]
    var_updates: Optional[List[float]] = None
    map_position: Optional[np.ndarray] = None      # Move() target before fusing (None = stay at origin)
    rgb: Optional[List[np.ndarray]] = None

    @property
    def n_points(self) -> int:
        return int(sum(c.shape[0] for c in self.clouds))


def _frame_for(T_map_sensor: np.ndarray, model: SensorModel, flt: Optional[RejectFilter] = None) -> Frame:
    # sensor frame == base frame, as in the reference's demo configurations
    sp = SensorProcessor(model, flt)
    sp.update_transformations(T_map_sensor, np.eye(4), T_map_sensor)
    return sp.frame()


# ------------------------------------------------------------------------------------------------
def lidar_sweep(rng: np.random.Generator, T_map_sensor: np.ndarray, beams: int = 64, azimuth_steps: int = 2048,
                max_range: float = 80.0, range_sigma: float = 0.02, ground=None) -> np.ndarray:
    """64-beam spinning LiDAR over the ground plane z = ground(x, y) (default 0), beam-major order."""
    elev = np.deg2rad(np.linspace(-24.8, 2.0, beams))
    az = np.linspace(0.0, 2 * np.pi, azimuth_steps, endpoint=False)
    el, aa = np.meshgrid(elev, az, indexing="ij")
    d = np.stack([np.cos(el) * np.cos(aa), np.cos(el) * np.sin(aa), np.sin(el)], -1).reshape(-1, 3)
    R, t = T_map_sensor[:3, :3], T_map_sensor[:3, 3]
    dm = d @ R.T                                                   # ray directions in the map frame
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(dm[:, 2] < -1e-6, -t[2] / dm[:, 2], np.inf)   # hit with z = 0
    if ground is not None:                                         # one fixed-point refinement is plenty for gentle terrain
        hit = t[None, :] + np.minimum(s, max_range)[:, None] * dm
        with np.errstate(divide="ignore", invalid="ignore"):
            s = np.where(dm[:, 2] < -1e-6, (ground(hit[:, 0], hit[:, 1]) - t[2]) / dm[:, 2], np.inf)
    rng_ = np.minimum(s, max_range) + rng.normal(0.0, range_sigma, s.shape)
    pts = d * rng_[:, None]
    inten = rng.uniform(1.0, 255.0, (pts.shape[0], 1))
    return np.concatenate([pts, inten], 1).astype(np.float32)


def config_c1(seed: int = 1) -> Workload:
    """10 k-point planar cloud -> 200 x 200 @ 0.1 m, identity transform."""
    rng = np.random.default_rng(seed)
    n = 10_000
    xy = rng.uniform(-11.0, 11.0, (n, 2))
    z = 0.03 * xy[:, 0] + 0.005 * xy[:, 1] + rng.normal(0, 0.01, n)
    xyzi = np.concatenate([xy, z[:, None], rng.uniform(1, 255, (n, 1))], 1).astype(np.float32)
    return Workload("C1", 200, 0.1, [xyzi], [_frame_for(np.eye(4), SensorModel.velodyne())])


C2_POSE = dict(x=0.3, y=-0.2, z=1.73, yaw=np.deg2rad(10.0), pitch=np.deg2rad(1.0), roll=np.deg2rad(-0.5))


def config_c2(seed: int = 2, reference_filter: bool = False) -> Workload:
    """single 64 x 2048 = 131072-point sweep -> 600 x 600 @ 0.05 m."""
    rng = np.random.default_rng(seed)
    T = pose_matrix(**C2_POSE)
    flt = RejectFilter.reference() if reference_filter else None
    return Workload("C2", 600, 0.05, [lidar_sweep(rng, T)], [_frame_for(T, SensorModel.velodyne(), flt)])


def config_c3(seed: int = 3, structured_light: bool = False) -> Workload:
    """640 x 480 depth camera, 0.6 m high, pitched 35 deg down -> 400 x 400 @ 0.025 m, map centred 3 m ahead."""
    rng = np.random.default_rng(seed)
    W, H, fx, fy, cx, cy = 640, 480, 380.0, 380.0, 320.0, 240.0
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    d_opt = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u, float)], -1).reshape(-1, 3)   # optical frame: z forward
    pitch = np.deg2rad(35.0)
    # optical (x right, y down, z fwd) -> camera body (x fwd, y left, z up), then pitch down
    R_body_opt = np.array([[0, 0, 1.0], [-1, 0, 0], [0, -1, 0]])
    T = np.eye(4); T[:3, :3] = rot_zyx(0.0, pitch, 0.0) @ R_body_opt; T[:3, 3] = [0, 0, 0.6]
    dm = d_opt @ T[:3, :3].T
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.where(dm[:, 2] < -1e-6, -0.6 / dm[:, 2], np.nan)
    hit = T[:3, 3][None, :] + s[:, None] * dm
    bump = 0.03 * np.sin(2 * np.pi * hit[:, 0] / 0.5) * np.sin(2 * np.pi * hit[:, 1] / 0.5)
    hit[:, 2] = bump + rng.normal(0, 0.003, hit.shape[0])
    pts = (hit - T[:3, 3][None, :]) @ T[:3, :3]                    # back to the optical (sensor) frame
    ok = np.isfinite(pts).all(1) & (s < 12.0)
    pix = (v.reshape(-1) * W + u.reshape(-1))[ok]
    pts = pts[ok]
    xyzi = np.concatenate([pts, rng.uniform(1, 255, (pts.shape[0], 1))], 1).astype(np.float32)
    model = SensorModel.realsense_d435() if structured_light else SensorModel.velodyne()
    model.ignore_points_above, model.ignore_points_below = float("inf"), float("-inf")
    model.original_width = W
    wl = Workload("C3", 400, 0.025, [xyzi], [_frame_for(T, model)], map_position=np.array([3.0, 0.0, 0.0], np.float32))
    wl.orig_index = pix.astype(np.int32)
    return wl


def config_c4(n_sweeps: int = 32, seed0: int = 100) -> Workload:
    """32 consecutive sweeps, sensor advancing 0.05 m / sweep along +x, yaw drifting 0.1 deg / sweep,
    variance increment 1e-6 * (1 + k mod 3) before sweep k; map centre fixed for the batch."""
    clouds, frames, upd = [], [], []
    for k in range(n_sweeps):
        rng = np.random.default_rng(seed0 + k)
        pose = dict(C2_POSE); pose["x"] += 0.05 * k; pose["yaw"] += np.deg2rad(0.1 * k)
        T = pose_matrix(**pose)
        clouds.append(lidar_sweep(rng, T))
        frames.append(_frame_for(T, SensorModel.velodyne()))
        upd.append(1e-6 * (1 + k % 3))
    return Workload("C4", 600, 0.05, clouds, frames, var_updates=upd)


def c5_offsets(n_points: int = 10_000_000) -> np.ndarray:
    """offsets[n_sweeps + 1] of config_c5's concatenated cloud (every sweep 64 x 2048 points, the last one cut)."""
    per = 64 * 2048
    n_sweeps = (n_points + per - 1) // per
    return np.minimum(np.arange(n_sweeps + 1, dtype=np.int64) * per, n_points)


def config_c5(n_points: int = 10_000_000, seed: int = 5, length: int = 2400, sweeps=None) -> Workload:
    """aggregated cloud: 76 C2-style sweeps from a Lissajous path over 100 m x 100 m, concatenated in
    sweep order -> 2400 x 2400 @ 0.05 m.  Each sweep keeps its own transform (the aggregation is a batch).
    sweeps: generate only these sweeps' clouds (the others are None; every sweep has its own seed) -- a rank of the multi-GPU
    bench needs its share only; the frames are always complete."""
    per = 64 * 2048
    n_sweeps = (n_points + per - 1) // per
    want = None if sweeps is None else set(int(k) for k in sweeps)
    clouds, frames = [], []
    for k in range(n_sweeps):
        ph = 2 * np.pi * k / n_sweeps
        x, y = 50.0 * np.sin(3 * ph), 50.0 * np.sin(2 * ph + 0.5)
        yaw = np.arctan2(2 * np.cos(2 * ph + 0.5), 3 * np.cos(3 * ph))
        T = pose_matrix(x, y, 1.73, yaw, np.deg2rad(1.0), np.deg2rad(-0.5))
        frames.append(_frame_for(T, SensorModel.velodyne()))
        if want is not None and k not in want:
            clouds.append(None)
            continue
        rng = np.random.default_rng(seed * 1000 + k)
        c = lidar_sweep(rng, T)
        if k == n_sweeps - 1:
            c = c[: n_points - per * (n_sweeps - 1)]
        clouds.append(c)
    return Workload("C5", length, 0.05, clouds, frames)


def random_cloud(seed: int, n: int, extent: float, z_sigma: float = 0.3, dup_fraction: float = 0.3) -> np.ndarray:
    """Adversarial test cloud: uniform x,y with a fraction of exact duplicates / same-cell clusters."""
    rng = np.random.default_rng(seed)
    xy = rng.uniform(-extent, extent, (n, 2))
    k = int(n * dup_fraction)
    if k:
        src = rng.integers(0, n, k); dst = rng.integers(0, n, k)
        xy[dst] = xy[src] + rng.normal(0, 0.004, (k, 2))
    z = rng.normal(0, z_sigma, n)
    return np.concatenate([xy, z[:, None], rng.uniform(0, 3, (n, 1)).round()], 1).astype(np.float32)
