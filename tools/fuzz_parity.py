#!/usr/bin/env python3
"""Randomised soak test on the GPU box: random map sizes, poses, clouds (LiDAR sweeps, uniform clouds with same-cell clusters, depth-image
like grids), single adds and batches with and without variance increments, moves, lowest tracking + ray tracing, random pipeline knobs --
every step compared bit for bit with the CPU oracle (test infrastructure: this tool is a test, not the product).

    python tools/fuzz_parity.py [--seconds 120] [--seed 1]
Prints one line per scenario and a final summary; exit code 1 on the first mismatch (the scenario's seed is printed).
"""
import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import torch  # noqa: E402
import oracle  # noqa: E402
from gem_amd import ElevationMap, RejectFilter, SensorModel, synth  # noqa: E402

F32 = np.float32
KNOBS = [{}, {}, {"sort_min_points": 1}, {"sort_min_points": 1, "sort_form": 1}, {"sort_min_points": 1, "sort_form": 2},
         {"sort_min_points": 1, "sort_form": 2, "sort_passes": 2}, {"sort_min_points": 1, "sort_form": 2, "sort_passes": 2, "blk_batch": 2048},
         {"sort_min_points": 1, "sort_form": 2, "blk_batch": 2048}, {"sort_min_points": 1, "sort_form": 1, "sort_passes": 3},
         {"sort_min_points": 1, "fast_laser": 0}, {"dense_min": 0}, {"sort_min_points": 1, "sort_form": 2, "lane_sort": 0},
         {"sort_min_points": 1, "sort_form": 1, "sort_chunk": 4096}, {"sort_min_points": 1, "sort_form": 2, "sort_chunk": 4096},
         {"sort_min_points": 1, "sort_form": 2, "sort_passes": 2, "sort_chunk": 1024}, {"sort_min_points": 1, "sort_form": 1, "fuse_count": 0},
         {"sort_min_points": 1, "sort_form": 1, "sort_passes": 3, "fuse_count": 2, "sort_chunk": 4096}, {"sort_min_points": 1, "sort_form": 2, "sort_passes": 2, "fuse_count": 0}]


def make_cloud(rng, kind, n, extent, T):
    if kind == 0:                                             # LiDAR sweep, truncated / repeated to n points
        c = synth.lidar_sweep(rng, T, beams=int(rng.integers(8, 65)), azimuth_steps=int(rng.integers(256, 2049)), max_range=float(rng.uniform(5, 80)))
        reps = -(-n // c.shape[0])
        return np.tile(c, (reps, 1))[:n].copy()
    if kind == 1:
        return synth.random_cloud(int(rng.integers(1 << 30)), n, extent, z_sigma=float(rng.uniform(0.01, 0.5)), dup_fraction=float(rng.uniform(0, 0.9)))
    # depth-image like: a grid seen from above, image row by image row, many points per cell
    w = int(rng.integers(64, 640)); h = max(n // w, 1)
    u, v = np.meshgrid(np.linspace(-1, 1, w), np.linspace(-1, 1, h))
    s = float(rng.uniform(0.05, 0.4)) * extent
    pts = np.stack([u.ravel() * s, v.ravel() * s, rng.normal(0, 0.02, u.size) - 1.0, np.ones(u.size)], 1).astype(F32)
    return pts[:n].copy() if pts.shape[0] >= n else pts


def compare(gpu, ora, what, layers=("elevation", "variance")):
    for name in layers:
        g, o = gpu.layer(name), ora.layer(name)
        if not np.array_equal(g, o):
            bad = np.flatnonzero(g.ravel() != o.ravel())
            raise AssertionError(f"{what}: {name}: {bad.size} cells differ, first {bad[:5]}, gpu {g.ravel()[bad[:5]]} oracle {o.ravel()[bad[:5]]}")


def scenario(seed):
    rng = np.random.default_rng(seed)
    L = int(rng.choice([33, 64, 75, 100, 128, 200, 251, 300, 400, 600, 700, 1000])) if rng.random() < 0.9 else int(rng.integers(20, 900))
    res = float(rng.choice([0.025, 0.05, 0.1, 0.2]))
    knobs = dict(KNOBS[int(rng.integers(len(KNOBS)))])
    lowest = bool(rng.random() < 0.25)
    # calls in a row with nothing read in between (what a node does): deferred halves -- the tile fusion of the newest sweep, the walk
    # of an overlapped sorted pass -- are then launched by LATER calls; with the streams overlapped for passes of any size
    eager = bool(rng.random() < 0.5)
    if eager and rng.random() < 0.7:
        knobs["overlap_min_points"] = 1
    gpu = ElevationMap(L, res, debug=knobs); ora = oracle.OracleMap(L, res)
    if lowest:
        gpu.set_lowest_tracking(True)
    ct = int(rng.choice([0, 1, 4, 4, 8]))                      # host arrays: the runtime's path, or the pinned staging + copy threads
    try:
        gpu.debug_set("copy_threads", ct)
    except Exception:                                          # (a library from before the knob, when bisecting)
        ct = -1
    if os.environ.get("FUZZ_COPY_THREADS") and ct >= 0:                    # (reproducing a scenario with another route for its host arrays)
        gpu.debug_set("copy_threads", int(os.environ["FUZZ_COPY_THREADS"]))
    extent = 0.5 * L * res
    pts_total = 0
    steps = int(rng.integers(2, 6))
    attr_seen = ()                                             # attribute layers some step so far has written: compared from then on
    for step in range(steps):
        if rng.random() < 0.5:
            pos = [float(rng.uniform(-1, 1)) * extent * 0.3, float(rng.uniform(-1, 1)) * extent * 0.3, float(rng.uniform(0.3, 1.5))]
            gpu.move(pos); ora.move(pos)
        cx, cy = ora.pose()[0]
        n_sweeps = int(rng.choice([1, 1, 2, 3, 5, 9]))
        frames, clouds = [], []
        for s in range(n_sweeps):
            T = synth.pose_matrix(float(cx) + float(rng.normal(0, 0.2)), float(cy) + float(rng.normal(0, 0.2)), float(rng.uniform(0.3, 1.5)),
                                  float(rng.uniform(-3, 3)), float(rng.normal(0, 0.03)), float(rng.normal(0, 0.03)))
            f = synth._frame_for(T, SensorModel.velodyne(), RejectFilter.reference() if rng.random() < 0.2 else None)
            if rng.random() < 0.3:
                f.lower, f.upper = -3.0, 3.0
            n = int(rng.choice([1, 63, 64, 1000, 4097, 30000, 131072, 131073, 150000, 199999, 200000, 300000]))
            n = min(n, 2_000_000 // n_sweeps)
            c = make_cloud(rng, int(rng.integers(3)), n, extent, T)
            if c.shape[0] == 0:
                c = np.zeros((1, 4), F32)
            frames.append(f); clouds.append(c)
        pts_total += sum(c.shape[0] for c in clouds)
        incs = [float(rng.uniform(0, 1e-4)) for _ in range(n_sweeps)] if rng.random() < 0.6 else None
        entry = "add_batch"; attr_layers = ()
        if n_sweeps == 1 and rng.random() < 0.5:
            if incs:
                gpu.mapvar_update(incs[0]); ora.mapvar_update(incs[0])
            how = rng.random()
            c0 = clouds[0]; n0 = c0.shape[0]
            coloured = bool(rng.random() < 0.35)                 # colours / intensity ride along: the attribute layers are compared too
            if coloured:
                attr_layers = ("intensity", "color_r", "color_g", "color_b")
            oi = np.ascontiguousarray(rng.permutation(n0).astype(np.int32)) if rng.random() < 0.3 else None     # (the reject filter reads it)
            entry = ("process_points + fuse (host arrays)" if how < 0.25 else ("add (device cloud)" if how < 0.65 else "add (host cloud)")) + \
                    (", colours" if coloured else "") + (", orig_index" if oi is not None else "")
            if how < 0.25:                                     # the node's two calls with host arrays (Process_points, then Fuse)
                g = gpu.process_points(frames[0], c0[:, 0], c0[:, 1], c0[:, 2], orig_index=oi); o = ora.process_points(frames[0], c0[:, 0], c0[:, 1], c0[:, 2], orig_index=oi)
                for k in ("index", "var", "x_ts", "y_ts", "height"):
                    if not np.array_equal(g[k], o[k]):
                        raise AssertionError(f"seed {seed} step {step}: process_points {k}")
                if coloured:
                    R, G, B = (rng.integers(0, 3, n0).astype(np.int32) * 100 for _ in range(3)); I = rng.integers(0, 3, n0).astype(F32)
                    gpu.fuse(g["index"], g["height"], g["var"], R, G, B, I); ora.fuse(o["index"], o["height"], o["var"], R, G, B, I)
                else:
                    gpu.fuse(g["index"], g["height"], g["var"]); ora.fuse(o["index"], o["height"], o["var"])
            else:
                rgb = None
                if coloured:
                    rgb = ((rng.integers(0, 3, n0).astype(np.uint32) * 100) << 16) | ((rng.integers(0, 3, n0).astype(np.uint32) * 100) << 8) | (rng.integers(0, 3, n0).astype(np.uint32) * 100)
                    c0 = c0.copy(); c0[:, 3] = rng.integers(0, 3, n0).astype(F32)
                if how < 0.65:
                    dv = lambda a, dt=None: None if a is None else torch.from_numpy(a if dt is None else a.view(dt)).cuda()
                    gpu.add(frames[0], dv(c0), dv(rgb, np.int32), dv(oi))
                else:
                    gpu.add(frames[0], c0, rgb, oi)
                ora.add(frames[0], c0, rgb, oi)
        else:
            off = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])])
            if rng.random() < 0.3:                             # the batch as separate HOST arrays (gem_add_batch)
                entry = "add_batch (host arrays)"
                gpu.add_batch_host(frames, clouds, incs)
            else:
                gpu.add_batch(frames, torch.from_numpy(np.concatenate(clouds, 0)).cuda(), off, incs)
            for k in range(n_sweeps):
                if incs:
                    ora.mapvar_update(incs[k])
                ora.add(frames[k], clouds[k])
        attr_seen = attr_seen or attr_layers
        if eager and step < steps - 1 and rng.random() < 0.6:
            continue                                           # (the next step's comparison sees this one's result too)
        compare(gpu, ora, f"seed {seed} step {step} after the fusion by {entry}, copy_threads {ct}, {[c.shape[0] for c in clouds]} points ({knobs}, L {L}, sweeps {n_sweeps})", ("elevation", "variance") + attr_seen + (("lowest",) if lowest else ()))
        if lowest and rng.random() < 0.7:
            gpu.map_feature(fetch=False); ora.map_feature()
            gpu.debug_set("ray_lanes", int(rng.choice([1, 4, 8, 16]))); gpu.debug_set("ray_depth", int(rng.choice([4, 8])))
            gpu.raytracing(); ora.raytracing()
            compare(gpu, ora, f"seed {seed} step {step} after ray tracing ({knobs}, L {L})", ("elevation", "variance", "lowest"))
    gpu.close()
    return L, knobs, lowest, steps, pts_total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    t0 = time.time(); n = 0; pts = 0
    seed = a.seed * 100000
    while time.time() - t0 < a.seconds:
        try:
            L, knobs, lowest, steps, p = scenario(seed)
        except AssertionError as e:
            print(f"MISMATCH in scenario seed {seed}: {e}", flush=True)
            return 1
        n += 1; pts += p
        print(f"ok seed {seed} L {L} lowest {int(lowest)} steps {steps} points {p} knobs {knobs}", flush=True)
        seed += 1
    print(f"SUMMARY: {n} scenarios, {pts} points, {time.time() - t0:.0f} s, no mismatch", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
