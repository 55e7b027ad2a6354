#!/usr/bin/env python3
"""Throughput of every BASELINE.json configuration on one MI355X (inputs resident in HBM).

    python tools/bench_configs.py [--reps R] [--configs c2,c3,c4,c5]

Per config: wall time per call (stream-synchronised loop), per-kernel dispatch times (gem_set_timing),
points/s and algorithmic GB/s  B_alg = 16 N + 16 C_touched (+ 8 L^2 per dense variance pass), SURVEY 8d.
bench.py stays the contract line (C2); this tool gives the other rows of DESIGN.md section 6.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from gem_amd import ElevationMap, synth  # noqa: E402


LAST_SORT = None


def timed(emap, fn, reps, warm=3):
    for _ in range(warm):
        fn()
    emap.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    emap.synchronize()
    wall = (time.perf_counter() - t0) / reps
    emap.set_timing(True); emap.stats(reset=True)
    for _ in range(reps):
        fn()
    st = emap.stats(); emap.set_timing(False)
    global LAST_SORT
    LAST_SORT = None
    if st["launches_walk"]:
        LAST_SORT = {"us_sort": [round(1e3 * v / st["launches_sort"], 2) for v in st["ms_sort"]], "us_walk": round(1e3 * st["ms_walk"] / st["launches_walk"], 2),
                     "kernels": ["count1", "scan1", "scatter1", "count2", "scan2", "scatter2"]}
        return wall, 1e3 * sum(st["ms_sort"]) / reps, 1e3 * st["ms_walk"] / reps
    return wall, 1e3 * st["ms_bin"] / reps, 1e3 * (st["ms_fuse"] + st["ms_frame"]) / reps


def touched(emap, fn):
    emap.set_counting(True)
    fn()
    c = emap.stats()["cells_touched"]
    emap.set_counting(False)
    return c


def report(name, n_pts, cells, dense_passes, L, wall, us_bin, us_fuse):
    alg = 16.0 * n_pts + 16.0 * cells + 8.0 * L * L * dense_passes
    out = {"config": name, "points": n_pts, "cells_touched": int(cells), "dense_passes": dense_passes,
           "wall_us": wall * 1e6, "us_bin": us_bin, "us_fuse": us_fuse,
           "points_per_s": n_pts / wall, "alg_MB": alg / 1e6, "alg_GBps_wall": alg / wall / 1e9,
           "frac_of_8TBps": alg / wall / 8e12}
    if LAST_SORT:
        out["sorted_pipeline"] = LAST_SORT
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--configs", default="c2,c3,c4,c5")
    ap.add_argument("--c5-points", type=int, default=10_000_000)
    ap.add_argument("--debug", default="", help="gem_debug_set knobs applied to every map, e.g. overlap=0,dense_min=300")
    args = ap.parse_args()
    if args.debug:
        ElevationMap.default_debug = {k: int(v) for k, v in (kv.split("=") for kv in args.debug.split(","))}
    want = args.configs.split(",")
    dev = torch.device("cuda", 0)

    if "c2" in want:
        wl = synth.config_c4(n_sweeps=8)
        d = [torch.from_numpy(c).to(dev) for c in wl.clouds]
        m = ElevationMap(wl.length, wl.resolution)
        k = [0]
        def f():
            m.add(wl.frames[k[0] % 8], d[k[0] % 8]); k[0] += 1
        for _ in range(16): f()
        wall, ub, uf = timed(m, f, args.reps * 4)
        report("C2 single sweep", d[0].shape[0], touched(m, f), 0, wl.length, wall, ub, uf)
        # the traversability stage that follows every frame (Map_feature): kernel only, layers stay resident
        for _ in range(5): m.map_feature(fetch=False)
        m.synchronize(); t0 = time.perf_counter()
        for _ in range(args.reps * 4): m.map_feature(fetch=False)
        m.synchronize(); dt = (time.perf_counter() - t0) / (args.reps * 4)
        cells = wl.length * wl.length
        print(json.dumps({"config": "Map_feature on the fused C2 map (600x600), device-resident", "wall_us": dt * 1e6,
                          "cells_per_s": cells / dt, "alg_MB": cells * 16 / 1e6, "alg_GBps_wall": cells * 16 / dt / 1e9,
                          "note": "B_alg = 4 B elevation read + 12 B rough/slope/traver written per cell"}), flush=True)
        # the per-frame sequence of the unmodified node (adapter): fuse with lowest tracking, Map_feature, Raytracing
        m.set_lowest_tracking(True)
        def g():
            f(); m.map_feature(fetch=False); m.raytracing()
        for _ in range(16): g()
        m.synchronize(); t0 = time.perf_counter()
        for _ in range(args.reps * 2): g()
        m.synchronize(); dt = (time.perf_counter() - t0) / (args.reps * 2)
        for _ in range(8): f()
        m.synchronize(); t0 = time.perf_counter()
        for _ in range(args.reps * 2): f()
        m.synchronize(); dtf = (time.perf_counter() - t0) / (args.reps * 2)
        print(json.dumps({"config": "C2 node sequence: add (lowest tracking on) + Map_feature + Raytracing, device-resident",
                          "wall_us": dt * 1e6, "add_with_lowest_tracking_us": dtf * 1e6,
                          "note": "tracking on: k_frame<LOWEST> (one launch per frame, the fusion deferred like the plain stream)"}), flush=True)
        m.close()

    if "color" in want:
        # the step in front of the path: input colourisation of a C2-sized cloud from a 1280 x 720 camera image (gem_colorize_device)
        rng = np.random.default_rng(3)
        n, w, h = 131072, 1280, 720
        pts = np.empty((n, 4), np.float32)
        pts[:, 0] = rng.uniform(0.5, 30.0, n); pts[:, 1] = rng.normal(0, 6.0, n); pts[:, 2] = rng.normal(0, 3.0, n); pts[:, 3] = 1.0
        tl = np.array([[0, -1, 0, 0.02], [0, 0, -1, -0.05], [1, 0, 0, 0.1], [0, 0, 0, 1]], np.float64)
        tc = np.array([[0.8 * w, 0, 0.5 * w, 0], [0, 0.8 * w, 0.5 * h, 0], [0, 0, 1, 0]], np.float64)
        P = ElevationMap.lidar_to_image(tc, tl)
        img = torch.from_numpy(rng.integers(1, 256, (h, w, 3)).astype(np.uint8)).to(dev)
        d0 = torch.from_numpy(pts).to(dev)
        m = ElevationMap(40, 0.1)
        d = d0.clone()
        for _ in range(5): m.colorize(P, img, d)
        m.synchronize(); t0 = time.perf_counter()
        for _ in range(args.reps * 2): m.colorize(P, img, d)
        m.synchronize(); dt = (time.perf_counter() - t0) / (args.reps * 2)
        rgb, _ = m.colorize(P, img, d); m.synchronize()
        print(json.dumps({"config": "colourisation of 131072 points from a 1280x720 image, device-resident", "wall_us": dt * 1e6,
                          "points_per_s": n / dt, "coloured": int((rgb != 0).sum().item()),
                          "note": "memset of the pixel table (3.7 MB) + k_sort_project<camera> + 2 x (scan, scatter) + count + 3 colour kernels"}), flush=True)
        m.close()

    if "reserve" in want:
        # the first big frame after small ones: the arenas grow (everything in flight is waited for, hipFree + hipMalloc) unless
        # gem_reserve sized them beforehand
        wl = synth.config_c4(n_sweeps=2)
        big = torch.from_numpy(wl.clouds[1]).to(dev)
        small = torch.from_numpy(wl.clouds[0][:8192].copy()).to(dev)
        out = {}
        for label, reserve in (("first_big_frame_us_without_reserve", False), ("first_big_frame_us_after_gem_reserve", True)):
            m = ElevationMap(wl.length, wl.resolution)
            if reserve:
                m.reserve(big.shape[0], 1)
            for _ in range(8):
                m.add(wl.frames[0], small)
            m.synchronize(); t0 = time.perf_counter()
            m.add(wl.frames[1], big); m.synchronize()
            out[label] = (time.perf_counter() - t0) * 1e6
            t0 = time.perf_counter()
            m.add(wl.frames[1], big); m.synchronize()
            out[label.replace("first", "second")] = (time.perf_counter() - t0) * 1e6
            out["arena_allocations" + ("_after_reserve" if reserve else "")] = m.debug_get("arena_allocations")
            m.close()
        print(json.dumps({"config": "gem_reserve: 8 x 8192-pt frames, then a 131072-pt frame (each timed with a synchronisation)", **out}), flush=True)

    if "c3" in want:
        wl = synth.config_c3()
        d = torch.from_numpy(wl.clouds[0]).to(dev)
        m = ElevationMap(wl.length, wl.resolution)
        m.move(wl.map_position)
        def f():
            m.add(wl.frames[0], d)
        wall, ub, uf = timed(m, f, args.reps)
        report("C3 depth 640x480", d.shape[0], touched(m, f), 0, wl.length, wall, ub, uf)
        m.close()

    if "c4" in want:
        wl = synth.config_c4(n_sweeps=32)
        cat = torch.from_numpy(np.concatenate(wl.clouds)).to(dev)
        off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
        m = ElevationMap(wl.length, wl.resolution)
        # consecutive calls carry DIFFERENT frames (four pose sets in turn, bench.perturbed_frames): every call builds and uploads
        # its device tables, as a mapping loop's calls do (a replayed batch finds them cached)
        import bench
        pbs = [m.pack_batch(bench.perturbed_frames(wl.frames, j), off, wl.var_updates) for j in range(4)]
        turn = [0]
        def f():
            m.add_batch(pbs[turn[0] & 3], cat); turn[0] += 1
        wall, ub, uf = timed(m, f, max(args.reps, 20), warm=6)
        report("C4 batch of 32 sweeps + var updates", cat.shape[0], touched(m, f), 32, wl.length, wall, ub, uf)
        m.close()

    if "c5" in want:
        wl = synth.config_c5(n_points=args.c5_points)
        cat = torch.from_numpy(np.concatenate(wl.clouds)).to(dev)
        off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
        m = ElevationMap(wl.length, wl.resolution)
        import bench
        pbs = [m.pack_batch(bench.perturbed_frames(wl.frames, j), off, None) for j in range(4)]     # (see C4)
        turn = [0]
        def f():
            m.add_batch(pbs[turn[0] & 3], cat); turn[0] += 1
        wall, ub, uf = timed(m, f, max(args.reps // 2, 20), warm=6)
        report(f"C5 aggregated {cat.shape[0]} pts -> {wl.length}^2 (one GPU)", cat.shape[0], touched(m, f), 0, wl.length, wall, ub, uf)
        m.close()


if __name__ == "__main__":
    main()
