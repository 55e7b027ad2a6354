#!/usr/bin/env python3
"""Regenerates the golden fixtures from the CPU oracle (oracle/gem_oracle.c).

The reference ships no tests, golden vectors or fixtures and cannot be built or imported here
(CUDA + Eigen + ROS), so these vectors are produced by our own restatement of its semantics; the restatement is pinned against the
reference's own gpu_process.cu compiled for the CPU (tests/test_reference_compiled.py), and make_ref_golden.py records
vectors from that compiled reference directly (ref_scene.npz).

    python tests/golden/make_golden.py        # rewrites c1.npz, chain.npz, digests.json

  c1.npz       BASELINE config 1 (10 k planar points -> 200 x 200 @ 0.1 m): inputs AND outputs
  chain.npz    a 600-point single-cell chain + collisions case: inputs and outputs
  feature.npz  a 48 x 48 terrain (slopes, a step, holes) after a Move: elevation in, rough / slope / traver out
               (the traversability stage, gemo_map_feature)
  digests.json SHA-256 of the oracle's layers for configs C2 / C3 / C4(4 sweeps), for the FULL-SIZE configurations that
               bench.py times (C4: 32 sweeps batched; C2 stream of 16 sweeps; C5: 10^7 points -> 2400^2, whole map and
               the eight row strips) and of the input clouds (detects drift of the seeded generators)
"""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))

import oracle  # noqa: E402
from gem_amd import synth  # noqa: E402
from gem_amd.api import SensorModel  # noqa: E402


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def frame_arrays(f):
    return dict(T=np.asarray(f.T, np.float32), lower=np.float64(f.lower), upper=np.float64(f.upper),
                model=np.int32(f.model.kind), params=np.asarray(list(f.model.params), np.float64),
                sensor_jacobian=np.asarray(f.sensor_jacobian, np.float32), C_SB_T=np.asarray(f.C_SB_T, np.float32),
                P_mul_C_BM_T=np.asarray(f.P_mul_C_BM_T, np.float32), B_r_BS_skew=np.asarray(f.B_r_BS_skew, np.float32))


def main():
    digests = {}
    # ---- C1 with inputs -------------------------------------------------------------------------
    wl = synth.config_c1()
    m = oracle.OracleMap(wl.length, wl.resolution)
    c = wl.clouds[0]
    pp = m.process_points(wl.frames[0], c[:, 0], c[:, 1], c[:, 2])
    m.add(wl.frames[0], c)
    e1, v1 = m.layer("elevation"), m.layer("variance")
    m.mapvar_update(2e-5); m.add(wl.frames[0], c[::-1].copy())
    np.savez_compressed(HERE / "c1.npz", cloud=c, index=pp["index"], var=pp["var"], height=pp["height"],
                        elevation_1=e1, variance_1=v1, elevation_2=m.layer("elevation"), variance_2=m.layer("variance"),
                        **{"frame_" + k: v for k, v in frame_arrays(wl.frames[0]).items()})
    digests["c1_cloud"] = sha(c)

    # ---- chain / collision case ------------------------------------------------------------------
    rng = np.random.default_rng(42)
    n = 600
    pts = np.zeros((n, 4), np.float32)
    pts[:, 0] = 0.33 + rng.uniform(0, 0.04, n); pts[:, 1] = -0.21 + rng.uniform(0, 0.04, n)
    pts[:300, 2] = rng.normal(0.2, 0.01, 300); pts[300:, 2] = rng.normal(0.2, 0.3, 300)      # inliers then outliers
    pts[::50, 0] += 0.1                                                                       # a few neighbours
    f = synth._frame_for(np.eye(4), SensorModel.velodyne())
    m2 = oracle.OracleMap(32, 0.1)
    m2.add(f, pts)
    np.savez_compressed(HERE / "chain.npz", cloud=pts, elevation=m2.layer("elevation"), variance=m2.layer("variance"))

    # ---- traversability stage on a rough terrain ----------------------------------------------------
    L, res = 48, 0.1
    rng = np.random.default_rng(7)
    x, y = np.meshgrid(np.arange(L) * res, np.arange(L) * res, indexing="ij")
    z = 0.7 * x - 0.2 * y + 0.3 * np.sin(2 * np.pi * x / 0.9) * np.cos(2 * np.pi * y / 0.7) + rng.normal(0, 0.01, (L, L))
    z[rng.random((L, L)) < 0.12] = -10.0
    z[20:21, :] += 0.4
    z = z.astype(np.float32)
    m3 = oracle.OracleMap(L, res)
    m3.move(np.array([0.73, -0.41, 0.0], np.float32))
    m3.set_layer("elevation", z)
    ft = m3.map_feature()
    np.savez_compressed(HERE / "feature.npz", elevation=z, position=np.array([0.73, -0.41, 0.0], np.float32),
                        rough=ft["rough"], slope=ft["slope"], traver=ft["traver"])

    # ---- digests of the big configurations --------------------------------------------------------
    for name, mk in (("c2", synth.config_c2), ("c2_filter", lambda: synth.config_c2(reference_filter=True)), ("c3", synth.config_c3)):
        wl = mk()
        mm = oracle.OracleMap(wl.length, wl.resolution)
        if wl.map_position is not None:
            mm.move(wl.map_position)
        mm.add(wl.frames[0], wl.clouds[0])
        digests[name] = {"cloud": sha(wl.clouds[0]), "elevation": sha(mm.layer("elevation")), "variance": sha(mm.layer("variance")),
                         "accepted": mm.last_counts[0], "cells_touched": mm.last_counts[1]}
    wl = synth.config_c4(n_sweeps=4)
    mm = oracle.OracleMap(wl.length, wl.resolution)
    for k in range(4):
        mm.mapvar_update(wl.var_updates[k]); mm.add(wl.frames[k], wl.clouds[k])
    digests["c4_4sweeps"] = {"cloud": sha(np.concatenate(wl.clouds)), "elevation": sha(mm.layer("elevation")), "variance": sha(mm.layer("variance"))}
    # ---- the configurations bench.py TIMES, at full size (BASELINE configs[3] and [4]) ------------------------------
    wl = synth.config_c4(n_sweeps=32)                       # one batched call: 32 sweeps, a variance increment before each
    mm = oracle.OracleMap(wl.length, wl.resolution)
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    mm.add_batch_mt(wl.frames, np.concatenate(wl.clouds), off, wl.var_updates)
    digests["c4_32"] = {"cloud": sha(np.concatenate(wl.clouds)), "elevation": sha(mm.layer("elevation")), "variance": sha(mm.layer("variance"))}
    mm.add_batch_mt(wl.frames, np.concatenate(wl.clouds), off, wl.var_updates)          # the same batch again, into the populated map
    digests["c4_32_twice"] = {"elevation": sha(mm.layer("elevation")), "variance": sha(mm.layer("variance"))}
    wl = synth.config_c4(n_sweeps=8, seed0=100)             # the stream bench.py cycles through: 8 sweeps, two rounds, no increments
    mm = oracle.OracleMap(wl.length, wl.resolution)
    for k in range(16):
        mm.add(wl.frames[k % 8], wl.clouds[k % 8])
    digests["c2_stream16"] = {"cloud": sha(np.concatenate(wl.clouds)), "elevation": sha(mm.layer("elevation")), "variance": sha(mm.layer("variance"))}
    wl = synth.config_c5()                                  # 10^7 points -> 2400 x 2400, 77 sweeps with their own frames, no increments
    mm = oracle.OracleMap(wl.length, wl.resolution)
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    mm.add_batch_mt(wl.frames, np.concatenate(wl.clouds), off, None)
    digests["c5_full"] = {"cloud": sha(np.concatenate(wl.clouds)), "elevation": sha(mm.layer("elevation")), "variance": sha(mm.layer("variance")),
                          "strips": [{"elevation": sha(mm.layer("elevation")[r * 300:(r + 1) * 300]), "variance": sha(mm.layer("variance")[r * 300:(r + 1) * 300])}
                                     for r in range(8)]}
    (HERE / "digests.json").write_text(json.dumps(digests, indent=1) + "\n")
    print(json.dumps(digests, indent=1))


if __name__ == "__main__":
    main()
