#!/usr/bin/env python3
"""Golden vectors produced by the REFERENCE'S OWN CODE: /root/reference/.../gpu_process.cu compiled for the CPU
(oracle/ref_build/build_ref.py -> oracle/_ref/libgem_ref.so; kernels run sequentially over their grids, CUDA runtime and
Eigen are stand-ins).  Run in the build container (needs /root/reference):

    python tests/golden/make_ref_golden.py        # rewrites tests/golden/ref_scene.npz

ref_scene.npz is ONE scripted session of the reference's interface on an 80 x 80 @ 0.1 m map, inputs and outputs:
  Init_GPU_elevationmap -> Move -> [Mapvar_update, Process_points, Fuse] x 3 frames (a Move between them) -> Map_feature
  -> Raytracing.  Stored: the clouds and frame constants; per frame the five Process_points outputs and the elevation /
  variance / lowest layers after the Fuse; the pose after every Move; rough / slope / traver of Map_feature; elevation and
  lowest after Raytracing.
tests/test_golden.py replays the session on the oracle (CPU) and on libgem_hip (GPU) against these arrays, so the pin
travels with the repository (the GPU box has no /root/reference).
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))

import ref  # noqa: E402
from gem_amd import RejectFilter, SensorModel, synth  # noqa: E402

F32 = np.float32
L, RES = 80, 0.1
POSES = [(0.3, -0.2, 0.5, 0.0), (0.9, 0.4, 0.55, 0.8), (0.1, 1.3, 0.6, -0.6)]        # x, y, z, yaw of the sensor per frame


def scene_frame(k):
    x, y, z, yaw = POSES[k]
    f = synth._frame_for(synth.pose_matrix(x, y, z, yaw, 0.01, -0.015), SensorModel.velodyne())
    f.filter = RejectFilter.reference()                     # the reference's hard-coded filter (gpu_process.cu:393)
    f.lower, f.upper = -3.0, 3.0
    return f


def scene_cloud(k):
    c = synth.random_cloud(900 + k, 6_000, 4.2, z_sigma=0.25)
    c[:, 3] = 0
    return c


def main():
    if ref.lib() is None:
        raise SystemExit("needs /root/reference (or a prebuilt oracle/_ref/libgem_ref.so)")
    r = ref.RefMap(L, RES)
    out = {"length": np.int32(L), "resolution": F32(RES)}
    for k in range(3):
        pos = np.array(POSES[k][:3], F32)
        c, s, a = r.move(pos)
        out[f"pos_{k}"], out[f"center_{k}"], out[f"start_{k}"], out[f"shift_{k}"] = pos, c, s, a
        f, cloud = scene_frame(k), scene_cloud(k)
        u = F32(1e-5 * (k + 1))
        r.mapvar_update(float(u))
        pp = r.process_points(f, cloud[:, 0], cloud[:, 1], cloud[:, 2])
        r.fuse(pp["index"], pp["height"], pp["var"])
        out[f"cloud_xyz_{k}"], out[f"var_update_{k}"] = np.ascontiguousarray(cloud[:, :3]), u
        for key in ("index", "var", "x_ts", "y_ts", "height"):
            out[f"pp_{key}_{k}"] = pp[key]
        for name in ("elevation", "variance", "lowest"):
            out[f"{name}_{k}"] = r.layer(name)
    feat = r.map_feature()
    live = r.layer("elevation") != -10
    for key in ("rough", "slope", "traver"):
        out[f"feature_{key}"] = np.where(live, feat[key], 0).astype(F32)          # empty cells: uninitialised in the reference
    out["feature_live"] = live
    out["traver_layer"] = r.layer("traver")
    r.raytracing()
    out["elevation_after_raytracing"] = r.layer("elevation")
    out["lowest_after_raytracing"] = r.layer("lowest")
    out["deleted"] = np.int32(int(live.sum()) - int((out["elevation_after_raytracing"] != -10).sum()))
    np.savez_compressed(HERE / "ref_scene.npz", **out)
    print("ref_scene.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(out.items())[:6]}, "... deleted by raytracing:", int(out["deleted"]),
          "accepted per frame:", [int((out[f"pp_index_{k}"] >= 0).sum()) for k in range(3)])


if __name__ == "__main__":
    main()
