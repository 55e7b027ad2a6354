#!/usr/bin/env python3
"""Builds oracle/_ref/libgem_ref.so: the REFERENCE's own gpu_process.cu, compiled for the CPU.

TEST INFRASTRUCTURE ONLY.  The source is read where it lies (/root/reference/...), nothing of it is copied into the
repository: the only textual change is the kernel-launch syntax (`k<<<g, b>>>(args)` -> `GEMREF_LAUNCH(g, b, k(args))`,
which g++ can parse), made in a temporary file that is deleted after the compile.  CUDA runtime calls, `__global__`,
threadIdx ... come from cuda_runtime.h next to this script (kernels run sequentially over the grid), Eigen from the
stand-in under Eigen/ (Eigen is not installed here; its fixed-size product order is restated there, see its header).
Compiled with -ffp-contract=off: the comparison is with the reference's source-level arithmetic (nvcc would
contract a*b+c into FMAs, a build-flag effect that no CPU build of the reference shares either).

    python oracle/ref_build/build_ref.py [--reference /root/reference] [--force]
"""
from __future__ import annotations

import argparse
import re
import subprocess
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
OUT = HERE.parent / "_ref" / "libgem_ref.so"
REL = "elevation_mapping/elevation_mapping/cuda/gpu_process.cu"


def rewrite_launches(text: str) -> tuple[str, int]:
    """`name<<<grid, block>>>(args)` -> `GEMREF_LAUNCH(grid, block, name(args))` (arguments may span lines)."""
    out, pos, n = [], 0, 0
    for m in re.finditer(r"(\w+)\s*<<<([^<>]*?)>>>\s*\(", text):
        if m.start() < pos:
            continue
        line_start = text.rfind("\n", 0, m.start()) + 1
        if "//" in text[line_start:m.start()]:             # a commented-out launch
            continue
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        grid, block = [s.strip() for s in m.group(2).split(",")]
        out.append(text[pos:m.start()])
        out.append(f"GEMREF_LAUNCH({grid}, {block}, {m.group(1)}({text[m.end():i - 1]}))")
        pos, n = i, n + 1
    out.append(text[pos:])
    return "".join(out), n


def build(reference: Path = Path("/root/reference"), force: bool = False, verbose: bool = False) -> Path | None:
    src = reference / REL
    if not src.exists():
        return OUT if OUT.exists() else None                # e.g. on the GPU box: use the prebuilt library
    deps = [src, HERE / "cuda_runtime.h", HERE / "ref_exports.inc", HERE / "Eigen" / "Core", Path(__file__)]
    if OUT.exists() and not force and all(OUT.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return OUT
    text, n = rewrite_launches(src.read_text(errors="replace"))
    if n < 10:
        raise RuntimeError(f"only {n} kernel launches found in {src}")
    OUT.parent.mkdir(exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        tu = Path(tmp) / "gem_ref_tu.cpp"
        tu.write_text(text + f'\n#include "{HERE / "ref_exports.inc"}"\n')
        cmd = ["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=c++14", "-w", "-fPIC", "-shared",
               f"-I{HERE}", str(tu), "-o", str(OUT)]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("g++ failed on the reference translation unit:\n" + res.stderr[-4000:])
    return OUT


OUT_MOTION = HERE.parent / "_ref" / "libgem_ref_motion.so"
REL_MOTION = "elevation_mapping/elevation_mapping/src/RobotMotionMapUpdater.cpp"
REL_INCLUDE = "elevation_mapping/elevation_mapping/include"


def build_motion(reference: Path = Path("/root/reference"), force: bool = False, verbose: bool = False) -> Path | None:
    """oracle/_ref/libgem_ref_motion.so: the reference's RobotMotionMapUpdater.cpp (and its own header) compiled where they lie,
    against the stand-ins under motion/ for what is not installed here -- Eigen, kindr, ROS, and the one method of ElevationMap the
    class touches.  The Jacobians, the F matrix and the covariance products are the reference's text; nothing of it is copied."""
    src = reference / REL_MOTION
    if not src.exists():
        return OUT_MOTION if OUT_MOTION.exists() else None
    m = HERE / "motion"
    deps = [src, reference / REL_INCLUDE / "elevation_mapping" / "RobotMotionMapUpdater.hpp", Path(__file__)] + [f for f in m.rglob("*") if f.is_file()]
    if OUT_MOTION.exists() and not force and all(OUT_MOTION.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return OUT_MOTION
    OUT_MOTION.parent.mkdir(exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        tu = Path(tmp) / "gem_ref_motion_tu.cpp"
        tu.write_text(f'#include "{src}"\n#include "{m / "motion_exports.inc"}"\n')
        cmd = ["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=c++14", "-w", "-fPIC", "-shared",
               f"-I{m}", f"-I{reference / REL_INCLUDE}", str(tu), "-o", str(OUT_MOTION)]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("g++ failed on the reference's RobotMotionMapUpdater.cpp:\n" + res.stderr[-4000:])
    return OUT_MOTION


OUT_SENSORS = HERE.parent / "_ref" / "libgem_ref_sensors.so"
REL_SENSORS = "elevation_mapping/elevation_mapping/src/sensor_processors"
SENSOR_FILES = ["PerfectSensorProcessor.cpp", "StereoSensorProcessor.cpp", "StructuredLightSensorProcessor.cpp"]


def build_sensors(reference: Path = Path("/root/reference"), force: bool = False, verbose: bool = False) -> Path | None:
    """oracle/_ref/libgem_ref_sensors.so: the reference's Perfect / Stereo / StructuredLight SensorProcessor.cpp (and their own
    headers, point structs included) compiled where they lie against the stand-ins under sensors/ for what is not installed here --
    Eigen, kindr, PCL, ROS, TF.  The noise models (computeVariances: the sensor covariance of every point, the Jacobians, the error
    propagation) are the reference's text; nothing of it is copied.  (LaserSensorProcessor.cpp is left out: its CPU loop is commented
    out in the reference, the laser model lives in gpu_process.cu -- libgem_ref.so.)  `private` / `protected` are lifted for the
    reference's files so that the harness can set the transformation members TF would have filled."""
    srcs = [reference / REL_SENSORS / f for f in SENSOR_FILES]
    if not all(s.exists() for s in srcs):
        return OUT_SENSORS if OUT_SENSORS.exists() else None
    m = HERE / "sensors"
    deps = srcs + [reference / REL_SENSORS / "SensorProcessorBase.cpp", Path(__file__)] + [f for f in m.rglob("*") if f.is_file()]
    if OUT_SENSORS.exists() and not force and all(OUT_SENSORS.stat().st_mtime >= d.stat().st_mtime for d in deps):
        return OUT_SENSORS
    OUT_SENSORS.parent.mkdir(exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        tu = Path(tmp) / "gem_ref_sensors_tu.cpp"
        std = "\n".join(f"#include <{h}>" for h in ("cmath", "cstdint", "iostream", "limits", "memory", "string", "unordered_map", "vector",
                                                    "Eigen/Core", "kindr/Core", "ros/ros.h", "tf/transform_listener.h", "pcl/point_cloud.h",
                                                    "pcl/filters/filter.h", "pcl/filters/passthrough.h", "boost/shared_ptr.hpp"))
        body = "\n".join(f'#include "{s}"' for s in srcs)
        # SensorProcessorBase::readcomputerparam (SPB.cpp:270-290): the function's own text, cut out of the reference's file where it
        # lies (the rest of that file needs TF look-ups, PCL transforms and the GPU entry points) into this temporary translation
        # unit -- never into the repository
        spb = (reference / REL_SENSORS / "SensorProcessorBase.cpp").read_text()
        at = spb.index("void SensorProcessorBase::readcomputerparam(")
        depth, end = 0, None
        for i in range(spb.index("{", at), len(spb)):
            depth += spb[i] == "{"; depth -= spb[i] == "}"
            if depth == 0:
                end = i + 1; break
        rcp = "namespace elevation_mapping {\n" + spb[at:end] + "\n}\n"
        tu.write_text(f'{std}\n#define private public\n#define protected public\n{body}\n#undef private\n#undef protected\n'
                      f'#include "{m / "sensor_exports.inc"}"\n{rcp}')
        cmd = ["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=c++14", "-w", "-fPIC", "-shared",
               f"-I{m}", f"-I{reference / REL_INCLUDE}", str(tu), "-o", str(OUT_SENSORS)]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("g++ failed on the reference's sensor processors:\n" + res.stderr[-6000:])
    return OUT_SENSORS


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    p = build(Path(a.reference), a.force, verbose=True)
    print(p if p else "reference not found and no prebuilt library", file=sys.stderr if not p else sys.stdout)
    p = build_motion(Path(a.reference), a.force, verbose=True)
    print(p if p else "reference not found and no prebuilt motion library", file=sys.stderr if not p else sys.stdout)
    p = build_sensors(Path(a.reference), a.force, verbose=True)
    print(p if p else "reference not found and no prebuilt sensor-model library", file=sys.stderr if not p else sys.stdout)
