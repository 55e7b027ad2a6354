#!/bin/bash
# rocprofv3 kernel trace + marker trace of a short node sequence with the library's roctx ranges on (gem_debug_set "roctx"):
# which entry point enqueued which kernels.  Run through gpurun; the summary goes to gpurun_out/profiles/<tag>_roctx_trace.txt.
#   tools/roctx_trace.sh [tag]
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r06}
O=gpurun_out/prof_${R}_roctx
mkdir -p $O gpurun_out/profiles
cat > /tmp/roctx_seq.py <<'PY'
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from gem_amd import ElevationMap, synth
wl = synth.config_c2()
m = ElevationMap(wl.length, wl.resolution)
m.set_lowest_tracking(True)
m.debug_set("roctx", 1)
d = torch.from_numpy(wl.clouds[0]).cuda()
for k in range(8):
    m.mapvar_update(1e-6)
    m.add(wl.frames[0], d)
    m.map_feature(fetch=False)
    m.raytracing()
m.synchronize()
PY
timeout 200 rocprofv3 --kernel-trace --marker-trace -d $O -o t --output-format csv -- python /tmp/roctx_seq.py > $O/run.log 2>&1
python - "$O" > gpurun_out/profiles/${R}_roctx_trace.txt <<'PY'
import csv, glob, sys
o = sys.argv[1]
mk = sorted(glob.glob(o + "/**/*marker_api_trace.csv", recursive=True))
kt = sorted(glob.glob(o + "/**/*kernel_trace.csv", recursive=True))
print("# rocprofv3 --kernel-trace --marker-trace -- (8 frames of: mapvar_update, add (lowest tracking on), map_feature, raytracing) with gem_debug_set roctx = 1")
if not mk or not kt:
    print("# no marker / kernel trace found:", mk, kt); sys.exit(0)
ranges = []
for r in csv.DictReader(open(mk[0])):
    name = r.get("Function") or r.get("Name") or ""
    ranges.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
kern = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in csv.DictReader(open(kt[0]))]
ranges.sort(); kern.sort()
t0 = ranges[0][0] if ranges else 0
print(f"# {len(ranges)} ranges, {len(kern)} kernel dispatches; times in us from the first range (host clock for ranges, device dispatch stamps for kernels)")
for a, b, n in ranges[-16:]:
    print(f"range  {(a - t0) / 1e3:10.1f} .. {(b - t0) / 1e3:10.1f}  {n}")
for a, b, n in kern[-20:]:
    print(f"kernel {(a - t0) / 1e3:10.1f} .. {(b - t0) / 1e3:10.1f}  {n}")
PY
tail -40 gpurun_out/profiles/${R}_roctx_trace.txt
