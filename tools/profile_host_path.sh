#!/bin/bash
# Runs on the MI355X box (through gpurun): the node's frame with caller-owned HOST arrays (bench.node_host_arrays): per-call times
# for the runtime's pageable path (copy_threads 0) and the library's pinned staging + copy threads, the raw link / memcpy rates of
# the box, and a rocprofv3 kernel trace of the staged path.   tools/profile_host_path.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r04}
O=gpurun_out/prof_${R}_host_path
mkdir -p $O gpurun_out/profiles
OUT=gpurun_out/profiles/${R}_host_path.txt
{
echo "# the node's frame with caller-owned host arrays: python tools/dbg/host_path.py 0 4 8 0 4  (us; medians of 60 frames)"
timeout 200 python tools/dbg/host_path.py 0 4 8 0 4 2>/dev/null
echo
echo "# raw rates of this box: tools/ubench/bin/pcie (13 MB = Map_feature's nine 600 x 600 layers, whole and in pieces; us)"
timeout 100 tools/ubench/bin/pcie 2>/dev/null
echo
echo "# python tools/dbg/pcie_rates.py (torch copies; the memcpy column reads a cache-resident source)"
timeout 100 python tools/dbg/pcie_rates.py 2>/dev/null
echo
} > $OUT
timeout 150 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python tools/dbg/host_path.py 4 > $O/trace.log 2>&1
python tools/rocprof_summary.py --trace $O/trace --note "command: rocprofv3 --kernel-trace --stats -- python tools/dbg/host_path.py 4  (64 frames: Mapvar_update + Process_points + Fuse + Map_feature + Raytracing with host arrays)" >> $OUT
tail -30 $OUT
