#!/bin/bash
# Runs on the MI355X box (through gpurun): rocprofv3 kernel trace of tools/bench_configs.py (every BASELINE configuration).
# Kernel trace only -- a PMC pass over the multi-millisecond batched kernels took > 10 minutes of box time in round 1.
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r01}
O=gpurun_out/prof_cfg_$R
mkdir -p $O gpurun_out/profiles
timeout 240 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python tools/bench_configs.py --reps 20 > $O/trace.log 2>&1
python tools/rocprof_summary.py --trace $O/trace --note "command: rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py --reps 20 (C2, Map_feature, node sequence, C3, C4, C5 in one process)" \
    > gpurun_out/profiles/${R}_configs_kernel_trace.txt
cat gpurun_out/profiles/${R}_configs_kernel_trace.txt
