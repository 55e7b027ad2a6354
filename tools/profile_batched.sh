#!/bin/bash
# Runs on the MI355X box (through gpurun): rocprofv3 over the batched configurations (tools/bench_configs.py): BASELINE
# configs[3] (C4: 32 sweeps + variance increments) and configs[4] (C5: 10^7 points -> 2400^2), and the depth image (C3).
#   kernel trace + stats (two streams, as the product runs; and one stream, every kernel alone)
#   PMC passes in their own runs (FETCH_SIZE | WRITE_SIZE | SQ_*), as /opt/skills/guides/MI355X_MICROARCH.md prescribes
# Every rocprofv3 call is bounded.  Summaries go to gpurun_out/profiles/; copy what should be judged into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r02}
O=gpurun_out/prof_batched_$R
mkdir -p $O gpurun_out/profiles
CFG="--configs c3,c4,c5 --reps 10"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace2 -o t --output-format csv -- python tools/bench_configs.py $CFG > $O/trace2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace1 -o t --output-format csv -- python tools/bench_configs.py $CFG --debug overlap=0 > $O/trace1.log 2>&1
PM="--configs c4,c5 --reps 5 --debug overlap=0"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f --output-format csv -- python tools/bench_configs.py $PM > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w --output-format csv -- python tools/bench_configs.py $PM > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/sq -o s --output-format csv -- python tools/bench_configs.py $PM > $O/sq.log 2>&1
python tools/rocprof_summary.py --trace $O/trace1 --pmc $O/fetch --pmc $O/write --pmc $O/sq --json gpurun_out/profiles/${R}_batched.json \
    --note "command: rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_*} -- python tools/bench_configs.py $PM  (separate runs; one stream: every kernel alone; C4 then C5 in one process: per-kernel averages mix both configurations, the per-configuration split is in ${R}_batched_timeline.txt)" \
    > gpurun_out/profiles/${R}_batched.txt
python tools/rocprof_summary.py --trace $O/trace2 --note "command: rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py $CFG (two streams, as the product runs)" > gpurun_out/profiles/${R}_batched_two_streams.txt
python tools/rocprof_timeline.py $O/trace1 > gpurun_out/profiles/${R}_batched_timeline.txt
python tools/rocprof_timeline.py $O/trace2 > gpurun_out/profiles/${R}_batched_timeline_two_streams.txt
grep -h '"config"' $O/trace1.log > gpurun_out/profiles/${R}_batched_lines_under_rocprof.jsonl
tail -30 gpurun_out/profiles/${R}_batched.txt
