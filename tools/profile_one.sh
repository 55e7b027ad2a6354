#!/bin/bash
# Runs on the MI355X box (through gpurun): rocprofv3 over ONE configuration of tools/bench_configs.py, so that the per-kernel
# averages and the counter traffic belong to that configuration alone.
#   tools/profile_one.sh <tag> <config: c2|c3|c4|c5> [debug knobs, e.g. sort_form=2]
# kernel trace + stats with the product's stream overlap, then every kernel alone (overlap=0) for: kernel trace, FETCH_SIZE,
# WRITE_SIZE and the SQ counters, each in its own run as /opt/skills/guides/MI355X_MICROARCH.md prescribes.
# Summaries go to gpurun_out/profiles/<tag>_<config>*.txt|json; copy what should be judged into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r03}; C=${2:-c4}; K=${3:-}
O=gpurun_out/prof_${R}_${C}
mkdir -p $O gpurun_out/profiles
DBG2=${K:+--debug $K}
DBG1="--debug overlap=0${K:+,$K}"
CFG="--configs $C --reps 10"
timeout 150 rocprofv3 --kernel-trace --stats -d $O/trace2 -o t --output-format csv -- python tools/bench_configs.py $CFG $DBG2 > $O/trace2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $O/trace1 -o t --output-format csv -- python tools/bench_configs.py $CFG $DBG1 > $O/trace1.log 2>&1
PM="--configs $C --reps 4 $DBG1"
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f --output-format csv -- python tools/bench_configs.py $PM > $O/fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w --output-format csv -- python tools/bench_configs.py $PM > $O/write.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/sq -o s --output-format csv -- python tools/bench_configs.py $PM > $O/sq.log 2>&1
if [ "${GEM_PROFILE_LDS:-0}" = "1" ]; then
timeout 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace -d $O/lds -o l --output-format csv -- python tools/bench_configs.py $PM > $O/lds.log 2>&1
EXTRA="--pmc $O/lds"
else EXTRA=""; fi
python tools/rocprof_summary.py --trace $O/trace1 --pmc $O/fetch --pmc $O/write --pmc $O/sq $EXTRA --json gpurun_out/profiles/${R}_${C}.json \
    --note "command: rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_*} -- python tools/bench_configs.py $PM  (separate runs; one stream: every kernel alone; this configuration only)" \
    > gpurun_out/profiles/${R}_${C}.txt
python tools/rocprof_summary.py --trace $O/trace2 --note "command: rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py $CFG $DBG2 (streams overlapped, as the product runs)" > gpurun_out/profiles/${R}_${C}_overlapped.txt
python tools/rocprof_timeline.py $O/trace2 > gpurun_out/profiles/${R}_${C}_timeline_overlapped.txt 2>/dev/null
grep -h '"config"' $O/trace1.log $O/trace2.log > gpurun_out/profiles/${R}_${C}_lines_under_rocprof.jsonl
tail -40 gpurun_out/profiles/${R}_${C}.txt
