#!/bin/bash
# Runs on the MI355X box (through gpurun): rocprofv3 kernel trace + the two HBM PMC passes (separate runs, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes) + an SQ pass on the contract bench (C2), and writes the
# summaries under gpurun_out/profiles/.  Copy what should be judged into profiles/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=${1:-r01}
O=gpurun_out/prof_$R
mkdir -p $O gpurun_out/profiles
CMD="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f --output-format csv -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w --output-format csv -- $CMD > $O/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/sq -o s --output-format csv -- $CMD > $O/sq.log 2>&1
python tools/rocprof_summary.py --trace $O/trace --pmc $O/fetch --pmc $O/write --pmc $O/sq --json gpurun_out/profiles/${R}_c2_bench.json \
    --note "command: rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_*} -- $CMD  (separate runs)" \
    > gpurun_out/profiles/${R}_c2_bench.txt
tail -1 $O/trace.log > gpurun_out/profiles/${R}_c2_bench_line_under_rocprof.json
cat gpurun_out/profiles/${R}_c2_bench.txt
