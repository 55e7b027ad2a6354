#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the contract bench's dominant kernel, one rocprofv3 pass each (through gpurun)
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5m
CMD="python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/r5m/$c -o f --output-format csv -- $CMD > gpurun_out/r5m/$c.log 2>&1
  python tools/rocprof_summary.py --pmc gpurun_out/r5m/$c 2>/dev/null | grep "k_frame"
done
