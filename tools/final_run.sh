#!/bin/bash
# One consolidated validation / evidence run on the MI355X box (through gpurun): the GPU suite, the contract bench, every configuration,
# the rocprofv3 summaries per configuration, the soak.  Outputs under gpurun_out/ (copy what should be judged into profiles/).
#   tools/final_run.sh [tag, default r06] [skip the GPU suite: nosuite]
set -u
R=${1:-r06}
mkdir -p gpurun_out/final
if [ "${2:-}" != nosuite ]; then
timeout -s KILL 1100 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; tail -3 gpurun_out/final/pytest_gpu.log
fi
timeout -s KILL 400 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err; cut -c1-300 gpurun_out/final/bench_line.json
timeout -s KILL 300 python tools/bench_configs.py > gpurun_out/final/bench_configs.jsonl 2>/dev/null; cut -c1-200 gpurun_out/final/bench_configs.jsonl
timeout -s KILL 400 bash tools/profile_c2.sh $R > gpurun_out/final/prof_c2_bench.log 2>&1
timeout -s KILL 500 bash tools/profile_one.sh $R c4 > gpurun_out/final/prof_c4.log 2>&1
timeout -s KILL 500 bash tools/profile_one.sh $R c5 > gpurun_out/final/prof_c5.log 2>&1
timeout -s KILL 400 bash tools/profile_one.sh $R c3 > gpurun_out/final/prof_c3.log 2>&1
timeout -s KILL 400 bash tools/profile_one.sh $R c2 > gpurun_out/final/prof_c2.log 2>&1
timeout -s KILL 100 python tools/frame_phases.py > gpurun_out/final/c2_frame_phases.txt 2>/dev/null
timeout -s KILL 100 python tools/block_phases.py c4 > gpurun_out/final/c4_block_phases.txt 2>/dev/null
timeout -s KILL 100 python tools/block_phases.py c5 > gpurun_out/final/c5_block_phases.txt 2>/dev/null
timeout -s KILL 200 python bench.py --workload c5 --steps 40 --warmup 5 > gpurun_out/final/bench_c5_n1.json 2>/dev/null
timeout -s KILL 100 python tools/dbg/host_cost.py > gpurun_out/final/host_cost.txt 2>/dev/null
timeout -s KILL 100 python tools/dbg/e2e_h2d.py > gpurun_out/final/e2e_h2d.txt 2>/dev/null
timeout -s KILL 100 python tools/dbg/host_c3.py > gpurun_out/final/host_c3.txt 2>/dev/null
timeout -s KILL 100 python tools/dbg/host_c2.py > gpurun_out/final/host_c2.txt 2>/dev/null
timeout -s KILL 300 bash tools/profile_host_path.sh $R > gpurun_out/final/prof_host_path.log 2>&1
timeout -s KILL 60 tools/ubench/bin/handover > gpurun_out/final/ubench_handover.txt 2>/dev/null
timeout -s KILL 60 tools/ubench/bin/sync > gpurun_out/final/ubench_sync.txt 2>/dev/null
timeout -s KILL 100 python tools/dbg/sync_cost.py >> gpurun_out/final/ubench_sync.txt 2>/dev/null
timeout -s KILL 200 python tools/dbg/walk_defer.py > gpurun_out/final/walk_defer.txt 2>/dev/null
timeout -s KILL 100 bash tools/roctx_trace.sh $R > gpurun_out/final/roctx_trace.log 2>&1
timeout -s KILL 200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver_style.json 2>/dev/null; cut -c1-200 gpurun_out/final/bench_driver_style.json
timeout -s KILL 200 python tools/fuzz_parity.py --seconds 150 --seed 9 > gpurun_out/final/fuzz9.log 2>&1; tail -1 gpurun_out/final/fuzz9.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
