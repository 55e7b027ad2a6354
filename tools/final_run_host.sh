#!/bin/bash
# The part of tools/final_run.sh that the host-array work of round 4 touches (k_ray_list, k_map_feature, the staging path): the GPU
# suite, the contract bench line, C2's profiles (the node-shaped frame is in them), the host-array frame, the soak.
set -u
mkdir -p gpurun_out/final
timeout -s KILL 900 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.log 2>&1; tail -3 gpurun_out/final/pytest_gpu.log
timeout -s KILL 300 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err; cut -c1-300 gpurun_out/final/bench_line.json
timeout -s KILL 400 bash tools/profile_c2.sh r04 > gpurun_out/final/prof_c2_bench.log 2>&1
timeout -s KILL 400 bash tools/profile_one.sh r04 c2 > gpurun_out/final/prof_c2.log 2>&1
timeout -s KILL 300 bash tools/profile_host_path.sh r04 > gpurun_out/final/prof_host_path.log 2>&1; head -8 gpurun_out/profiles/r04_host_path.txt | cut -c1-200
timeout -s KILL 150 python tools/fuzz_parity.py --seconds 100 --seed 5 > gpurun_out/final/fuzz5.log 2>&1; tail -1 gpurun_out/final/fuzz5.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -2 gpurun_out/final/smoke.log
