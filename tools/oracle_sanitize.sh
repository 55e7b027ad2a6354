#!/bin/bash
# The CPU oracle (and the compiled reference's stand-ins) under AddressSanitizer + UndefinedBehaviorSanitizer:
# builds an instrumented libgem_oracle.so, runs the CPU test files that exercise it, restores the normal build.
# (The reference's own gpu_process.cu is NOT clean: see DESIGN.md section 5 for the two defects ASan finds in it.)
set -eu
cd "$(dirname "$0")/.."
gcc -O1 -g -std=c11 -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared \
    -o /tmp/libgem_oracle_asan.so oracle/gem_oracle.c oracle/gem_oracle_motion.c oracle/gem_oracle_feature.c oracle/gem_oracle_raytrace.c \
    oracle/gem_oracle_mt.c oracle/gem_oracle_show.c oracle/gem_oracle_color.c -lm -lpthread
make -s -C oracle libgem_oracle.so
cp oracle/libgem_oracle.so /tmp/libgem_oracle_plain.so
trap 'cp /tmp/libgem_oracle_plain.so oracle/libgem_oracle.so; touch oracle/libgem_oracle.so' EXIT
cp /tmp/libgem_oracle_asan.so oracle/libgem_oracle.so
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_oracle_kat.py tests/test_golden.py tests/test_reference_compiled.py tests/test_map_feature.py \
    tests/test_show.py tests/test_colorize.py \
    -x -q -m "not gpu" -p no:cacheprovider
