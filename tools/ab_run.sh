#!/bin/bash
# A/B of build variants (gem_amd/lib_ab/<name>, python -m gem_amd.build --variant ...) on the GPU box: the same tool per variant, interleaved, twice.
# Run it THROUGH gpurun only: it swaps gem_amd/lib/libgem_hip.so of the box's scratch copy of the tree (and puts the product library back).
#   tools/ab_run.sh OUTDIR "tool command" variant1 variant2 ...     ("base" = the product library)
set -u
out=$1; shift; cmd=$1; shift
mkdir -p "$out"
lib=gem_amd/lib/libgem_hip.so
cp -p $lib /tmp/libgem_hip_base.so
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then cp -p /tmp/libgem_hip_base.so $lib; else cp -p gem_amd/lib_ab/$v/libgem_hip.so $lib; fi
    touch $lib
    echo "== $v (rep $rep)" >> "$out/ab.txt"
    timeout 200 $cmd >> "$out/ab.txt" 2>> "$out/ab.err"
  done
done
cp -p /tmp/libgem_hip_base.so $lib; touch $lib
