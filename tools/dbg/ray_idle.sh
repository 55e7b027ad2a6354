cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
for mode in 0 1 2 3 4; do
  O=gpurun_out/ray_idle_$mode
  timeout 100 rocprofv3 --kernel-trace --stats -d $O -o t --output-format csv -- python tools/dbg/ray_idle.py $mode > $O.log 2>&1
  echo "== mode $mode"; python tools/rocprof_summary.py --trace $O 2>/dev/null | grep -E "k_ray_list|k_raytracing|k_map_feature|k_copy_list" | cut -c1-110
done
