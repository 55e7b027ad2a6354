N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
for v in base numa base numa; do
  cp gem_amd/lib/variant_$v.so gem_amd/lib/libgem_hip.so
  echo "== $v, caller on node0"; timeout 100 taskset -c $N0 python tools/dbg/host_path.py 4 2>&1 | tail -1 | cut -c1-150
  echo "== $v, caller on node1"; timeout 100 taskset -c $N1 python tools/dbg/host_path.py 4 2>&1 | tail -1 | cut -c1-150
done
