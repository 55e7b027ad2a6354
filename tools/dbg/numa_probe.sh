lscpu | grep -i -E "numa|socket|model name|^CPU\(s\)|thread" | head -12
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) $(cat $d/local_cpulist 2>/dev/null)"; done | head -4
nproc
N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
echo "node0 $N0 ; node1 $N1"
echo "== unrestricted"; timeout 100 python tools/dbg/host_path.py 4 8 2>&1 | tail -2
echo "== taskset node0"; timeout 100 taskset -c $N0 python tools/dbg/host_path.py 4 8 2>&1 | tail -2
if [ -n "$N1" ]; then echo "== taskset node1"; timeout 100 taskset -c $N1 python tools/dbg/host_path.py 4 8 2>&1 | tail -2; fi
echo "== taskset 8 cpus"; timeout 100 taskset -c 0-7 python tools/dbg/host_path.py 4 8 2>&1 | tail -2
