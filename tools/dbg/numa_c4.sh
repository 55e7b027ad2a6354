N0=$(cat /sys/devices/system/node/node0/cpulist); N1=$(cat /sys/devices/system/node/node1/cpulist 2>/dev/null)
echo "gpu numa node: $(cat /sys/class/drm/card0/device/numa_node 2>/dev/null) (card1: $(cat /sys/class/drm/card1/device/numa_node 2>/dev/null))"
for rep in 1 2; do
echo "== C4 caller on node0"; timeout 100 taskset -c $N0 python tools/dbg/c4_ab.py "" 2>&1 | grep default | head -1
echo "== C4 caller on node1"; timeout 100 taskset -c $N1 python tools/dbg/c4_ab.py "" 2>&1 | grep default | head -1
done
echo "== C5 caller on node0"; timeout 100 taskset -c $N0 python tools/dbg/c4_ab.py --c5 "" 2>&1 | grep default | head -1
echo "== C5 caller on node1"; timeout 100 taskset -c $N1 python tools/dbg/c4_ab.py --c5 "" 2>&1 | grep default | head -1
echo "== C2 bench caller on node0"; timeout 200 taskset -c $N0 python bench.py --steps 400 --warmup 40 2>/dev/null | cut -c1-200
echo "== C2 bench caller on node1"; timeout 200 taskset -c $N1 python bench.py --steps 400 --warmup 40 2>/dev/null | cut -c1-200
