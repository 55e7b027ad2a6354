"""Why is k_ray_list 31 us in the host-array frame and 5 us in the device frame?  usage: ray_idle.py <mode>
   0: add(device) + map_feature(fetch=False) + raytracing, back to back      1: ... a synchronize before raytracing
   2: ... map_feature fetching three layers to the host (copy_threads 4)     3: same with copy_threads 0 (the runtime's copies)
   4: synchronize + 200 us of host sleep before raytracing"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from gem_amd import ElevationMap, synth
mode = int(sys.argv[1])
wl = synth.config_c2(reference_filter=True)
m = ElevationMap(wl.length, wl.resolution, device=0)
m.set_lowest_tracking(True)
if mode == 3: m.debug_set("copy_threads", 0)
d = torch.from_numpy(wl.clouds[0]).cuda()
for r in range(30):
    m.mapvar_update(1e-6)
    m.add(wl.frames[0], d)
    if mode in (2, 3): m.map_feature(fetch=True)
    else: m.map_feature(fetch=False)
    if mode in (1, 4): m.synchronize()
    if mode == 4:
        t = time.perf_counter()
        while time.perf_counter() - t < 200e-6: pass
    m.raytracing()
m.synchronize()
print("done", mode)
