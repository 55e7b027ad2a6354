import sys, time, numpy as np
sys.path.insert(0, ".")
from gem_amd import ElevationMap, synth
wl4 = synth.config_c4(n_sweeps=32)
m = ElevationMap(wl4.length, wl4.resolution)
pb = m.pack_batch(wl4.frames, np.concatenate([[0], np.cumsum([c.shape[0] for c in wl4.clouds])]), wl4.var_updates)
for thr in (4, 8, 12, 16, 0):
    m.debug_set("copy_threads", thr)
    for _ in range(2): m.add_batch_host(pb, wl4.clouds)
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(6): m.add_batch_host(pb, wl4.clouds)
    m.synchronize(); dt = (time.perf_counter() - t0) / 6
    print(f"copy_threads {thr}: host batch {dt*1e6:.0f} us, link {67.1/dt/1e3:.1f} GB/s")
