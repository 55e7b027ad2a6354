"""Is C4's walk bound by its heaviest blocks?  The same batch with the points nearer than R metres to the sensor removed (the blocks under
the sensor go with them), every kernel alone on the GPU (overlap = 0) and with the streams overlapped: the walk's time against the
records that are left.  usage: python tools/dbg/c4_tail.py"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from gem_amd import ElevationMap, synth

wl = synth.config_c4()
for R in (0.0, 2.0, 3.0, 4.0, 6.0):
    clouds = [c[np.hypot(c[:, 0], c[:, 1]) >= R] for c in wl.clouds]
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in clouds])])
    cat = torch.from_numpy(np.concatenate(clouds)).cuda()
    for dbg in ({"overlap": 0}, {}):
        m = ElevationMap(wl.length, wl.resolution, debug=dbg)
        pb = m.pack_batch(wl.frames, off, wl.var_updates)
        for _ in range(5):
            m.add_batch(pb, cat)
        m.synchronize()
        m.set_timing(True); m.stats(reset=True)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        t0 = time.perf_counter()
        for _ in range(20):
            m.add_batch(pb, cat)
        m.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e6
        st = m.stats()
        kept = None
        print(f"R {R:3.1f} points {int(off[-1]):8d} {'alone  ' if dbg else 'overlap'}: call {wall:7.1f} us  walk {1e3 * st['ms_walk'] / max(st['launches_walk'], 1):6.1f}  sort kernels {1e3 * sum(st['ms_sort']) / max(st['launches_walk'], 1):6.1f}", flush=True)
        m.close()
