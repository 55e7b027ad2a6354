"""Pure host cost of one batched call: the host time of calls made while the device is far from full (right after a synchronisation,
only as many as the buffer ring takes without waiting), C4 and C5, with and without the cached tables."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from gem_amd import ElevationMap, synth
def run(name, wl, with_vu, dbg):
    cat = torch.from_numpy(np.concatenate(wl.clouds)).cuda()
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    m = ElevationMap(wl.length, wl.resolution, debug=dbg)
    pb = m.pack_batch(wl.frames, off, wl.var_updates if with_vu else None)
    for _ in range(6): m.add_batch(pb, cat)
    first, second = [], []
    for rep in range(30):
        m.synchronize()
        t0 = time.perf_counter(); m.add_batch(pb, cat); t1 = time.perf_counter(); m.add_batch(pb, cat); t2 = time.perf_counter()
        first.append((t1 - t0) * 1e6); second.append((t2 - t1) * 1e6)
    print(json.dumps({"config": name, "debug": dbg, "host_us_first_call_after_sync_median": round(float(np.median(first)), 1), "second_call_median": round(float(np.median(second)), 1)}), flush=True)
    m.close()
for dbg in ({"cache_tables": 0}, {"cache_tables": 1}):
    run("c4", synth.config_c4(n_sweeps=32), True, dbg)
    run("c5", synth.config_c5(), False, dbg)
