"""Pure host cost of one batched call: the host time of calls made while the device is far from full (right after a synchronisation,
only as many as the buffer ring takes without waiting), C4 and C5, with frames that change from call to call and with one batch replayed."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from gem_amd import ElevationMap, synth
def run(name, wl, with_vu, dbg):
    cat = torch.from_numpy(np.concatenate(wl.clouds)).cuda()
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    m = ElevationMap(wl.length, wl.resolution, debug=dbg)
    import bench
    # "changing": four pose sets in turn, as a mapping loop's calls (every call builds and uploads its tables); "replayed": one batch again
    # and again (the library finds its tables cached)
    for label, pbs in (("changing frames", [m.pack_batch(bench.perturbed_frames(wl.frames, j), off, wl.var_updates if with_vu else None) for j in range(4)]),
                       ("one batch replayed", [m.pack_batch(wl.frames, off, wl.var_updates if with_vu else None)] * 4)):
        for k in range(6): m.add_batch(pbs[k & 3], cat)
        first, second = [], []
        for rep in range(30):
            m.synchronize()
            t0 = time.perf_counter(); m.add_batch(pbs[(2 * rep) & 3], cat); t1 = time.perf_counter(); m.add_batch(pbs[(2 * rep + 1) & 3], cat); t2 = time.perf_counter()
            first.append((t1 - t0) * 1e6); second.append((t2 - t1) * 1e6)
        print(json.dumps({"config": name, "frames": label, "debug": dbg, "host_us_first_call_after_sync_median": round(float(np.median(first)), 1), "second_call_median": round(float(np.median(second)), 1)}), flush=True)
    m.close()
run("c4", synth.config_c4(n_sweeps=32), True, {})
run("c5", synth.config_c5(), False, {})
