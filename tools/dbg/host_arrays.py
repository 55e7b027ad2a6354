"""The reference-shaped calls with caller-owned host arrays (Process_points + Fuse [+ Map_feature fetching nine layers]) on a C2 sweep:
host wall time per call, with the call-scoped pinning of the arrays on and off."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from gem_amd import ElevationMap, synth
wl = synth.config_c2()
c = wl.clouds[0]; f = wl.frames[0]
x, y, z = (np.ascontiguousarray(c[:, k]) for k in range(3))
def run(dbg, reps=30):
    m = ElevationMap(wl.length, wl.resolution, debug=dbg)
    m.set_lowest_tracking(True)
    t = {"process_points": [], "fuse": [], "map_feature": [], "add (host xyzi)": [], "get_layer": []}
    for r in range(reps + 3):
        t0 = time.perf_counter(); pp = m.process_points(f, x, y, z)
        t1 = time.perf_counter(); m.fuse(pp["index"], pp["height"], pp["var"])
        t2 = time.perf_counter(); m.map_feature(fetch=True)
        t3 = time.perf_counter(); m.add(f, c); m.synchronize()
        t4 = time.perf_counter(); m.layer("elevation")
        t5 = time.perf_counter()
        if r >= 3:
            for k, v in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)): t[k].append(v * 1e6)
    m.close()
    out = {k: round(float(np.median(v)), 1) for k, v in t.items()}
    out["node_host_arrays (process + fuse + map_feature)"] = round(out["process_points"] + out["fuse"] + out["map_feature"], 1)
    print(json.dumps({"debug": dbg, "us": out}), flush=True)
for dbg in ({"pin_host": 0}, {"pin_host": 1}, {"pin_host": 1, "pin_host_min_bytes": 1 << 20}):
    run(dbg)
