import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
from gem_amd import ElevationMap, synth
import oracle
ns = int(sys.argv[1]); use_var = int(sys.argv[2])
wl = synth.config_c4(n_sweeps=ns)
cat = torch.from_numpy(np.concatenate(wl.clouds)).cuda()
off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
m = ElevationMap(wl.length, wl.resolution)
ref = oracle.OracleMap(wl.length, wl.resolution)
for rep in range(2):
    m.add_batch(wl.frames, cat, off, wl.var_updates if use_var else None)
    m.synchronize()
    for k in range(ns):
        if use_var: ref.mapvar_update(wl.var_updates[k])
        ref.add(wl.frames[k], wl.clouds[k])
    e = m.layer("elevation"); print("ns", ns, "var", use_var, "rep", rep, "equal", np.array_equal(e, ref.layer("elevation")), np.array_equal(m.layer("variance"), ref.layer("variance")), flush=True)
