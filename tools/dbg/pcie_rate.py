"""PCIe-inclusive rate of the host-buffer entries (never bench.py's `value`): gem_add (XYZI host array), gem_add_aos (PCL structs)
and the reference-shaped pair gem_process_points + gem_fuse (host SoA arrays in, host arrays out, host arrays in again)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from gem_amd import ElevationMap, synth

wl = synth.config_c4(n_sweeps=8)
m = ElevationMap(wl.length, wl.resolution)
n = wl.clouds[0].shape[0]
def rate(fn, reps=100):
    for _ in range(10): fn()
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    m.synchronize(); dt = (time.perf_counter() - t0) / reps
    return dt * 1e6, n / dt
k = [0]
def add_host():
    m.add(wl.frames[k[0] % 8], wl.clouds[k[0] % 8]); k[0] += 1
pts = [np.zeros((n, 8), np.float32) for _ in range(8)]
for i in range(8):
    pts[i][:, :3] = wl.clouds[i][:, :3]; pts[i][:, 6] = wl.clouds[i][:, 3]
def add_aos():
    m.add_aos(wl.frames[k[0] % 8], pts[k[0] % 8], off_rgb=-1); k[0] += 1
def pp_fuse():
    c = wl.clouds[k[0] % 8]; f = wl.frames[k[0] % 8]; k[0] += 1
    o = m.process_points(f, c[:, 0], c[:, 1], c[:, 2]); m.fuse(o["index"], o["height"], o["var"])
for name, fn in (("gem_add (host XYZI, 16 B/pt over PCIe)", add_host), ("gem_add_aos (32-byte PCL structs over PCIe)", add_aos),
                 ("gem_process_points + gem_fuse (reference-shaped host arrays)", pp_fuse)):
    us, r = rate(fn)
    print(f"{name}: {us:.1f} us per 131072-point sweep, {r:.3g} points/s", flush=True)
