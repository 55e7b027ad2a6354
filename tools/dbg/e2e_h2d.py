#!/usr/bin/env python3
"""gem_add from a HOST array, call after call (bench.py's e2e_with_h2d loop), per number of copy threads: wall per sweep, host time
per call, and the library's own account of where the host time went (staging copy / DMA enqueue)."""
import sys, time, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from gem_amd import ElevationMap, synth, _lib
wl = synth.config_c2()
m = ElevationMap(wl.length, wl.resolution)
lib = _lib.load()
host = wl.clouds[0]
fp = wl.frames[0].to_struct()
n = host.shape[0]
ptr = host.ctypes.data_as(C.c_void_p)
def get(k):
    return m.debug_get(k)
for rep in range(2):
    for thr in (4, 1, 2, 8, 0):
        m.debug_set("copy_threads", thr)
        for _ in range(10): m.add(wl.frames[0], host)
        m.synchronize()
        a0, b0 = get("xfer_upload_memcpy_ns"), get("xfer_upload_enqueue_ns")
        t0 = time.perf_counter()
        for _ in range(200): lib.gem_add(m._h, C.byref(fp), n, ptr, None, None)
        t1 = time.perf_counter(); m.synchronize(); dt = (time.perf_counter() - t0) / 200
        a1, b1 = get("xfer_upload_memcpy_ns"), get("xfer_upload_enqueue_ns")
        print(f"copy_threads {thr}: {dt * 1e6:6.1f} us per sweep ({n / dt / 1e9:.2f} G points/s), host {(t1 - t0) / 200 * 1e6:6.1f} us of which staging copy {(a1 - a0) / 200e3:5.1f} us, DMA enqueue {(b1 - b0) / 200e3:5.1f} us")
