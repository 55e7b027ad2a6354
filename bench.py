#!/usr/bin/env python3
"""bench.py -- fused points/s of the GEM point-cloud -> elevation-grid hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1 (the contract line).  A "step" is one pass of the hot path over one batch of synthetic input: one 64-beam LiDAR sweep
(64 x 2048 = 131 072 XYZI points, BASELINE.json configs[1]) projected, binned and Kalman-fused into the robot-centric
600 x 600 @ 0.05 m map with gem_add_device (inputs already resident in HBM).  Steps cycle through 8 distinct seeded sweeps
(moving sensor).  W warm-up steps, then EXACTLY K steps are timed between synchronisations -- and, because K steps of ~10 us are a
sub-millisecond sample, that K-step region is repeated until at least 50 ms have been measured; `ms_per_step` is the MEDIAN
over the repetitions (`timed_repetitions` says how many).
What was timed is checked: (1) `parity_checked` -- the same code paths on a fresh map reproduce the committed golden digests
(tests/golden/digests.json: the C2 stream, the C4 batch), no oracle involved; (2) `cpu_baseline.replay_matches_timed_map` -- the
CPU oracle replays the very sequence of sweeps the timed map has seen and the two maps are compared bit for bit.
`roofline` describes the dominant kernel (dispatch time stamps on the stream it runs on, second loop), `batched_c4` the
bandwidth-regime configuration (BASELINE configs[3]), `cpu_baseline` the oracle on this box's host cores.

N > 1 (one rank per GPU; launched by torch.distributed.run -- or by this script itself when WORLD_SIZE is not set: it re-executes
under torch.distributed.run with N ranks, and exits non-zero when fewer than N devices are visible): BASELINE configs[4], STRONG
scaling -- the same 10^7-point aggregated cloud -> 2400 x 2400 map on N ranks per step: every rank projects / bins / sorts its
N-th of the points, the sorted records go to the tile-row strip owners (RCCL send / recv), the owners fuse, and the fused strips
are all-gathered over xGMI (gem_add_sharded_device + gem_allgather_layers).  value = 10^7 points / step time (max over ranks).
The same workload on ONE GPU (plain gem_add_batch_device) is in both kinds of line: `c5_one_gpu` at N = 1, and
`one_gpu_us_per_step` / `speedup_vs_one_gpu` (measured by rank 0 in the same process) at N > 1.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import platform
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0    # the same guide: 6.29 TB/s measured copy rate (SURVEY 8d asks for both fractions)
N_DISTINCT = 8
MIN_TIMED_S = 0.05


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU-oracle leg (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary (batched C4) measurement")
    ap.add_argument("--workload", choices=["c2", "c5"], default=None,
                    help="default: c2 (the headline configuration) at --gpus 1, c5 at --gpus N > 1; `--workload c5 --gpus 1` times C5 through the "
                         "same sharded entry points on one rank -- the like-for-like N = 1 point of the scaling curve")
    return ap.parse_args()


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


# ---- CPU leg: baseline AND checker -------------------------------------------------------------------------------------------
def cpu_baseline(wl, n_sweeps_timed_map: int, timed_layers, budget_s: float):
    """The CPU oracle (plain-C port of the reference semantics) on this box's host cores.
      all cores : gemo_add_batch_mt -- cells split into row strips, every thread scans the index array in order --
                  REPLAYING the sequence of sweeps the timed GPU map has seen, which is then compared with it;
      1 thread  : gemo_add on a bounded sample (`cpu_baseline_1t`);
      reference : oracle/_ref (the reference's own gpu_process.cu compiled for the CPU: its literal O(L^2 N) G_fuse) on C1."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle
    from gem_amd import synth
    ncpu = os.cpu_count() or 1
    n_per = wl.clouds[0].shape[0]
    cat = np.concatenate(wl.clouds)
    off = np.arange(N_DISTINCT + 1, dtype=np.int64) * n_per
    t_begin = time.perf_counter()

    def run(m, sweeps, nt):                                  # `sweeps` consecutive sweeps of the cycle, from the map's own position
        done = 0
        while done < sweeps:
            k0 = m._pos % N_DISTINCT
            take = min(sweeps - done, N_DISTINCT - k0)
            m.add_batch_mt(wl.frames[k0:k0 + take], cat[off[k0]:off[k0 + take]], off[k0:k0 + take + 1] - off[k0], None, nt)
            m._pos += take; done += take

    # thread count: the fastest of a few on a short sample (more threads = shorter strips but two barriers per sweep)
    best = None
    scaling = []                                             # the row SURVEY 8d asks for beside "all host cores": why `cores` is what it is
    for nt in sorted({min(ncpu, v) for v in (1, 8, 16, 32, 64, 128, ncpu)}):
        m = oracle.OracleMap(wl.length, wl.resolution); m._pos = 0
        run(m, N_DISTINCT, nt)                               # warm
        t0 = time.perf_counter(); run(m, 2 * N_DISTINCT, nt); dt = time.perf_counter() - t0
        rate = 2 * N_DISTINCT * n_per / dt
        scaling.append({"threads": nt, "points_per_s": rate})
        if best is None or rate > best[1]:
            best = (nt, rate)
    nt, rate_est = best
    out = {"unit": "points/s", "kind": "port", "cores": nt, "host_cores": ncpu, "cpu_model": cpu_model(), "thread_scaling": scaling,
           "thread_scaling_note": "gemo_add_batch_mt on 32 sweeps of the workload per thread count; every thread scans the whole index array of a "
                                  "sweep for its row strip (SURVEY 8d(ii)), so beyond the knee more threads only add scans and barriers"}
    left = budget_s - (time.perf_counter() - t_begin) - 4.0
    replay_cost = n_sweeps_timed_map * n_per / rate_est
    if replay_cost <= left:
        m = oracle.OracleMap(wl.length, wl.resolution); m._pos = 0
        t0 = time.perf_counter(); run(m, n_sweeps_timed_map, nt); dt = time.perf_counter() - t0
        out["value"] = n_sweeps_timed_map * n_per / dt
        ok = bool(np.array_equal(m.layer("elevation"), timed_layers[0]) and np.array_equal(m.layer("variance"), timed_layers[1]))
        out["replay_matches_timed_map"] = ok
        out["sample"] = (f"the {n_sweeps_timed_map} sweeps x {n_per} pts the timed GPU map has fused (warm-up + timed repetitions), replayed by "
                         f"oracle/gem_oracle_mt.c gemo_add_batch_mt on {nt} threads ({dt:.1f} s) and compared with it bit for bit")
    else:
        sweeps = max(N_DISTINCT, int(min(left, 15.0) * rate_est / n_per))
        m = oracle.OracleMap(wl.length, wl.resolution); m._pos = 0
        t0 = time.perf_counter(); run(m, sweeps, nt); dt = time.perf_counter() - t0
        out["value"] = sweeps * n_per / dt
        out["replay_matches_timed_map"] = None
        out["sample"] = (f"{sweeps} sweeps x {n_per} pts of the same workload on {nt} threads ({dt:.1f} s); the timed map has fused "
                         f"{n_sweeps_timed_map} sweeps, too many to replay inside the CPU budget")
    # one thread: the faithful sequential form
    ref = oracle.OracleMap(wl.length, wl.resolution)
    ref.add(wl.frames[0], wl.clouds[0])
    t0, k, pts = time.perf_counter(), 0, 0
    while time.perf_counter() - t0 < 3.0:
        i = k % N_DISTINCT
        ref.add(wl.frames[i], wl.clouds[i]); pts += n_per; k += 1
    dt = time.perf_counter() - t0
    one = {"value": pts / dt, "unit": "points/s", "cores": 1, "kind": "port",
           "sample": f"{k} sweeps x {n_per} pts, oracle/gem_oracle.c gemo_add, 1 thread ({dt:.1f} s)"}
    # the reference's own code (kernels emulated sequentially on the CPU): only C1 is affordable, its G_fuse is O(L^2 N)
    lit = None
    try:
        import ref as refmod
        if refmod.lib() is not None:
            c1 = synth.config_c1()
            c = c1.clouds[0]
            # (the reference's Init printf's to stdout, which belongs to the ONE JSON line: park fd 1 on /dev/null meanwhile)
            sys.stdout.flush()
            keep, null = os.dup(1), os.open(os.devnull, os.O_WRONLY)
            os.dup2(null, 1)
            try:
                rm = refmod.RefMap(c1.length, c1.resolution)
                t0 = time.perf_counter()
                pp = rm.process_points(c1.frames[0], c[:, 0], c[:, 1], c[:, 2])
                rm.fuse(pp["index"], pp["height"], pp["var"])
                dt = time.perf_counter() - t0
            finally:
                os.dup2(keep, 1); os.close(keep); os.close(null)
            lit = {"value": c.shape[0] / dt, "unit": "points/s", "cores": 1, "kind": "reference",
                   "sample": f"C1 only (10 000 pts -> 200 x 200): Process_points + Fuse of the reference's gpu_process.cu compiled for the CPU "
                             f"(oracle/_ref), every kernel thread run sequentially -- G_fuse scans all N points per cell ({dt * 1e3:.0f} ms)"}
    except Exception as e:      # the compiled reference is optional
        lit = {"error": str(e)[:200]}
    return out, one, lit


def pmc_traffic(kernel_prefix: str, pattern: str = "r*_c2_bench.json"):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 PMC summary of its configuration (profiles/rNN_c2_bench.json:
    this command; profiles/rNN_c4.json: tools/profile_one.sh on C4 alone), corrected as /opt/skills/guides/MI355X_MICROARCH.md
    prescribes (FETCH_SIZE doubled)."""
    best = None
    for f in sorted((ROOT / "profiles").glob(pattern)):
        try:
            doc = json.loads(f.read_text())
        except Exception:
            continue
        cand = None
        for k, v in doc.get("counters", {}).items():
            if k.split("<")[0].split("(")[0].endswith(kernel_prefix) and "hbm_bytes_high" in v:
                kr = doc.get("kernels", {}).get(k, {})                      # (several instantiations: the one the timed loop launches most)
                if cand is None or kr.get("calls", 0) > cand[3]:
                    cand = (f.name, v, kr.get("avg_us"), kr.get("calls", 0))
        if cand:
            best = cand[:3]
    return best


def golden():
    return json.loads((ROOT / "tests" / "golden" / "digests.json").read_text())


def check_c2_stream(emap_cls, dev, torch):
    """The device stream of single sweeps on a fresh map against the committed digest (16 sweeps: 8 distinct, two rounds)."""
    from gem_amd import synth
    d = golden()["c2_stream16"]
    wl = synth.config_c4(n_sweeps=8, seed0=100)
    if sha(np.concatenate(wl.clouds)) != d["cloud"]:
        return False, "generator drift: regenerate tests/golden"
    dc = [torch.from_numpy(c).to(dev) for c in wl.clouds]
    m = emap_cls(wl.length, wl.resolution, device=dev.index)
    for k in range(16):
        m.add(wl.frames[k % 8], dc[k % 8])
    ok = sha(m.layer("elevation")) == d["elevation"] and sha(m.layer("variance")) == d["variance"]
    m.close()
    return ok, "16-sweep device stream vs tests/golden/digests.json c2_stream16"


def perturbed_frames(frames, j: int):
    """The same sweeps seen from a pose `j` tenths of a millimetre further along x: a stream of batches whose frames DIFFER from call
    to call, as a mapping loop's do (the library caches a batched call's device tables by what they were built from; a replayed
    batch hits that cache every time, a real stream never does)."""
    import copy
    out = []
    for f in frames:
        g = copy.copy(f)
        T = np.array(f.T, np.float32, copy=True)
        T[0, 3] += np.float32(1e-4 * j)
        g.T = T                                                       # (reassigning a field drops the frame's cached ctypes struct)
        out.append(g)
    return out


N_FRAME_SETS = 4                                                       # rotated in the batched timed loops (the pass buffers rotate through three: never the same pair)


def batched_c4(emap_cls, dev, torch, reps: int = 60):
    """Secondary figure: BASELINE configs[3] -- 32 consecutive sweeps with a variance increment before each, one
    gem_add_batch_device call (the regime in which the path is bandwidth- rather than launch-bound).  The first two batches
    into the fresh map are compared with the committed digests; the timed batches follow on the same map."""
    from gem_amd import synth
    d = golden()
    wl = synth.config_c4(n_sweeps=32)
    cat = torch.from_numpy(np.concatenate(wl.clouds)).to(dev)
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    m = emap_cls(wl.length, wl.resolution, device=dev.index)
    pb = m.pack_batch(wl.frames, off, wl.var_updates)                  # the C-ABI arrays, built once
    m.add_batch(pb, cat)
    ok = sha(m.layer("elevation")) == d["c4_32"]["elevation"] and sha(m.layer("variance")) == d["c4_32"]["variance"]
    m.add_batch(pb, cat)
    ok = ok and sha(m.layer("elevation")) == d["c4_32_twice"]["elevation"] and sha(m.layer("variance")) == d["c4_32_twice"]["variance"]
    # The timed stream: N_FRAME_SETS different frame sets in turn (poses a tenth of a millimetre apart) -- every call builds and
    # uploads its tables, as a mapping loop's calls do.  The replay of ONE batch (tables cached by the library) is reported beside it.
    pbs = [pb] + [m.pack_batch(perturbed_frames(wl.frames, j), off, wl.var_updates) for j in range(1, N_FRAME_SETS)]
    for k in range(8):
        m.add_batch(pbs[k % N_FRAME_SETS], cat)
    m.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        m.add_batch(pbs[k % N_FRAME_SETS], cat)
    m.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        m.add_batch(pb, cat)
    m.synchronize()
    dt_replay = (time.perf_counter() - t0) / reps
    # per-kernel dispatch times of the same batches: as the timed loop runs them (the passes' kernels overlapping on three
    # streams, each stretched by the others), and -- on a second map with the overlap switched off -- every kernel alone on the GPU
    m.set_timing(True); m.stats(reset=True)
    for _ in range(20):
        m.add_batch(pb, cat)
    st_overlapped = m.stats(); m.set_timing(False)
    m.set_counting(True)
    m.add_batch(pb, cat)
    cells = m.stats()["cells_touched"]; records = m.stats()["points_binned"]
    m.close()
    m1 = emap_cls(wl.length, wl.resolution, device=dev.index, debug={"overlap": 0})
    for _ in range(3):
        m1.add_batch(pb, cat)
    m1.set_timing(True); m1.stats(reset=True)
    for _ in range(10):
        m1.add_batch(pb, cat)
    st = m1.stats(); m1.close()
    n = cat.shape[0]
    L2 = wl.length * wl.length
    alg = 16.0 * n + 16.0 * cells + 8.0 * L2 * 32                      # points + touched cells + one dense variance pass per sweep
    must_move = 16.0 * n + 16.0 * float(L2)                            # what one batched call has to move at least: the cloud, the map once
    names = ["k_sort_project", "k_sort_scan", "k_sort_scatter", "k_sort_count", "k_sort_scan(2)", "k_sort_scatter(2)"]
    kern = {}
    if st["launches_walk"]:
        ls = max(st["launches_sort"], 1)
        kern = {nm: 1e3 * v / ls for nm, v in zip(names, st["ms_sort"]) if v > 0}
        # one counting-sort pass = the block-sorted form (k_fuse_block orders a block's records by cell in LDS); two = cell-sorted (k_fuse_walk)
        one_pass = "k_sort_scatter(2)" not in kern
        walk = "k_fuse_block" if one_pass else "k_fuse_walk"
        kern[walk] = 1e3 * st["ms_walk"] / st["launches_walk"]
        # bytes each kernel moves by construction (records of 12 B: {h, var} + key; `records` of the n points are kept)
        moved = {"k_sort_project": 16.0 * n + 12.0 * records, "k_sort_scatter": 24.0 * records, "k_sort_count": 4.0 * records,
                 "k_sort_scatter(2)": 24.0 * records, walk: 12.0 * records + (0.0 if one_pass else 4.0 * records) + 16.0 * L2}
        dom = max((k for k in ("k_sort_project", "k_sort_scatter", "k_sort_scatter(2)", walk) if k in kern), key=lambda k: kern[k])
        pm = pmc_traffic(dom.split("(")[0], "r[0-9][0-9]_c4.json")
        traffic = pm[1]["hbm_bytes_high"] if pm else None
        lo = max(st_overlapped["launches_sort"], 1)
        kern_overlapped = {nm: 1e3 * v / lo for nm, v in zip(names, st_overlapped["ms_sort"]) if v > 0}
        kern_overlapped[walk] = 1e3 * st_overlapped["ms_walk"] / max(st_overlapped["launches_walk"], 1)
        roof = {"bound": "hbm", "kernel": dom, "us_per_launch": kern[dom], "bytes_moved_by_construction": moved[dom],
                "traffic": traffic,
                "traffic_source": (f"profiles/{pm[0]} (tools/profile_one.sh: rocprofv3 --pmc passes over C4 alone, every kernel alone on the GPU): FETCH_SIZE "
                                   f"{pm[1]['FETCH_SIZE']:.0f} KB (doubled per the guide) + WRITE_SIZE {pm[1]['WRITE_SIZE']:.0f} KB per launch, kernel {pm[2]:.1f} us there"
                                   if pm else "no committed PMC summary of C4 found under profiles/"),
                "achieved": moved[dom] / (kern[dom] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": moved[dom] / (kern[dom] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "pipeline_bytes_by_construction": sum(moved[k] for k in kern if k in moved),
                "us_per_kernel_overlapped": kern_overlapped,
                "note": "the kernel that takes longest ALONE on the GPU (second map, overlap off), on the bytes it reads + writes; in the timed loop the "
                        "passes' kernels overlap on three streams and stretch each other (us_per_kernel_overlapped); the pipeline-level figures are "
                        "frac_of_hbm_peak (SURVEY 8d algorithmic bytes / wall) and frac_of_hbm_peak_must_move above"}
    else:
        kern = {"k_bin_wave": 1e3 * st["ms_bin"] / max(st["launches_bin"], 1), "k_fuse_list": 1e3 * st["ms_fuse"] / max(st["launches_fuse"], 1)}
        roof = None
    moved_total = sum(moved[k] for k in kern if k in moved) if st["launches_walk"] else None
    per_kernel = {k: {"us_alone": kern[k], "bytes": moved[k], "frac_of_hbm_peak": moved[k] / (kern[k] * 1e-6) / 1e9 / HBM_PEAK_GBS}
                  for k in kern if st["launches_walk"] and k in moved and kern[k] > 0}
    return {"workload": "C4: 32 consecutive 131072-pt sweeps + Mapvar_update before each, one batched call, 600x600 map",
            "value": n / dt, "unit": "points/s", "us_per_batch": dt * 1e6, "records_kept": int(records), "cells_touched": int(cells),
            "frame_sets_rotated": N_FRAME_SETS, "us_per_batch_one_batch_replayed": dt_replay * 1e6,
            "tables_note": "us_per_batch: consecutive calls carry DIFFERENT frames (N_FRAME_SETS pose sets in turn), so every call builds and uploads its "
                           "device tables; us_per_batch_one_batch_replayed: the same batch again and again, whose tables the library finds cached",
            # the bytes, from the most to the least demanding reading:
            "must_move_bytes": must_move, "frac_of_hbm_peak_must_move": must_move / dt / 1e9 / HBM_PEAK_GBS,
            "bytes_moved_by_construction": moved_total, "frac_moved": (moved_total / dt / 1e9 / HBM_PEAK_GBS) if moved_total else None,
            "algorithmic_bytes": alg, "achieved_GBps": alg / dt / 1e9, "frac_of_hbm_peak": alg / dt / 1e9 / HBM_PEAK_GBS, "frac_of_6300": alg / dt / 1e9 / HBM_ACHIEVABLE_GBS,
            "bytes_note": "must_move = the cloud + the two layers once (what ONE call has to move at least); bytes_moved_by_construction = what the four kernels "
                          "read + write (records of 12 B written once, moved once by the scatter, read once by the walk); algorithmic_bytes = SURVEY 8d's figure, "
                          "which counts 32 dense variance passes (92 MB) that the implementation folds into the walk and never performs",
            "per_kernel": per_kernel, "us_per_kernel": kern, "roofline": roof, "parity_checked": bool(ok)}


def c5_cloud():
    from gem_amd import synth
    wl = synth.config_c5()
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    return wl, np.concatenate(wl.clouds), off


def c5_one_gpu(emap_cls, dev, torch, reps: int = 12, wl=None, cat=None, off=None):
    """BASELINE configs[4] on ONE GPU: the 10^7-point aggregated cloud -> 2400 x 2400 map with one plain gem_add_batch_device
    call per step (what the N > 1 lines scale).  The first pass into the fresh map is compared with the committed digest."""
    if wl is None:
        wl, cat, off = c5_cloud()
    d_cat = torch.from_numpy(cat).to(dev)
    m = emap_cls(wl.length, wl.resolution, device=dev.index)
    pb = m.pack_batch(wl.frames, off, None)
    m.add_batch(pb, d_cat)
    d = golden()["c5_full"]
    ok = sha(m.layer("elevation")) == d["elevation"] and sha(m.layer("variance")) == d["variance"]
    pbs = [pb] + [m.pack_batch(perturbed_frames(wl.frames, j), off, None) for j in range(1, N_FRAME_SETS)]     # (see batched_c4)
    for k in range(3):
        m.add_batch(pbs[k % N_FRAME_SETS], d_cat)
    m.synchronize()
    t0 = time.perf_counter()
    for k in range(reps):
        m.add_batch(pbs[k % N_FRAME_SETS], d_cat)
    m.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        m.add_batch(pb, d_cat)
    m.synchronize()
    dt_replay = (time.perf_counter() - t0) / reps
    m.set_counting(True); m.add_batch(pb, d_cat)
    cells = m.stats()["cells_touched"]; records = m.stats()["points_binned"]
    m.close()
    # every kernel alone on the GPU (second map, the passes' overlap switched off)
    m1 = emap_cls(wl.length, wl.resolution, device=dev.index, debug={"overlap": 0})
    for _ in range(2):
        m1.add_batch(pb, d_cat)
    m1.set_timing(True); m1.stats(reset=True)
    for _ in range(4):
        m1.add_batch(pb, d_cat)
    st = m1.stats(); m1.close()
    del d_cat
    n = int(off[-1])
    alg = 16.0 * n + 16.0 * cells
    names = ["k_sort_project", "k_sort_scan", "k_sort_scatter", "k_sort_count", "k_sort_scan(2)", "k_sort_scatter(2)"]
    ls = max(st["launches_sort"], 1)
    kern = {nm: 1e3 * v / ls for nm, v in zip(names, st["ms_sort"]) if v > 0}
    kern["k_fuse_block"] = 1e3 * st["ms_walk"] / max(st["launches_walk"], 1)
    moved = {"k_sort_project": 16.0 * n + 12.0 * records, "k_sort_scatter": 24.0 * records, "k_sort_count": 4.0 * records,
             "k_sort_scatter(2)": 24.0 * records, "k_fuse_block": 12.0 * records + 16.0 * cells}
    moved_total = sum(moved[k] for k in kern if k in moved)
    must_move = 16.0 * n + 16.0 * cells
    per_kernel = {k: {"us_alone": kern[k], "bytes": moved[k], "frac_of_hbm_peak": moved[k] / (kern[k] * 1e-6) / 1e9 / HBM_PEAK_GBS} for k in kern if k in moved and kern[k] > 0}
    return {"workload": "C5: 10 M-point aggregated cloud (77 sweeps, no variance increments) -> 2400x2400 @ 0.05 m, one gem_add_batch_device call per step, ONE GPU",
            "records_kept": int(records), "must_move_bytes": must_move, "frac_of_hbm_peak_must_move": must_move / dt / 1e9 / HBM_PEAK_GBS,
            "bytes_moved_by_construction": moved_total, "frac_moved": moved_total / dt / 1e9 / HBM_PEAK_GBS, "per_kernel": per_kernel,
            "value": n / dt, "unit": "points/s", "us_per_step": dt * 1e6, "cells_touched": int(cells), "algorithmic_bytes": alg,
            "frame_sets_rotated": N_FRAME_SETS, "us_per_step_one_batch_replayed": dt_replay * 1e6,
            "achieved_GBps": alg / dt / 1e9, "frac_of_hbm_peak": alg / dt / 1e9 / HBM_PEAK_GBS, "frac_of_6300": alg / dt / 1e9 / HBM_ACHIEVABLE_GBS,
            "parity_checked": bool(ok), "parity": "first pass into a fresh map == tests/golden/digests.json c5_full"}


def c3_stream(emap_cls, dev, torch, reps: int = 40):
    """BASELINE configs[2]: 640 x 480 depth image -> 400 x 400 @ 0.025 m, a stream of frames.  The cells under the camera hold
    hundreds of points each and G_fuse's recurrence is sequential per cell (GPU:477-537), so a frame cannot take less than its
    longest chain x the time of one step: `longest_chain` and `ns_per_chain_step` let a reader check that claim."""
    from gem_amd import synth
    d = golden()["c3"]
    wl = synth.config_c3()
    dc = torch.from_numpy(wl.clouds[0]).to(dev)
    m = emap_cls(wl.length, wl.resolution, device=dev.index)
    m.move(wl.map_position)
    m.add(wl.frames[0], dc)
    ok = sha(m.layer("elevation")) == d["elevation"] and sha(m.layer("variance")) == d["variance"]
    c = wl.clouds[0]
    pp = m.process_points(wl.frames[0], c[:, 0].copy(), c[:, 1].copy(), c[:, 2].copy())
    idx = pp["index"]; idx = idx[idx >= 0]
    per_cell = np.bincount(idx)
    longest = int(per_cell.max())
    for _ in range(5):
        m.add(wl.frames[0], dc)
    m.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        m.add(wl.frames[0], dc)
    m.synchronize()
    dt = (time.perf_counter() - t0) / reps
    m.set_timing(True); m.stats(reset=True)
    for _ in range(reps):
        m.add(wl.frames[0], dc)
    st = m.stats(); m.set_timing(False)
    us_walk = 1e3 * st["ms_walk"] / max(st["launches_walk"], 1)
    m.close()
    n = c.shape[0]
    cells = int((per_cell > 0).sum())
    alg = 16.0 * n + 16.0 * cells
    return {"workload": "C3: 640x480 depth image (307200 pts) -> 400x400 @ 0.025 m, one gem_add_device per frame, laser variance model",
            "value": n / dt, "unit": "points/s", "us_per_frame": dt * 1e6, "cells_touched": cells, "mean_points_per_touched_cell": float(idx.size) / max(cells, 1),
            "longest_chain": longest, "us_fuse_kernel": us_walk, "ns_per_chain_step": 1e3 * us_walk / max(longest, 1),
            "algorithmic_bytes": alg, "achieved_GBps": alg / dt / 1e9, "frac_of_hbm_peak": alg / dt / 1e9 / HBM_PEAK_GBS,
            "parity_checked": bool(ok), "parity": "first frame into a fresh map == tests/golden/digests.json c3"}


def c2_variants(emap_cls, dev, torch, reps: int = 400):
    """SURVEY 8d's other C2 figures: the sweep with the reference's sensor-frame reject filter ON (gpu_process.cu:393), and the
    end-to-end rate when the boundary hands over HOST buffers (gem_add: pinned staging + H2D inside the call; never `value`)."""
    from gem_amd import synth
    d = golden()
    out = {}
    wl = synth.config_c2(reference_filter=True)
    dc = torch.from_numpy(wl.clouds[0]).to(dev)
    m = emap_cls(wl.length, wl.resolution, device=dev.index)
    m.add(wl.frames[0], dc)
    ok = sha(m.layer("elevation")) == d["c2_filter"]["elevation"] and sha(m.layer("variance")) == d["c2_filter"]["variance"]
    for _ in range(20):
        m.add(wl.frames[0], dc)
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        m.add(wl.frames[0], dc)
    m.synchronize(); dt = (time.perf_counter() - t0) / reps
    n = wl.clouds[0].shape[0]
    alg = 16.0 * n + 16.0 * d["c2_filter"]["cells_touched"]
    out["c2_reference_filter_on"] = {"workload": "C2 sweep, reject filter of gpu_process.cu:393 ON (55879 of 131072 points accepted)", "value": n / dt, "unit": "points/s",
                                     "us_per_step": dt * 1e6, "achieved_GBps": alg / dt / 1e9, "parity_checked": bool(ok),
                                     "parity": "first sweep into a fresh map == tests/golden/digests.json c2_filter"}
    m.close()
    wl = synth.config_c2()
    m = emap_cls(wl.length, wl.resolution, device=dev.index)
    host = wl.clouds[0]
    m.add(wl.frames[0], host)
    ok = sha(m.layer("elevation")) == d["c2"]["elevation"] and sha(m.layer("variance")) == d["c2"]["variance"]
    for _ in range(10):
        m.add(wl.frames[0], host)
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(reps // 4):
        m.add(wl.frames[0], host)
    m.synchronize(); dt = (time.perf_counter() - t0) / (reps // 4)
    out["e2e_with_h2d"] = {"workload": "C2 sweep handed over as a HOST array, call after call (gem_add: a copy into the pinned staging buffer, which the kernels read over the link)", "value": n / dt, "unit": "points/s",
                           "us_per_step": dt * 1e6, "parity_checked": bool(ok), "parity": "first sweep into a fresh map == tests/golden/digests.json c2"}
    m.close()
    # BASELINE configs[3] handed over as 32 HOST arrays (gem_add_batch): the link carries 67 MB per batch -- its rate, not
    # the kernels', sets this figure; never `value`
    wl4 = synth.config_c4(n_sweeps=32)
    m = emap_cls(wl4.length, wl4.resolution, device=dev.index)
    pb = m.pack_batch(wl4.frames, np.concatenate([[0], np.cumsum([c.shape[0] for c in wl4.clouds])]), wl4.var_updates)
    m.add_batch_host(pb, wl4.clouds)
    ok = sha(m.layer("elevation")) == d["c4_32"]["elevation"] and sha(m.layer("variance")) == d["c4_32"]["variance"]
    for _ in range(2):
        m.add_batch_host(pb, wl4.clouds)
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        m.add_batch_host(pb, wl4.clouds)
    m.synchronize(); dt = (time.perf_counter() - t0) / 8
    n4 = sum(c.shape[0] for c in wl4.clouds)
    out["c4_host_batch"] = {"workload": "C4 batch handed over as 32 HOST arrays (gem_add_batch: staging copies + DMA of sweep k + 1 beside sweep k, one batched pass)",
                            "value": n4 / dt, "unit": "points/s", "us_per_batch": dt * 1e6, "bytes_over_pcie_per_batch": 16.0 * n4,
                            "link_GBps": 16.0 * n4 / dt / 1e9, "parity_checked": bool(ok), "parity": "first batch into a fresh map == tests/golden/digests.json c4_32"}
    m.close()
    out["node_host_arrays"] = node_host_arrays(emap_cls, dev)
    return out


def node_host_arrays(emap_cls, dev, reps: int = 40, copy_threads=None, debug=None):
    """The path the UNMODIFIED node drives, with its caller-owned host arrays (never `value`): per frame Mapvar_update, Process_points
    (3 arrays up, 5 down: gpu_process.cu:1096-1141), Fuse (7 arrays up: :1165-1192), Map_feature (nine L x L layers down: :1283-1291)
    and Raytracing, through the C ABI entry points the nine-symbol adapter calls (include/gem/gem_compat_eigen.hpp), arrays
    allocated once like the node's.  PCIe and the runtime's staging of pageable memory set this rate."""
    import ctypes as C
    from gem_amd import synth, _lib
    wl = synth.config_c2(reference_filter=True)
    c = wl.clouds[0]; f = wl.frames[0]; n = c.shape[0]; L = wl.length
    x, y, z = (np.ascontiguousarray(c[:, k]) for k in range(3))
    idx = np.empty(n, np.int32); var, xt, yt, zt = (np.empty(n, np.float32) for _ in range(4))
    col = [np.full(n, 120, np.int32) for _ in range(3)]; inten = np.full(n, 7.0, np.float32)
    layers_f = [np.empty(L * L, np.float32) for _ in range(6)]; layers_i = [np.empty(L * L, np.int32) for _ in range(3)]
    m = emap_cls(L, wl.resolution, device=dev.index)
    m.set_lowest_tracking(True)
    if copy_threads is not None:
        m.debug_set("copy_threads", int(copy_threads))
    for k, v in (debug or {}).items():
        m.debug_set(k, int(v))
    m.reserve(n, 1, True)                               # (the node's maximum cloud: every arena and the pinned staging sized up front)
    lib, h, P = m._lib, m._h, f.to_struct()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    t = {"mapvar_update": [], "process_points": [], "fuse": [], "map_feature": [], "raytracing": []}
    for r in range(reps + 4):
        t0 = time.perf_counter(); lib.gem_mapvar_update(h, 1e-6)
        t1 = time.perf_counter(); rc1 = lib.gem_process_points(h, C.byref(P), n, vp(x), vp(y), vp(z), None, 0, vp(idx), vp(var), vp(xt), vp(yt), vp(zt))
        t2 = time.perf_counter(); rc2 = lib.gem_fuse(h, n, vp(idx), vp(col[0]), vp(col[1]), vp(col[2]), vp(inten), vp(zt), vp(var))
        t3 = time.perf_counter(); rc3 = lib.gem_map_feature(h, vp(layers_f[0]), vp(layers_f[1]), vp(layers_i[0]), vp(layers_i[1]), vp(layers_i[2]),
                                                            vp(layers_f[2]), vp(layers_f[3]), vp(layers_f[4]), vp(layers_f[5]))
        t4 = time.perf_counter(); rc4 = lib.gem_raytracing(h)
        t5 = time.perf_counter()
        assert rc1 == 0 and rc2 == 0 and rc3 == 0 and rc4 == 0
        if r >= 4:
            for k, v in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                t[k].append(v * 1e6)
    m_threads = m.debug_get("copy_threads")
    xfer = {k: m.debug_get(f"xfer_{k}_ns") / (reps + 4) / 1e3 for k in ("upload_memcpy", "upload_enqueue", "download_enqueue", "download_wait", "download_memcpy")}
    m.synchronize(); m.close()
    us = {k: float(np.median(v)) for k, v in t.items()}
    total = sum(us.values())
    # which socket the calling thread ran on, against the GPUs' (this frame costs ~10 % more from the far socket: tools/dbg/numa_ab.sh)
    numa = {"caller_node": None, "gpu_nodes": []}
    try:
        import ctypes, glob
        cpu = ctypes.CDLL(None).sched_getcpu()
        for nd in glob.glob("/sys/devices/system/node/node[0-9]*"):
            lst = open(nd + "/cpulist").read().strip()
            for part in lst.split(","):
                a, _, b = part.partition("-")
                if int(a) <= cpu <= int(b or a):
                    numa["caller_node"] = int(nd.rsplit("node", 1)[1])
        numa["gpu_nodes"] = sorted({int(open(f).read()) for f in glob.glob("/sys/class/drm/card*/device/numa_node") if open(f).read().strip() not in ("", "-1")})
    except Exception:
        pass
    return {"workload": "the node's frame with caller-owned HOST arrays: Mapvar_update + Process_points + Fuse (colours) + Map_feature (nine layers to the host) + "
                        "Raytracing, C2 sweep with the reference's reject filter, 600 x 600 map",
            "copy_threads": m_threads, "numa": numa, "us_per_frame": total, "us_per_call": us, "host_us_per_frame_in_transfers": xfer, "bytes_over_pcie_per_frame": 12.0 * n + 20.0 * n + 28.0 * n + 36.0 * L * L,
            "value": n / (total * 1e-6), "unit": "points/s"}


# ---- C5 through the sharded entry points: N > 1 (strong scaling), or N = 1 with --workload c5 -----------------------------------
def run_c5_distributed(args, torch, dist, world, rank, local_rank, dev):
    """BASELINE configs[4] on `world` ranks: gem_add_sharded_device + gem_allgather_layers per step.  With one rank (--workload c5
    --gpus 1) the same entry points run without an exchange: the like-for-like N = 1 point of the scaling curve."""
    from gem_amd import ElevationMap, synth
    from gem_amd.tiling import first_point_in_sweep, shard_batch, tile_strip_rows
    off = synth.c5_offsets()
    n_total = int(off[-1])
    first, local = shard_batch(off, world, rank)
    fp = first_point_in_sweep(off, first, local)
    # only this rank's share of the cloud is generated and made resident (every sweep has its own seed)
    mine = list(range(first, first + len(local) - 1))
    wl = synth.config_c5(sweeps=mine)
    share = np.concatenate([wl.clouds[k] for k in mine])[local[0] - int(off[first]):][: local[-1] - local[0]] if mine else np.zeros((0, 4), np.float32)
    d_share = torch.from_numpy(np.ascontiguousarray(share)).to(dev)
    n_sweeps = len(wl.frames)

    def bcast_uid(n):
        uid = [ElevationMap.comm_unique_id() if rank == 0 else None for _ in range(n)]
        if dist is not None:
            dist.broadcast_object_list(uid, src=0)
        return uid

    def barrier(*maps):
        if dist is not None:
            dist.barrier()
        for m in maps:
            m.synchronize()
        torch.cuda.synchronize()

    def all_max(v):
        if dist is None:
            return float(v)
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    uid = bcast_uid(2)                                       # one communicator for the timed map, one for the checked map
    emap = ElevationMap(wl.length, wl.resolution, device=local_rank)
    emap.comm_init_tiles(uid[0], world, rank)
    emap.reserve(n_total, n_sweeps)                          # the step's arenas, both sets of receive buffers: nothing allocates inside the timed loop
    pb = emap.pack_batch([wl.frames[first + i] for i in range(len(local) - 1)], [v - local[0] for v in local], None)

    def step(gather=True):
        emap.add_sharded(pb, d_share, first, n_sweeps, None, fp)
        if gather:
            emap.allgather_layers(False)

    def timed(k, gather=True):
        barrier(emap)
        t0 = time.perf_counter()
        for _ in range(k):
            step(gather)
        barrier(emap)
        return all_max(time.perf_counter() - t0)

    for _ in range(args.warmup):
        step()
    elapsed = timed(args.steps)
    # the same steps without the all-gather of the layers (what the exchange + walk pipeline does on its own)
    elapsed_nogather = timed(args.steps, gather=False)
    # what every rank holds after the all-gather is the whole map: check it against the committed digest of ONE pass
    chk = ElevationMap(wl.length, wl.resolution, device=local_rank)
    chk.comm_init_tiles(uid[1], world, rank)
    chk.add_sharded(pb, d_share, first, n_sweeps, None, fp)
    chk.allgather_layers(False)
    d = golden()["c5_full"]
    ok = sha(chk.layer("elevation")) == d["elevation"] and sha(chk.layer("variance")) == d["variance"]
    if dist is not None:
        okt = torch.tensor([1 if ok else 0], device=dev); dist.all_reduce(okt, op=dist.ReduceOp.MIN); ok = bool(okt.item() == 1)
    # per-phase device times of this rank: kernel dispatch stamps (sort, walk) and, on W > 1, events on the streams the phases run on
    emap.set_timing(True); emap.stats(reset=True)
    n_ph = max(args.steps // 4, 2)
    for _ in range(n_ph):
        step()
    st = emap.stats()                                        # (synchronises: the last step's second half included)
    phases = {"sort_kernels_sum": 1e3 * sum(st["ms_sort"]) / max(st["launches_walk"], 1), "k_fuse_block": 1e3 * st["ms_walk"] / max(st["launches_walk"], 1)}
    if world > 1:
        for key in ("exchange", "exchange_to_walk", "walk", "publish", "gather"):
            ns = emap.debug_get(f"step_{key}_ns")
            phases[key] = None if ns <= -(1 << 60) else ns / 1e3
    emap.set_timing(False)
    links = xgmi_summary(world, {k: emap.debug_get(k) for k in ("gather_bytes_in", "gather_bytes_out", "step_exchange_bytes_in", "step_exchange_bytes_out")},
                         phases) if world > 1 else None
    us_step = 1e6 * elapsed / args.steps
    us_step_nogather = 1e6 * elapsed_nogather / args.steps
    rows = tile_strip_rows(wl.length, world)
    barrier(emap, chk)
    emap.close(); chk.close()
    del d_share
    # the SAME workload through the plain one-GPU call, by rank 0 in this process while the other ranks wait: what the speed-up is measured against
    one = cpu = None
    if rank == 0:
        wl_full, cat, off_full = c5_cloud()
        one = c5_one_gpu(ElevationMap, dev, torch, reps=max(6, min(args.steps, 20)), wl=wl_full, cat=cat, off=off_full)
        if not args.no_cpu_baseline:
            cpu = cpu_baseline_c5(wl_full, cat, off_full, args.cpu_seconds)
    if dist is not None:
        dist.barrier()
    out = None
    if rank == 0:
        alg = 16.0 * n_total + 16.0 * one["cells_touched"]
        gathered_bytes = 8.0 * wl.length * wl.length * (world - 1) / world          # elevation + variance of the other ranks' strips, received per rank and step
        out = {
            "metric": "fused points/sec into 2400x2400 grid (BASELINE configs[4]); achieved HBM GB/s vs roofline",
            "value": n_total * args.steps / elapsed, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C5 (BASELINE configs[4]): single 10 M-point aggregated cloud (77 sweeps) -> 2400x2400 @ 0.05 m grid per step, "
                                   "points sharded by index range over the ranks, sorted records routed to tile-row strip owners (RCCL send/recv), "
                                   "RCCL all-gather of the fused elevation + variance layers" + (" -- ONE rank: no exchange, no all-gather" if world == 1 else ""),
                       "points_per_step": n_total, "grid": "2400x2400@0.05m", "parallelism": f"shard{world}+strips{world}", "strip_rows": rows},
            "one_gpu_us_per_step": one["us_per_step"], "speedup_vs_one_gpu": one["us_per_step"] / us_step, "c5_one_gpu": one,
            "value_without_allgather": n_total * args.steps / elapsed_nogather, "us_per_step_without_allgather": us_step_nogather,
            "xgmi": links,
            "allgather_us": phases.get("gather"), "allgather_bytes_received_per_rank": gathered_bytes if world > 1 else 0.0,
            "allgather_GBps_per_rank": (gathered_bytes / (phases["gather"] * 1e-6) / 1e9) if world > 1 and phases.get("gather") else None,
            "phases_us_rank0": dict(phases, step=us_step,
                                    note="device time of each phase of ONE step on the stream it runs on (sort on the binning streams, exchange on the "
                                         "communication stream, walk + publish on the handle's stream, gather on the gather stream); consecutive steps "
                                         "overlap them, so the step's period is about the longest phase, not their sum (DESIGN.md section 7)"),
            "roofline": {"bound": "hbm", "kernel": "pipeline (per rank)", "achieved": alg / world / (us_step * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": alg / world / (us_step * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "note": "SURVEY 8d algorithmic bytes of the whole cloud / ranks / step time; the all-gather adds 46 MB x (W - 1) / W received per rank and step"},
            "parity_checked": bool(ok and one["parity_checked"]),
            "parity": "every rank's all-gathered map after one pass == tests/golden/digests.json c5_full (elevation, variance); so is the one-GPU map",
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
    return out


def xgmi_summary(world: int, counts: dict, phases: dict):
    """What rank 0's links carried in ONE step (the library's own byte counts, gem_debug_get), priced against xGMI: 7 point-to-point
    links per GPU, ~153 GB/s each way at peak (/opt/skills/guides/MI355X_MICROARCH.md); the direct all-gather and the exchange use every
    link at once.  Pure arithmetic (tests/test_bench_contract.py runs it without a GPU)."""
    XGMI_LINK_GBPS = 153.0
    peers = max(world - 1, 1)
    g_in, g_out = int(counts.get("gather_bytes_in", 0)), int(counts.get("gather_bytes_out", 0))
    x_in, x_out = int(counts.get("step_exchange_bytes_in", 0)), int(counts.get("step_exchange_bytes_out", 0))
    per_link_gather = max(g_in, g_out) / peers                # every peer's strip arrives over its own link while ours leaves over the same one
    per_link_exchange = max(x_in, x_out) / peers
    us = lambda b, frac=1.0: b / (frac * XGMI_LINK_GBPS * 1e3)
    return {"rccl_ranks": world, "peers_per_rank": world - 1, "xgmi_link_peak_GBps_each_way": XGMI_LINK_GBPS,
            "allgather_bytes_in": g_in, "allgather_bytes_out": g_out, "allgather_bytes_per_link": per_link_gather,
            "allgather_us_predicted_at_link_peak": us(per_link_gather), "allgather_us_predicted_at_75pct": us(per_link_gather, 0.75),
            "allgather_us_measured": phases.get("gather"),
            "exchange_bytes_in": x_in, "exchange_bytes_out": x_out, "exchange_bytes_per_link_mean": per_link_exchange,
            "exchange_us_predicted_at_link_peak": us(per_link_exchange), "exchange_us_measured": phases.get("exchange"),
            "note": "rank 0's byte counts of ONE step; a measured time far above the prediction at 75 % of the link peak means the collective "
                    "is not using the links in parallel (or the ranks do not start it together: compare `phases_us_rank0`)"}


def cpu_baseline_c5(wl, cat, off, budget_s: float):
    """The all-core CPU oracle on the C5 cloud (one pass = 10^7 points; rank 0 only), checked against the committed digest."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle
    ncpu = os.cpu_count() or 1
    t_begin = time.perf_counter()
    best = None
    for nt in sorted({min(ncpu, v) for v in (16, 32, 64)}):
        m = oracle.OracleMap(wl.length, wl.resolution)
        t0 = time.perf_counter(); m.add_batch_mt(wl.frames, cat, off, None, nt); dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (nt, dt, m)
        if time.perf_counter() - t_begin > budget_s:
            break
    nt, dt, m = best
    d = golden()["c5_full"]
    ok = sha(m.layer("elevation")) == d["elevation"] and sha(m.layer("variance")) == d["variance"]
    return {"value": int(off[-1]) / dt, "unit": "points/s", "kind": "port", "cores": nt, "host_cores": ncpu, "cpu_model": cpu_model(),
            "matches_committed_digest": bool(ok),
            "sample": f"one pass of the whole C5 cloud (10^7 pts -> 2400x2400) by oracle/gem_oracle_mt.c gemo_add_batch_mt on {nt} threads ({dt:.2f} s), "
                      f"the map compared with tests/golden/digests.json c5_full"}


def self_launch(args) -> None:
    """`python bench.py --gpus N` (N > 1) without a launcher around it: run the N ranks ourselves."""
    import socket
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible on this box; the N > 1 bench needs one device per rank "
                         f"(set GEM_BENCH_FORCE_DIST=1 with --gpus 1 to exercise the distributed code path on one device)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    print("[bench] re-executing under torch.distributed.run: " + " ".join(cmd), file=sys.stderr, flush=True)
    os.execv(sys.executable, cmd)


_REAL_STDOUT = None


def emit(obj) -> None:
    """The ONE JSON line, on the process's real stdout."""
    line = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)                                    # does not return
    # Libraries underneath write to stdout on their own (RCCL prints a version banner at communicator creation, the compiled
    # reference printf's from its Init): everything but the JSON line goes to stderr.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from gem_amd import ElevationMap, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    workload = args.workload or ("c5" if world > 1 else "c2")
    if workload == "c2" and world > 1:
        raise SystemExit("--workload c2 is a single-GPU configuration (BASELINE configs[1]); N > 1 runs C5 (configs[4])")
    distributed = workload == "c5" or bool(os.environ.get("GEM_BENCH_FORCE_DIST"))      # (the env var: the N > 1 code path with one rank, kept for older scripts)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        use_dist = world > 1
        if use_dist:
            if "MASTER_ADDR" not in os.environ:
                os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ.setdefault("MASTER_PORT", "29655")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        out = run_c5_distributed(args, torch, dist if use_dist else None, world, rank, local_rank, dev)
        if rank == 0:
            emit(out)
            if not out["parity_checked"]:
                if use_dist:
                    dist.destroy_process_group()
                raise SystemExit("parity check FAILED: the tiled map differs from the committed digest")
        if use_dist:
            dist.destroy_process_group()
        return

    wl = synth.config_c4(n_sweeps=N_DISTINCT, seed0=100)
    n_per = wl.clouds[0].shape[0]
    d_clouds = [torch.from_numpy(c).to(dev) for c in wl.clouds]
    emap = ElevationMap(wl.length, wl.resolution, device=local_rank)
    done = [0]                                               # sweeps the map has fused

    def step():
        k = done[0] % N_DISTINCT
        emap.add(wl.frames[k], d_clouds[k]); done[0] += 1

    def sync():
        emap.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    # EXACTLY K steps between synchronisations, repeated until >= 50 ms have been timed; the median repetition counts
    rep_s, total = [], 0.0
    while total < MIN_TIMED_S and len(rep_s) < 2000:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        rep_s.append(time.perf_counter() - t0); total += rep_s[-1]
    elapsed = float(np.median(rep_s))
    n_timed_map = done[0]
    timed_layers = (emap.layer("elevation"), emap.layer("variance"))
    value = args.steps * n_per / elapsed

    # ---- roofline of the dominant kernel: dispatch time stamps of every kernel launch (recorded on the handle's stream by
    #      libgem_hip), same workload, second timed loop ------------------------------------------------------------------
    emap.set_timing(True)
    emap.stats(reset=True)
    for _ in range(max(args.steps, 200)):
        step()
    st = emap.stats()
    emap.set_timing(False)
    emap.set_counting(True)                                  # distinct touched cells per sweep (C_touched of SURVEY 8d)
    cells = []
    for _ in range(N_DISTINCT):
        step()
        cells.append(emap.stats()["cells_touched"])
    emap.set_counting(False)
    cells = float(np.mean(cells))
    us_bin = 1e3 * st["ms_bin"] / max(st["launches_bin"], 1)
    us_fuse = 1e3 * st["ms_fuse"] / max(st["launches_fuse"], 1)
    us_frame = 1e3 * st["ms_frame"] / max(st["launches_frame"], 1)
    # algorithmic bytes per launch (B_alg = 16 N + 16 C_touched)
    alg_bin, alg_fuse = 16.0 * n_per, 16.0 * cells
    if st["launches_frame"] > st["launches_fuse"]:
        # steady state of a stream of single sweeps: ONE launch per step, k_frame = fuse of the previous sweep's records +
        # binning of the new cloud, so its algorithmic bytes are the whole frame's
        dom, dom_us, dom_bytes = "k_frame", us_frame, alg_bin + alg_fuse
    elif us_fuse >= us_bin:
        dom, dom_us, dom_bytes = "k_fuse_list", us_fuse, alg_fuse
    else:
        dom, dom_us, dom_bytes = "k_bin_wave", us_bin, alg_bin
    achieved = dom_bytes / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
    traffic, traffic_note = None, "no committed PMC summary of this command found under profiles/"
    pm = pmc_traffic(dom)
    if pm:
        traffic = pm[1]["hbm_bytes_high"]
        traffic_note = (f"profiles/{pm[0]} (rocprofv3 --pmc passes of this command, tools/profile_c2.sh): FETCH_SIZE {pm[1]['FETCH_SIZE']:.0f} KB "
                        f"(doubled per the guide) + WRITE_SIZE {pm[1]['WRITE_SIZE']:.0f} KB per launch; uncorrected {pm[1]['hbm_bytes_low']:.0f} B")
        if pm[2]:
            # the counters were collected on a committed build: if the kernel has changed since, its duration gives it away
            stale = abs(pm[2] - dom_us) > 0.25 * dom_us
            traffic_note += (f"; kernel duration in that run {pm[2]:.2f} us vs {dom_us:.2f} us now" +
                             (" -- MORE THAN 25 % APART: the counters may describe an older kernel, re-run tools/profile_c2.sh" if stale else ""))
    # (the fuse launch of a stream of sweeps is k_frame WITHOUT a binning half: the last sweep's list, at the synchronisation)
    fl = "k_frame_fuse_half_only" if st["launches_frame"] > st["launches_fuse"] else "k_fuse_list"
    roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_of_6300": achieved / HBM_ACHIEVABLE_GBS, "traffic": traffic, "traffic_source": traffic_note,
                "us_per_launch": {"k_frame": us_frame, "k_bin_wave": us_bin, fl: us_fuse},
                "launches": {"k_frame": st["launches_frame"], "k_bin_wave": st["launches_bin"], fl: st["launches_fuse"]},
                "algorithmic_bytes_per_launch": {"k_frame": alg_bin + alg_fuse, "k_bin_wave": alg_bin, fl: alg_fuse},
                "note": "one sweep is ~3 MB of algorithmic traffic (0.5 us at the HBM rate): the frame is launch / latency bound, see "
                        "DESIGN.md section 6; `batched_c4` is the bandwidth-regime figure"}

    ok_stream, how = check_c2_stream(ElevationMap, dev, torch)
    out = {
        "metric": "fused points/sec into 600x600 grid; achieved HBM GB/s vs roofline",
        "value": value, "unit": "points/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "timed_repetitions": len(rep_s), "timed_total_ms": 1e3 * total,
        "ms_per_step_min_max": [1e3 * min(rep_s) / args.steps, 1e3 * max(rep_s) / args.steps],
        "config": {"workload": "C2: single 64-beam LiDAR sweep (64x2048 = 131072 XYZI pts) -> 600x600 @ 0.05 m grid, "
                               "per step; 8 distinct seeded sweeps cycled; reject filter off",
                   "points_per_step": n_per, "grid": "600x600@0.05m", "parallelism": "single",
                   "cells_touched_per_sweep": cells},
        "roofline": roofline,
        "parity_checked": bool(ok_stream), "parity": how,
    }
    emap.close()
    failed = not ok_stream
    if not args.no_extras:
        out["batched_c4"] = batched_c4(ElevationMap, dev, torch)
        out["c5_one_gpu"] = c5_one_gpu(ElevationMap, dev, torch)
        out["c3"] = c3_stream(ElevationMap, dev, torch)
        out.update(c2_variants(ElevationMap, dev, torch))
        extras = ("batched_c4", "c5_one_gpu", "c3", "c2_reference_filter_on", "e2e_with_h2d", "c4_host_batch")       # (node_host_arrays carries no digest: the same entry points are parity-tested in tests/)
        out["parity_checked"] = bool(out["parity_checked"] and all(out[k]["parity_checked"] for k in extras))
        out["parity"] += "; C4 batch (twice into a fresh map) vs c4_32 / c4_32_twice; C5 on one GPU vs c5_full; C3 vs c3; C2 with the reference filter vs c2_filter; host-array C2 vs c2"
        failed = failed or not out["parity_checked"]
    if not args.no_cpu_baseline:
        allc, one, lit = cpu_baseline(wl, n_timed_map, timed_layers, args.cpu_seconds)
        out["cpu_baseline"] = allc
        out["cpu_baseline_1t"] = one
        out["cpu_reference_literal"] = lit
        if allc.get("replay_matches_timed_map") is False:
            failed = True
    emit(out)
    if failed:
        raise SystemExit("parity check FAILED: see parity_checked / cpu_baseline.replay_matches_timed_map in the line above")


if __name__ == "__main__":
    main()
