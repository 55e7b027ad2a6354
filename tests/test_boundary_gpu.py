"""The boundary's behaviour under the reference's calling pattern:
  * the thread pair {Process_points -> Fuse} || {Mapvar_update} (EMg.cpp:391-394, 277-282, 292-299): the handle serialises the
    calls, so the map must equal SOME serial interleaving of them, replayed on the oracle;
  * device inputs produced on another stream: gem_wait_event orders the handle's work behind the producer without a host sync."""
import itertools
import threading

import numpy as np
import pytest

from gem_amd import ElevationMap, SensorModel, synth

pytestmark = pytest.mark.gpu
F32 = np.float32


@pytest.mark.parametrize("trial", range(4))
def test_two_threads_give_a_serial_interleaving(oracle_mod, trial):
    L, res = 96, 0.1
    frame = synth._frame_for(synth.pose_matrix(0.1, -0.2, 0.0, yaw=0.3), SensorModel.velodyne())
    clouds = [synth.random_cloud(50 + 10 * trial + k, 20_000, 4.0, z_sigma=0.1, dup_fraction=0.5) for k in range(3)]
    updates = [3e-4, 7e-4]                                   # big enough to move every fused variance visibly
    gpu = ElevationMap(L, res)
    gpu.add(frame, clouds[0])                                # a populated map: the increments act on every cell
    start = threading.Barrier(2)

    def thread_a():                                          # the sensor thread: Process_points -> Fuse, three frames
        start.wait()
        for c in clouds:
            pp = gpu.process_points(frame, c[:, 0], c[:, 1], c[:, 2])
            gpu.fuse(pp["index"], pp["height"], pp["var"])

    def thread_b():                                          # the motion thread: Mapvar_update, twice
        start.wait()
        for u in updates:
            gpu.mapvar_update(u)

    ta, tb = threading.Thread(target=thread_a), threading.Thread(target=thread_b)
    ta.start(); tb.start(); ta.join(); tb.join()
    ge, gv = gpu.layer("elevation"), gpu.layer("variance")
    # every order of 3 fuses (F) and 2 updates (U) that keeps each thread's own order: C(5, 2) = 10 candidates
    matches = []
    for where in itertools.combinations(range(5), 2):
        ref = oracle_mod.OracleMap(L, res)
        ref.add(frame, clouds[0])
        fi, ui = 0, 0
        for slot in range(5):
            if slot in where:
                ref.mapvar_update(updates[ui]); ui += 1
            else:
                c = clouds[fi]; fi += 1
                pp = ref.process_points(frame, c[:, 0], c[:, 1], c[:, 2])
                ref.fuse(pp["index"], pp["height"], pp["var"])
        if np.array_equal(ge, ref.layer("elevation")) and np.array_equal(gv, ref.layer("variance")):
            matches.append(where)
    assert len(matches) >= 1, "the map equals no serial interleaving of the two threads' calls"


def test_wait_event_orders_the_handle_behind_a_producer_stream(oracle_mod):
    import torch
    wl = synth.config_c4(n_sweeps=2)
    gpu, ref = ElevationMap(wl.length, wl.resolution), oracle_mod.OracleMap(wl.length, wl.resolution)
    producer = torch.cuda.Stream()
    staging = [torch.from_numpy(c).pin_memory() for c in wl.clouds]
    big = torch.randn(4096, 4096, device="cuda")
    for k in range(2):
        d = torch.zeros_like(staging[k], device="cuda")      # zeros: a handle that runs ahead of the producer fuses nothing
        with torch.cuda.stream(producer):
            for _ in range(4):                               # something slow in front of the copy
                big = big @ big * 1e-4
            d.copy_(staging[k], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(producer)
        gpu.wait_event(ev)
        gpu.add(wl.frames[k], d)
        ref.add(wl.frames[k], wl.clouds[k])
    assert np.array_equal(gpu.layer("elevation"), ref.layer("elevation")) and np.array_equal(gpu.layer("variance"), ref.layer("variance"))
    assert (ref.layer("elevation") != -10).sum() > 50_000
    # a big batch (its binning runs on the handle's second stream) behind a producer
    wl = synth.config_c4(n_sweeps=10)
    off = np.concatenate([[0], np.cumsum([c.shape[0] for c in wl.clouds])])
    host = torch.from_numpy(np.concatenate(wl.clouds)).pin_memory()
    gpu, ref = ElevationMap(wl.length, wl.resolution), oracle_mod.OracleMap(wl.length, wl.resolution)
    for rep in range(2):
        d = torch.zeros_like(host, device="cuda")
        with torch.cuda.stream(producer):
            for _ in range(4):
                big = big @ big * 1e-4
            d.copy_(host, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(producer)
        gpu.wait_event(ev)
        gpu.add_batch(wl.frames, d, off, wl.var_updates)
        for k in range(10):
            ref.mapvar_update(wl.var_updates[k]); ref.add(wl.frames[k], wl.clouds[k])
        assert np.array_equal(gpu.layer("variance"), ref.layer("variance")) and np.array_equal(gpu.layer("elevation"), ref.layer("elevation")), rep
