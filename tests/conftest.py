"""pytest configuration: `gpu` marker + shared fixtures.

CPU tests (-m "not gpu") cover the oracle against its known-answer tests and golden fixtures, the
host logic, and that libgem_hip.so loads and exports every symbol of include/gem_hip.h.
GPU tests (-m gpu) are the parity tests proper: HIP path vs oracle through the C ABI.
"""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    config.addinivalue_line("markers", "one_pipeline: a GPU test that does not depend on the pipeline knobs (child processes build their own maps, "
                                       "or no map is fused at all): it runs once, not once per entry of PIPELINES")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no HIP device in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle as om
    om.build()
    return om


@pytest.fixture(scope="session")
def ref_mod():
    """oracle/_ref/libgem_ref.so: the reference's own gpu_process.cu compiled for the CPU (oracle/ref_build)."""
    import ref
    if ref.lib() is None:
        pytest.skip("no /root/reference to build from and no prebuilt oracle/_ref/libgem_ref.so")
    return ref


PIPELINES = {
    "default": {},                                                                   # the library's own choice per pass
    "cell": {"sort_min_points": 1, "sort_form": 1},                                  # cell-sorted (k_fuse_walk) for every pass, however small
    "cell3": {"sort_min_points": 1, "sort_form": 1, "sort_passes": 3, "sort_chunk": 4096, "fuse_count": 0},   # ... with three counting-sort passes (maps beyond 2^20 cells), the chunks of 4096 records big passes take (small ones take 1024) and every pass counting for itself (small ones let the first scatter count for the second)
    "block": {"sort_min_points": 1, "sort_form": 2},                                 # block-sorted (k_fuse_block) for every pass
    "block2": {"sort_min_points": 1, "sort_form": 2, "sort_passes": 2, "blk_batch": 2048, "sort_chunk": 4096, "fuse_count": 2},   # ... with two passes (a block's records are found by search) and the rounds of 2048 records that heavy blocks take (small passes take 512)
    "cell_ballot": {"sort_min_points": 1, "sort_form": 1, "rank_by_ballot": 1},     # ... ranking by ballot instead of through the LDS: the two must agree (the LDS exchange leans on relaxed atomics staying in program order)
    "generic_laser": {"sort_min_points": 1, "fast_laser": 0},                        # the laser variance with its rotation term (frames that do not qualify for the short form)
    "guarded": {"sort_min_points": 1, "plain_loop": 0, "light_fast": 0},             # the walks' guarded chain loops and k_fuse_block's general rounds everywhere (what blocks / passes with values outside the plain range take)
}


def pytest_generate_tests(metafunc):
    # every GPU test runs on every pipeline (see the `pipeline` fixture)
    if metafunc.definition.get_closest_marker("gpu") is not None:
        metafunc.fixturenames.append("pipeline")
        one = metafunc.definition.get_closest_marker("one_pipeline") is not None
        metafunc.parametrize("pipeline", ["default"] if one else list(PIPELINES), indirect=True)


@pytest.fixture
def pipeline(request, monkeypatch):
    """GPU tests run once per entry of PIPELINES: with the library's own choice between the tile pipeline (k_frame / k_bin_wave +
    k_fuse_list) and the two sorted forms (gem_sort.hip), and with each sorted form forced for every pass, however small, at
    the pass counts only large maps would take."""
    from gem_amd import ElevationMap
    which = getattr(request, "param", "default")
    monkeypatch.setattr(ElevationMap, "base_debug", PIPELINES[which])
    return which
