"""ElevationMap::show's cell loop (elevation_mapping/src/ElevationMap.cpp:85-149), the step after the hot path:
nine flat storage-indexed arrays -> visualMap_'s grid_map layers (column-major by buffer index, NaN where the cell has no
elevation / traversability), a coloured point per kept cell in grid_map's iteration order, the orthomosaic.

CPU: hand-derived known answers for the oracle restatement (oracle/gem_oracle_show.c), incl. grid_map's getPositionFromIndex.
GPU: gem_show (device compaction, order preserved) against the oracle, exactly."""
import numpy as np
import pytest

from gem_amd import synth

F32 = np.float32


def test_show_known_answers(oracle_mod):
    # L = 8, res = 0.5: map length 4 m.  After Move to (1.1, -0.6) the centre is (1.0, -0.5) and the start index (6, 1).
    m = oracle_mod.OracleMap(8, 0.5)
    c, s, _ = m.move([1.1, -0.6, 0.0])
    assert np.allclose(c, [1.0, -0.5]) and list(s) == [6, 1]
    e = np.full((8, 8), -10, F32); t = np.full((8, 8), -10, F32)
    e[2, 3], t[2, 3] = 1.5, 0.4                       # kept
    e[0, 0], t[0, 0] = 0.25, 0.9                      # kept
    e[5, 5] = 2.0                                     # elevation but no traversability: dropped (EM.cpp:101)
    t[7, 7] = 0.3                                     # traversability but no elevation: dropped
    e[4, 1], t[4, 1] = 0.7, np.nan                    # NaN traversability: dropped
    m.set_layer("elevation", e); m.set_layer("traver", t)
    m.set_layer("color_r", np.full((8, 8), 200)); m.set_layer("color_g", np.full((8, 8), 17)); m.set_layer("color_b", np.full((8, 8), 255))
    o = m.show()
    assert o["count"] == 2
    # iteration order = linear index of the column-major matrix: (row 0, col 0) = 0 comes before (row 2, col 3) = 3 * 8 + 2
    # cell (0, 0): unwrapped (0 - 6 + 8, 0 - 1 + 8) = (2, 7): x = 1.0 + (2.0 - 0.25) - 0.5 * 2 = 1.75, y = -0.5 + 1.75 - 3.5 = -2.25
    # cell (2, 3): unwrapped (4, 2):                          x = 1.0 + 1.75 - 2.0 = 0.75,            y = -0.5 + 1.75 - 1.0 = 0.25
    assert np.array_equal(o["points_xyz"], np.array([[1.75, -2.25, 0.25], [0.75, 0.25, 1.5]], F32))
    assert np.array_equal(o["points_rgb"], np.array([[200, 17, 255]] * 2, np.uint8))
    v = o["visual"]                                   # [layer][col][row]: Eigen column-major
    kept = ~np.isnan(v[0])
    assert kept.sum() == 2 and kept[0, 0] and kept[3, 2]
    assert v[0][3, 2] == F32(1.5) and v[4][3, 2] == F32(0.4) and v[5][3, 2] == 200.0 and v[7][0, 0] == 255.0
    assert all(np.array_equal(np.isnan(v[l]), ~kept) for l in range(9))
    img = o["image_bgr"]                              # unwrapped pixel, b g r
    assert tuple(img[2, 7]) == (255, 17, 200) and tuple(img[4, 2]) == (255, 17, 200) and int(img.sum()) == 2 * (255 + 17 + 200)


def scene(make, L=100, res=0.1):
    m = make(L, res)
    m.move([1.37, -0.84, 0.0])
    rng = np.random.default_rng(4)
    c = synth.random_cloud(17, 40_000, 0.42 * L * res, z_sigma=0.15)
    rgb = rng.integers(0, 256, (c.shape[0], 3)).astype(np.uint32)
    f = synth._frame_for(synth.pose_matrix(1.3, -0.8, 0.4, 0.5, 0.01, -0.02), synth.SensorModel.velodyne())
    m.add(f, c, rgb=(rgb[:, 0] << 16) | (rgb[:, 1] << 8) | rgb[:, 2])
    feat = m.map_feature()
    return m, feat


@pytest.mark.gpu
@pytest.mark.parametrize("L", [100, 75, 251])          # (odd sizes: the nine-layer block and the point list start at addresses that are only word-aligned)
def test_show_parity(oracle_mod, L):
    from gem_amd import ElevationMap
    gpu, _ = scene(ElevationMap, L)
    ora, feat = scene(oracle_mod.OracleMap, L)
    # continue from the oracle's own traversability stage on both sides (the device's slope / traver may differ in a last bit,
    # tests/test_map_feature.py), so that the comparison below is exact
    for name in ("traver",):
        gpu.set_layer(name, ora.layer(name))
    g = gpu.show()
    o = ora.show(rough=gpu.layer("rough"), slope=gpu.layer("slope"))
    assert g["count"] == o["count"] and o["count"] > 900
    assert np.array_equal(g["points_xyz"], o["points_xyz"])            # order and values: the device compaction keeps grid_map's iteration order
    assert np.array_equal(g["points_rgb"], o["points_rgb"])
    assert np.array_equal(g["image_bgr"], o["image_bgr"]) and int(o["image_bgr"].astype(np.int64).sum()) > 0
    assert np.array_equal(np.isnan(g["visual"]), np.isnan(o["visual"]))
    assert np.array_equal(np.nan_to_num(g["visual"], nan=-1e9), np.nan_to_num(o["visual"], nan=-1e9))
    # explicit geometry (doubles, as the node passes them: EMg.cpp:178) instead of the handle's
    g2 = gpu.show(map_length=L * 0.1, resolution=0.1, position=[1.4, -0.8])
    o2 = ora.show(rough=gpu.layer("rough"), slope=gpu.layer("slope"), map_length=L * 0.1, resolution=0.1, position=[1.4, -0.8])
    assert np.array_equal(g2["points_xyz"], o2["points_xyz"]) and not np.array_equal(g2["points_xyz"], g["points_xyz"])


@pytest.mark.gpu
def test_show_of_an_empty_map():
    from gem_amd import ElevationMap
    g = ElevationMap(64, 0.1).show()
    assert g["count"] == 0 and np.isnan(g["visual"]).all() and not g["image_bgr"].any()
