"""Input colourisation (elevation_mapping/src/ElevationMapping.cpp:349-381), the step in front of the hot path:
every point is projected into the camera image, takes the BGR pixel it lands on and draws cv::circle(img, pixel, 1, colour) into
the image the LATER points sample.

CPU: hand-derived known answers for the oracle restatement (oracle/gem_oracle_color.c).
GPU: gem_colorize (sort by pixel, parent search, pointer jumping) against the oracle's point-by-point loop, exactly."""
import numpy as np
import pytest

from gem_amd import synth

F32 = np.float32


def pinhole(f, cx, cy):
    """T.camera with focal length f and principal point (cx, cy); the lidar frame IS the camera frame (z forward)."""
    return np.array([[f, 0, cx, 0], [0, f, cy, 0], [0, 0, 1, 0]], np.float64)


def make_image(h, w, seed=0):
    rng = np.random.default_rng(seed)
    return rng.integers(1, 256, (h, w, 3)).astype(np.uint8)


def pts_at(pixels, z=2.0, frac=0.25):
    """points (z forward) that a pinhole(1, 0, 0) camera maps to the given (x, y) pixels (+frac inside the pixel)"""
    out = np.zeros((len(pixels), 4), F32)
    for i, (x, y) in enumerate(pixels):
        out[i] = [(x + frac) * z, (y + frac) * z, z, 7.0 + i]
    return out


def word(bgr):
    return (int(bgr[2]) << 16) | (int(bgr[1]) << 8) | int(bgr[0])


def test_known_answers(oracle_mod):
    img = make_image(6, 8)
    P = pinhole(1.0, 0.0, 0.0)
    px = [(3, 3),    # A: the image's colour at row 3, column 3
          (4, 3),    # B: on A's circle -> A's colour
          (5, 3),    # C: on B's circle -> B's colour = A's
          (3, 1),    # D: untouched pixel
          (3, 2),    # E: on A's circle (above A) AND on D's (below D); D is later -> D's colour
          (9, 3),    # F: outside (x >= width)
          (0, 3),    # G: column 0 is rejected (x > 0)
          (3, 0),    # H: row 0 is rejected (y > 0)
          (7, 5),    # I: the last column / row are sampled; its circle is clipped
          (7, 4),    # J: above I -> I's colour
          (3, 3)]    # K: A's pixel again: B drew over it with A's colour, then E (later) with D's
    pts = pts_at(px)
    pts = np.vstack([pts, [[3.25 * -2, 3.25 * -2, -2.0, 5.0]]]).astype(F32)     # L: same pixel ratio as A but behind the camera
    o = oracle_mod.colorize(P, img, pts)
    a = word(img[3, 3]); d = word(img[1, 3]); i_ = word(img[5, 7])
    assert a != d and list(o["rgb"]) == [a, a, a, d, d, 0, 0, 0, i_, i_, d, 0]
    assert o["count"] == 8
    assert list(o["xyzi"][:, 3]) == [7, 8, 9, 10, 11, 0, 0, 0, 15, 16, 17, 0]
    # the drawn-on image: K drew D's colour over all four neighbours of (3, 3); the centre holds what E drew (D's colour too)
    assert word(o["image"][3, 2]) == d and word(o["image"][4, 3]) == d and word(o["image"][2, 3]) == d and word(o["image"][3, 4]) == d
    assert word(o["image"][3, 3]) == d and word(o["image"][3, 5]) == a      # (5, 3) was drawn by B only
    assert word(o["image"][5, 6]) == i_ and word(o["image"][4, 7]) == i_ and o["image"].shape == (6, 8, 3)
    # nothing outside the circles changed
    touched = np.zeros((6, 8), bool)
    for (x, y) in [p for k, p in enumerate(px) if k not in (5, 6, 7)]:
        for (qx, qy) in ((x - 1, y), (x + 1, y), (x, y - 1), (x, y + 1)):
            if 0 <= qx < 8 and 0 <= qy < 6:
                touched[qy, qx] = True
    assert np.array_equal(o["image"][~touched], img[~touched])


def test_truncation_and_float_rounding(oracle_mod):
    img = make_image(4, 4, 1)
    P = pinhole(1.0, 0.0, 0.0)
    # P_x is a float truncated toward zero: 1.999 -> 1; 0.999 -> 0 (rejected); -0.5 -> 0 (rejected, not -1)
    pts = np.array([[1.999, 1.5, 1, 1], [0.999, 1.5, 1, 1], [-0.5, 1.5, 1, 1], [2.0, 3.999, 1, 1]], F32)
    o = oracle_mod.colorize(P, img, pts)
    assert o["rgb"][0] == word(img[1, 1]) and o["rgb"][1] == 0 and o["rgb"][2] == 0 and o["rgb"][3] == word(img[3, 2])
    # the quotient is rounded to FLOAT before the truncation (EMg.cpp:319: "float P_x, P_y"): 2 - 2^-30 in double is 2.0f
    Pd = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], np.float64)
    Pd[0, 3] = -2.0 ** -30
    o = oracle_mod.colorize(Pd, img, np.array([[2.0, 1.5, 1, 1]], F32))
    assert o["rgb"][0] == word(img[1, 2])
    # z == 0: inf / NaN coordinates are outside
    o = oracle_mod.colorize(P, img, np.array([[1.5, 1.5, 0, 1], [0, 0, 0, 1]], F32))
    assert list(o["rgb"]) == [0, 0]


def test_lidar_to_image(oracle_mod):
    from gem_amd import ElevationMap
    rng = np.random.default_rng(3)
    a = rng.normal(size=(3, 4)) * 500; b = rng.normal(size=(4, 4))
    o = oracle_mod.lidar_to_image(a, b)
    assert np.allclose(o, a @ b, rtol=1e-13, atol=1e-10)
    assert np.array_equal(o, ElevationMap.lidar_to_image(a, b))
    assert o[1, 2] == ((a[1, 0] * b[0, 2] + a[1, 1] * b[1, 2]) + a[1, 2] * b[2, 2]) + a[1, 3] * b[3, 2]


def camera_scene(n, w, h, seed, spread=1.0):
    """a cloud in front of a pinhole camera looking along the lidar's x axis (T.lidar maps lidar axes to camera axes)"""
    rng = np.random.default_rng(seed)
    tl = np.array([[0, -1, 0, 0.02], [0, 0, -1, -0.05], [1, 0, 0, 0.1], [0, 0, 0, 1]], np.float64)
    tc = pinhole(0.8 * w, 0.5 * w + 0.3, 0.5 * h - 0.2)
    pts = np.empty((n, 4), F32)
    pts[:, 0] = rng.uniform(-2.0, 30.0, n)                       # some behind the camera
    pts[:, 1] = rng.normal(0, 6.0 * spread, n)
    pts[:, 2] = rng.normal(0, 3.0 * spread, n)
    pts[:, 3] = rng.uniform(1, 255, n)
    return tc, tl, pts


CASES = {
    "dense_small": dict(n=20_000, w=64, h=48, seed=1, spread=0.15),     # many points per pixel, long hand-me-down chains
    "vga": dict(n=100_000, w=640, h=480, seed=2, spread=0.6),
    "hd": dict(n=131_072, w=1280, h=720, seed=3, spread=1.0),
    "three_passes": dict(n=60_000, w=2048, h=1024, seed=4, spread=0.3),  # more than 2^20 pixels: three digits
    "one_pass": dict(n=5_000, w=32, h=24, seed=5, spread=0.2),           # at most 2^10 pixels: one digit
    "odd_sizes": dict(n=4_097, w=301, h=211, seed=6, spread=0.5),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", list(CASES))
def test_colorize_parity(oracle_mod, case):
    from gem_amd import ElevationMap
    c = CASES[case]
    tc, tl, pts = camera_scene(c["n"], c["w"], c["h"], c["seed"], c["spread"])
    img = make_image(c["h"], c["w"], c["seed"])
    P = ElevationMap.lidar_to_image(tc, tl)
    o = oracle_mod.colorize(P, img, pts)
    assert 0 < o["count"] < c["n"]
    m = ElevationMap(40, 0.1)
    rgb, out = m.colorize(P, img, pts)
    assert np.array_equal(rgb, o["rgb"])
    assert np.array_equal(out, o["xyzi"])
    # the order dependence is live in this case: sampling the untouched image gives something else
    plain = oracle_mod.colorize(P, img, pts[:1])["rgb"][0]
    assert plain == o["rgb"][0]
    if case in ("dense_small", "vga"):
        independent = np.array([oracle_mod.colorize(P, img, pts[i:i + 1])["rgb"][0] for i in range(2000)])
        assert (independent != o["rgb"][:2000]).any()


@pytest.mark.gpu
def test_colorize_adversarial_chains(oracle_mod):
    """every point on one of two neighbouring pixels, alternating: the chain of hand-me-downs is as long as the cloud; then all
    points on ONE pixel (nobody draws on the centre: everyone reads the image); then a zig-zag walk over a row and back"""
    from gem_amd import ElevationMap
    img = make_image(16, 16, 9)
    P = pinhole(1.0, 0.0, 0.0)
    m = ElevationMap(40, 0.1)
    n = 30_000
    zig = [(5 + (i & 1), 7) for i in range(n)]
    one = [(9, 9)] * 5000
    walk = [(1 + (i % 14), 3) for i in range(4000)] + [(14 - (i % 14), 3) for i in range(4000)]
    for px in (zig, one, walk, zig[:1], zig[:2]):
        pts = pts_at(px)
        o = oracle_mod.colorize(P, img, pts)
        rgb, out = m.colorize(P, img, pts)
        assert np.array_equal(rgb, o["rgb"]) and np.array_equal(out, o["xyzi"])
    assert set(oracle_mod.colorize(P, img, pts_at(zig))["rgb"]) == {word(img[7, 5])}


@pytest.mark.gpu
def test_colorize_device_tensors_padded_rows_and_empty(oracle_mod):
    import torch
    from gem_amd import ElevationMap
    tc, tl, pts = camera_scene(50_000, 320, 200, 11, 0.4)
    P = ElevationMap.lidar_to_image(tc, tl)
    padded = np.zeros((200, 1024), np.uint8)                        # rows of 1024 bytes, 960 used
    img = make_image(200, 320, 11)
    padded[:, :960] = img.reshape(200, 960)
    o = oracle_mod.colorize(P, img, pts)
    m = ElevationMap(40, 0.1)
    d_img = torch.as_strided(torch.from_numpy(padded).cuda(), (200, 320, 3), (1024, 3, 1))
    d_pts = torch.from_numpy(pts).cuda()
    rgb, out = m.colorize(P, d_img, d_pts)
    assert out.data_ptr() == d_pts.data_ptr()                       # in place
    m.synchronize()                                                 # the call only enqueued on the handle's stream
    assert np.array_equal(rgb.cpu().numpy().view(np.uint32), o["rgb"])
    assert np.array_equal(d_pts.cpu().numpy(), o["xyzi"])
    assert np.array_equal(d_img.cpu().numpy(), img)                 # the caller's image is not drawn on
    rgb0, out0 = m.colorize(P, img, pts[:0])
    assert rgb0.shape == (0,) and out0.shape == (0, 4)
    with pytest.raises(RuntimeError):
        m.colorize(P, np.zeros((0, 0, 3), np.uint8), pts[:4])


@pytest.mark.gpu
def test_coloured_cloud_through_the_path(oracle_mod):
    """colourise -> add: the map's colour layers equal the oracle's fed by the oracle's colourisation"""
    from gem_amd import ElevationMap
    wl = synth.config_c1()
    cloud = wl.clouds[0].copy()
    # a camera above the scene looking down the map's -z is no LiDAR geometry; any projection will do for the data flow
    tl = np.array([[0, -1, 0, 0.0], [0, 0, -1, 0.0], [1, 0, 0, 0.5], [0, 0, 0, 1]], np.float64)
    tc = pinhole(200.0, 160.0, 120.0)
    P = ElevationMap.lidar_to_image(tc, tl)
    img = make_image(240, 320, 21)
    o = oracle_mod.colorize(P, img, cloud)
    assert o["count"] > 100
    gpu = ElevationMap(wl.length, wl.resolution)
    ref = oracle_mod.OracleMap(wl.length, wl.resolution)
    rgb, pts = gpu.colorize(P, img, cloud)
    gpu.add(wl.frames[0], pts, rgb=rgb)
    ref.add(wl.frames[0], o["xyzi"], rgb=o["rgb"])
    for layer in ("elevation", "variance", "color_r", "color_g", "color_b", "intensity"):
        assert np.array_equal(gpu.layer(layer), ref.layer(layer)), layer
    assert (ref.layer("color_r") != 0).any()
