// Host-side C++ check of the drop-in boundary: the nine GEM-signature symbols (gem_compat_eigen.hpp)
// and the C++ facade (gem.hpp) against libgem_hip.so.  Without a GPU it verifies that everything
// links and that creation fails loudly (no CPU fallback); with a GPU it runs a small known-answer case.
#include "gem/gem_compat_eigen.hpp"
#include "gem/gem.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main(int argc, char** argv)
{
    const bool expect_gpu = argc > 1 && std::atoi(argv[1]) != 0;
    std::printf("abi %d\n", gem_abi_version());
    CHECK(gem_abi_version() == GEM_ABI_VERSION);

    if (!expect_gpu) {
        bool threw = false;
        try { gem::ElevationMap m(64, 0.1f); } catch (const gem::Error& e) { threw = true; std::printf("no device: %s\n", e.what()); CHECK(e.code() == GEM_ERR_NO_DEVICE); }
        CHECK(threw);
        std::printf(fails ? "FAILED\n" : "OK (no GPU: link + loud failure)\n");
        return fails;
    }

    // ---- the reference's call sequence, through the adapter symbols -------------------------------
    const int L = 64, N = 6;
    Init_GPU_elevationmap(L, 0.1f, 2.5f, 0.7f);
    float pos[3] = {0.f, 0.f, 0.f}, center[2], shift[2]; int start[2];
    Move(pos, 0.1f, L, center, start, shift);
    CHECK(center[0] == 0.f && start[0] == 0);

    Eigen::Matrix4f T; for (int i = 0; i < 4; ++i) T(i, i) = 1.f;
    Eigen::RowVector3f Js; Js(0, 2) = 1.f;
    Eigen::Matrix3f Q, C, Bs; for (int i = 0; i < 3; ++i) C(i, i) = 1.f;
    // sensor-frame points: 4 survive the reference filter (y <= -1, outside the +-1.5 box), 2 do not
    float x[N] = {1.95f, 1.95f, 1.97f, -2.5f, 0.5f, 1.0f};
    float y[N] = {-2.05f, -2.05f, -2.03f, -1.7f, -0.5f, 2.0f};
    float z[N] = {0.50f, 0.52f, 0.51f, 0.1f, 0.3f, 0.3f};
    int idx[N]; float var[N], xt[N], yt[N], zt[N];
    Process_points(idx, x, y, z, var, xt, yt, zt, T, N, -5.0, 0.8, 0.018f, 0.0006f, 0.0015f, Js, Q, C, Js, Bs);
    for (int i = 0; i < 4; ++i) { CHECK(idx[i] >= 0); CHECK(var[i] == 0.018f * 0.018f); CHECK(zt[i] == z[i]); }
    CHECK(idx[4] == -1 && idx[5] == -1 && zt[4] == -1.f && var[5] == -1.f);
    CHECK(idx[0] == idx[1] && idx[0] == idx[2]);     // same 0.1 m cell
    // ix = (int)(32 - 1.95/0.1) = (int)12.5 = 12, iy = (int)(32 + 2.05/0.1) = (int)52.5 = 52
    CHECK(idx[0] == 12 * L + 52);

    int R[N] = {0}, G[N] = {0}, B[N] = {0}; float inten[N] = {0};
    Fuse(L, N, idx, R, G, B, inten, zt, var);
    Mapvar_update(L, 1e-5f);
    std::vector<float> e(L * L), v(L * L), rough(L * L), slope(L * L), trav(L * L), in(L * L);
    std::vector<int> cr(L * L), cg(L * L), cb(L * L);
    Map_feature(L, e.data(), v.data(), cr.data(), cg.data(), cb.data(), rough.data(), slope.data(), trav.data(), in.data());
    // cell of points 0..2: three-point Kalman chain in input order, then +1e-5
    float ee = 0.50f, ss = var[0];
    for (int i = 1; i < 3; ++i) { float s2 = ss < 1e-4f ? 1e-4f : ss; float en = (s2 * z[i] + var[i] * ee) / (s2 + var[i]); ss = (var[i] * s2) / (var[i] + s2); ee = en; }
    if (ss < 1e-4f) ss = 1e-4f;
    CHECK(e[idx[0]] == ee);
    CHECK(v[idx[0]] == ss + 1e-5f);
    CHECK(e[idx[3]] == 0.1f);
    CHECK(e[0] == -10.f && v[0] == 1e-4f + 1e-5f);
    // traversability stage: a cell without elevation reports nothing, an isolated cell has < 8 neighbours -> traver -10
    CHECK(rough[0] == 0.f && slope[0] == 0.f && trav[0] == -10.f);
    CHECK(trav[idx[0]] == -10.f && rough[idx[0]] == 0.f);
    Raytracing(L);

    // ---- the C++ facade --------------------------------------------------------------------------
    {
        gem::ElevationMap m(L, 0.1f);
        gem::LaserSensorProcessor sp;
        sp.sensorParameters()["min_radius"] = 0.018; sp.sensorParameters()["beam_angle"] = 0.0006; sp.sensorParameters()["beam_constant"] = 0.0015;
        sp.setIgnorePoints(-5.0, 0.8);
        gem::Mat4 I{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        sp.updateTransformations(I, I, I);
        std::vector<gem::PointXYZRGBICT> cloud(N);
        for (int i = 0; i < N; ++i) { cloud[i] = gem::PointXYZRGBICT{}; cloud[i].x = x[i]; cloud[i].y = y[i]; cloud[i].z = z[i]; cloud[i].intensity = 1.f; }
        int pi[N], pr[N], pg[N], pb[N]; float pint[N], ph[N], pv[N];
        CHECK(sp.process(m, cloud.data(), N, pr, pg, pb, pi, pint, ph, pv));
        for (int i = 0; i < N; ++i) { CHECK(pi[i] == idx[i]); CHECK(pv[i] == var[i]); }
        m.fuse(N, pi, pr, pg, pb, pint, ph, pv);
        gem::RobotMotionMapUpdater rm;
        std::array<double, 36> cov{}; cov[2 * 6 + 2] = 4e-5;        // z variance only
        const float u = rm.update(m, {0.1, 0, 0}, {1, 0, 0, 0, 1, 0, 0, 0, 1}, cov);
        CHECK(std::fabs(u - 4e-5f) < 1e-9f);
        const std::vector<float> e2 = m.layer(GEM_LAYER_ELEVATION), v2 = m.layer(GEM_LAYER_VARIANCE);
        CHECK(e2[idx[0]] == ee);
        CHECK(v2[idx[0]] == ss + u);
        // the fused one-call path gives the same map
        gem::ElevationMap m2(L, 0.1f);
        std::vector<float> xyzi(4 * N);
        for (int i = 0; i < N; ++i) { xyzi[4 * i] = x[i]; xyzi[4 * i + 1] = y[i]; xyzi[4 * i + 2] = z[i]; xyzi[4 * i + 3] = 1.f; }
        m2.add(sp.frameParams(), xyzi.data(), N);
        m2.update(u);
        CHECK(m2.layer(GEM_LAYER_ELEVATION) == e2);
        CHECK(m2.layer(GEM_LAYER_VARIANCE) == v2);
        const std::vector<float> gm = m2.gridMapLayer(GEM_LAYER_ELEVATION);
        CHECK(std::isnan(gm[0]));
        // ElevationMap::show's feed: an isolated cell has no traversability (fewer than 8 neighbours), so nothing is kept ...
        m2.mapFeature();
        gem::ElevationMap::Shown sh = m2.show();
        CHECK(sh.count == 0 && std::isnan(sh.visual[0]) && sh.pointsXYZ.empty());
        // ... give every cell one and the fused cells appear, in grid_map's iteration order, with grid_map's positions
        std::vector<float> trav(static_cast<size_t>(L) * L, 0.5f);
        CHECK(gem_set_layer(m2.handle(), GEM_LAYER_TRAVER, trav.data()) == GEM_OK);
        sh = m2.show();
        int touched = 0;
        for (float ev : e2) touched += ev != -10.f;
        CHECK(sh.count == touched && touched > 0);
        CHECK(static_cast<int>(sh.pointsXYZ.size()) == 3 * touched);
        // the point of storage cell (ix, iy) sits at centre + (L res / 2 - res / 2) - res * index (start index 0, centre 0)
        const int c0 = idx[3], ix = c0 / L, iy = c0 % L;
        bool found = false;
        for (int k = 0; k < sh.count; ++k) {
            const float px = sh.pointsXYZ[3 * k], py = sh.pointsXYZ[3 * k + 1], pz = sh.pointsXYZ[3 * k + 2];
            const double wx = (0.0 + (0.5 * (L * (double)0.1f) - 0.5 * (double)0.1f)) + (double)0.1f * (double)(-ix);
            const double wy = (0.0 + (0.5 * (L * (double)0.1f) - 0.5 * (double)0.1f)) + (double)0.1f * (double)(-iy);
            if (px == (float)wx && py == (float)wy) { found = true; CHECK(pz == e2[c0]); }
        }
        CHECK(found);
        // input colourisation (ElevationMapping.cpp:349-381): a 6 x 4 image, an identity camera; point 1 lands on point 0's
        // circle and takes its colour, point 2 is outside
        std::vector<unsigned char> img(6 * 4 * 3);
        for (size_t k = 0; k < img.size(); ++k) img[k] = static_cast<unsigned char>(10 + k);
        const std::array<double, 12> tc = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        const gem::Mat4 tl = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        const std::array<double, 12> P = gem::ElevationMap::lidarToImage(tc, tl);
        CHECK(P == tc);
        gem::PointXYZRGBICT cl[3] = {};
        cl[0].x = 2.5f; cl[0].y = 1.5f; cl[0].z = 1.f; cl[0].intensity = 9.f;       // pixel (2, 1)
        cl[1].x = 3.5f; cl[1].y = 1.5f; cl[1].z = 1.f; cl[1].intensity = 8.f;       // pixel (3, 1): on the circle of point 0
        cl[2].x = 7.5f; cl[2].y = 1.5f; cl[2].z = 1.f; cl[2].intensity = 7.f;       // outside
        m2.colorize(P, 6, 4, img.data(), 0, cl, 3);
        const unsigned char* p21 = &img[(1 * 6 + 2) * 3];
        CHECK(cl[0].b == p21[0] && cl[0].g == p21[1] && cl[0].r == p21[2] && cl[0].intensity == 9.f);
        CHECK(cl[1].b == p21[0] && cl[1].g == p21[1] && cl[1].r == p21[2] && cl[1].intensity == 8.f);
        CHECK(cl[2].b == 0 && cl[2].g == 0 && cl[2].r == 0 && cl[2].intensity == 0.f);
    }
    std::printf(fails ? "FAILED (%d)\n" : "OK\n", fails);
    return fails;
}
