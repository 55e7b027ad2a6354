// header_pin.cpp -- the host math a C++ user of include/gem/gem.hpp gets (gem::RobotMotionMapUpdater::compute,
// gem::SensorProcessorBase::frameParams) against the REFERENCE'S OWN code: RobotMotionMapUpdater.cpp (RMU.cpp:42-145) and
// SensorProcessorBase::readcomputerparam (SPB.cpp:270-290), compiled where they lie into oracle/_ref/libgem_ref_motion.so and
// libgem_ref_sensors.so (oracle/ref_build/build_ref.py).  The Python twin of the header (gem_amd/api.py) has had this test since
// round 4 (tests/test_motion_update.py); this one drives the header itself.  CPU only: nothing here touches a device.
//     header_pin <libgem_ref_motion.so> <libgem_ref_sensors.so>
#include "gem/gem.hpp"

#include <dlfcn.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static gem::Mat3 rot_zyx(double yaw, double pitch, double roll)      // R = Rz(yaw) Ry(pitch) Rx(roll), as gem_amd/synth.py
{
    const double cy = std::cos(yaw), sy = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch), cr = std::cos(roll), sr = std::sin(roll);
    return {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr,
            sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr,
            -sp, cp * sr, cp * cr};
}

static void quat_of(const gem::Mat3& R, double q[4])                   // Hamilton unit quaternion (w, x, y, z), w >= 0
{
    const double t = R[0] + R[4] + R[8];
    if (t > 0.0) { const double s = 2.0 * std::sqrt(t + 1.0); q[0] = 0.25 * s; q[1] = (R[7] - R[5]) / s; q[2] = (R[2] - R[6]) / s; q[3] = (R[3] - R[1]) / s; }
    else if (R[0] > R[4] && R[0] > R[8]) { const double s = 2.0 * std::sqrt(1.0 + R[0] - R[4] - R[8]); q[0] = (R[7] - R[5]) / s; q[1] = 0.25 * s; q[2] = (R[1] + R[3]) / s; q[3] = (R[2] + R[6]) / s; }
    else if (R[4] > R[8]) { const double s = 2.0 * std::sqrt(1.0 + R[4] - R[0] - R[8]); q[0] = (R[2] - R[6]) / s; q[1] = (R[1] + R[3]) / s; q[2] = 0.25 * s; q[3] = (R[5] + R[7]) / s; }
    else { const double s = 2.0 * std::sqrt(1.0 + R[8] - R[0] - R[4]); q[0] = (R[3] - R[1]) / s; q[1] = (R[2] + R[6]) / s; q[2] = (R[5] + R[7]) / s; q[3] = 0.25 * s; }
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), sg = q[0] < 0.0 ? -1.0 : 1.0;   // unit, w >= 0 (oracle/ref.py)
    for (int i = 0; i < 4; ++i) q[i] = sg * q[i] / n;
}

static long long ulps(float a, float b)                                // distance in representable floats (same sign, finite)
{
    std::int32_t ia, ib; std::memcpy(&ia, &a, 4); std::memcpy(&ib, &b, 4);
    if ((ia < 0) != (ib < 0)) return a == b ? 0 : (1ll << 40);
    return ia > ib ? (long long)ia - ib : (long long)ib - ia;
}

typedef void* (*motion_create_t)(double);
typedef void (*motion_destroy_t)(void*);
typedef int (*motion_update_t)(void*, const double*, const double*, const double*, const double*, int, double, float*);
typedef int (*readparam_t)(const double*, const double*, const double*, const double*, float*);

int main(int argc, char** argv)
{
    if (argc < 3) { std::printf("usage: header_pin libgem_ref_motion.so libgem_ref_sensors.so\n"); return 2; }
    void* lm = dlopen(argv[1], RTLD_NOW); void* ls = dlopen(argv[2], RTLD_NOW);
    if (!lm || !ls) { std::printf("dlopen: %s\n", dlerror()); return 2; }
    const motion_create_t m_create = (motion_create_t)dlsym(lm, "gemref_motion_create");
    const motion_destroy_t m_destroy = (motion_destroy_t)dlsym(lm, "gemref_motion_destroy");
    const motion_update_t m_update = (motion_update_t)dlsym(lm, "gemref_motion_update");
    const readparam_t readparam = (readparam_t)dlsym(ls, "gemref_readcomputerparam");
    if (!m_create || !m_destroy || !m_update || !readparam) { std::printf("dlsym failed\n"); return 2; }

    // ---- gem::RobotMotionMapUpdater::compute (gem.hpp) vs RobotMotionMapUpdater::update (RMU.cpp:42-90), random trajectories:
    //      pitched and rolled robots, full covariances, a rotated map, several covariance scales (tests/test_motion_update.py's)
    long long worst = 0; int steps = 0, exact = 0;
    for (int seed = 0; seed < 12; ++seed) {
        std::mt19937_64 rng(100 + seed);
        std::normal_distribution<double> N01(0.0, 1.0);
        std::uniform_real_distribution<double> U(-3.0, 3.0);
        const double scales[4] = {1.0, 0.5, 2.5, 1.3};
        const double scale = scales[seed % 4];
        gem::RobotMotionMapUpdater mine(scale);
        void* theirs = m_create(scale);
        gem::Mat3 mapR{1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (seed % 3 == 0) mapR = rot_zyx(0.4 * N01(rng), 0.0, 0.0);
        double mq[4]; quat_of(mapR, mq);
        gem::Vec3 pos{0, 0, 0};
        for (int k = 0; k < 20; ++k) {
            for (int i = 0; i < 3; ++i) pos[i] += 0.3 * N01(rng);
            const gem::Mat3 R = rot_zyx(U(rng), 0.25 * N01(rng), 0.25 * N01(rng));
            double A[36], cov[36];
            for (double& a : A) a = 1e-2 * N01(rng);
            for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { double s = 0; for (int a = 0; a < 6; ++a) s += A[i * 6 + a] * A[j * 6 + a]; cov[i * 6 + j] = s * (1.0 + 0.2 * k); }
            std::array<double, 36> c; std::memcpy(c.data(), cov, sizeof cov);
            const float a = mine.compute(pos, R, c, mapR);
            double q[4]; quat_of(R, q);
            float b = 0.f;
            const int did = m_update(theirs, pos.data(), q, cov, mq, 600, 1.0 + k, &b);
            CHECK(did == 1);
            // The doubles' summation order differs between the two (Eigen's unrolled reductions vs plain loops) and the value is a
            // DIFFERENCE of covariances: one float ulp, or -- where the difference cancels -- 1e-12 absolute.
            const long long d = ulps(a, b);
            if (!(d <= 1 || std::fabs((double)a - (double)b) <= 1e-12)) { std::printf("motion seed %d step %d: %.9g vs %.9g (%lld ulps)\n", seed, k, a, b, d); ++fails; }
            if (std::fabs((double)a - (double)b) > 1e-12) worst = d > worst ? d : worst;
            exact += d == 0; ++steps;
        }
        m_destroy(theirs);
    }
    std::printf("RobotMotionMapUpdater: %d steps, %d bit-equal, worst %lld float ulp(s)\n", steps, exact, worst);

    // ---- gem::SensorProcessorBase::frameParams (gem.hpp) vs SensorProcessorBase::readcomputerparam (SPB.cpp:270-290)
    int cases = 0; long long worst_p = 0;
    std::mt19937_64 rng(7);
    std::normal_distribution<double> N01(0.0, 1.0);
    std::uniform_real_distribution<double> U(-3.1, 3.1);
    for (int t = 0; t < 400; ++t) {
        const gem::Mat3 C_BM = rot_zyx(U(rng), 0.4 * N01(rng), 0.4 * N01(rng));      // rotationMapToBase_
        const gem::Mat3 C_SB = rot_zyx(U(rng), 0.8 * N01(rng), 0.8 * N01(rng));      // rotationBaseToSensor_
        const double b[3] = {0.5 * N01(rng), 0.5 * N01(rng), 0.3 + 0.5 * N01(rng)};  // translationBaseToSensorInBaseFrame_
        const double laser[3] = {0.018 * (1.0 + 0.1 * N01(rng)), 0.0006 * (1.0 + 0.1 * N01(rng)), 0.0015 * (1.0 + 0.1 * N01(rng))};
        // the three TF look-ups (SPB.cpp:97-124) as the members they fill: base<-sensor carries rotationBaseToSensor_ and the offset,
        // map<-base rotationMapToBase_
        gem::Mat4 baseFromSensor{C_SB[0], C_SB[1], C_SB[2], b[0], C_SB[3], C_SB[4], C_SB[5], b[1], C_SB[6], C_SB[7], C_SB[8], b[2], 0, 0, 0, 1};
        gem::Mat4 mapFromBase{C_BM[0], C_BM[1], C_BM[2], 1.0, C_BM[3], C_BM[4], C_BM[5], -2.0, C_BM[6], C_BM[7], C_BM[8], 0.4, 0, 0, 0, 1};
        gem::Mat4 I{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        gem::LaserSensorProcessor sp;
        sp.sensorParameters()["min_radius"] = laser[0]; sp.sensorParameters()["beam_angle"] = laser[1]; sp.sensorParameters()["beam_constant"] = laser[2];
        sp.updateTransformations(I, baseFromSensor, mapFromBase);
        const gem_frame_params p = sp.frameParams();
        float ref[27];
        CHECK(readparam(C_BM.data(), C_SB.data(), b, laser, ref) == 0);
        // the casts: min_r / beam_a / beam_c reach the kernel as floats (SPB.cpp:286-288: float = double)
        CHECK((float)p.sensor_params[0] == ref[0]); CHECK((float)p.sensor_params[1] == ref[1]); CHECK((float)p.sensor_params[2] == ref[2]);
        long long w = 0;
        for (int j = 0; j < 3; ++j) { w = std::max(w, ulps(p.sensor_jacobian[j], ref[3 + j])); w = std::max(w, ulps(p.P_mul_C_BM_T[j], ref[15 + j])); }
        for (int i = 0; i < 9; ++i) { w = std::max(w, ulps(p.C_SB_T[i], ref[6 + i])); w = std::max(w, ulps(p.B_r_BS_skew[i], ref[18 + i])); }
        if (w > 1) { std::printf("readcomputerparam case %d: %lld ulps\n", t, w); ++fails; }
        worst_p = std::max(worst_p, w);
        // the height window of GPUPointCloudprocess (SPB.cpp:183-184): doubles
        ++cases;
    }
    std::printf("readcomputerparam: %d cases, worst %lld float ulp(s)\n", cases, worst_p);
    std::printf(fails ? "FAILED\n" : "ok\n");
    return fails ? 1 : 0;
}
