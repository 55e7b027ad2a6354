// gem.hpp -- C++ host-side mirror of the reference's interface for the hot path, header-only, on top
// of the C ABI (include/gem_hip.h).  No Eigen / ROS / PCL / kindr types: rotations are row-major
// double[9], transforms row-major double[16]; the unmodified ROS node converts at the call site
//  or uses include/gem/gem_compat_eigen.hpp, which re-creates the nine
// C++-linkage symbols of the reference's libgpu.so.
//
// Mirrors (names and argument meaning):
//   SensorProcessorBase::updateTransformations / readcomputerparam / GPUPointCloudprocess / process
//       elevation_mapping/src/sensor_processors/SensorProcessorBase.cpp:97-124, 270-290, 126-211, 66-94
//   Laser / StructuredLight / Stereo / Perfect ::readParameters  (the *.yaml keys they read)
//   ElevationMapping::processpoints / updateMapLocation           src/ElevationMapping.cpp:254-283, 1001-1044
//   RobotMotionMapUpdater::update (+ computeReducedCovariance / computeRelativeCovariance)
//       src/RobotMotionMapUpdater.cpp:42-90, 92-109, 111-145
//   ElevationMap layer names                                      src/ElevationMap.cpp:43-44
#pragma once

#include "../gem_hip.h"

#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace gem {

class Error : public std::runtime_error {
public:
    Error(int code, const std::string& what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }
private:
    int code_;
};

// PointXYZRGBICT (include/elevation_mapping/PointXYZRGBICT.hpp:26-48): the reference's input record.
struct alignas(16) PointXYZRGBICT {
    float x, y, z, pad;
    union { float rgb; struct { std::uint8_t b, g, r, a; }; };
    float covariance, intensity, travers;
};
static_assert(sizeof(PointXYZRGBICT) == 32, "Anypoint is 32 bytes");

using Mat3 = std::array<double, 9>;     // row-major
using Mat4 = std::array<double, 16>;    // row-major homogeneous transform
using Vec3 = std::array<double, 3>;

inline Mat3 transposed(const Mat3& m) { return {m[0], m[3], m[6], m[1], m[4], m[7], m[2], m[5], m[8]}; }
inline Mat3 mul(const Mat3& a, const Mat3& b)
{
    Mat3 c{};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
    return c;
}
inline Mat3 rotation_of(const Mat4& t) { return {t[0], t[1], t[2], t[4], t[5], t[6], t[8], t[9], t[10]}; }
inline Vec3 translation_of(const Mat4& t) { return {t[3], t[7], t[11]}; }

class ElevationMap;

// ---------------------------------------------------------------------------------------------
// SensorProcessorBase and its four subclasses
// ---------------------------------------------------------------------------------------------
class SensorProcessorBase {
public:
    virtual ~SensorProcessorBase() = default;

    // sensor_processor/ignore_points_above|below (SensorProcessorBase.cpp:61-62)
    void setIgnorePoints(double below, double above) { ignoreLower_ = below; ignoreUpper_ = above; }
    // the hard-coded sensor-frame reject filter of gpu_process.cu:393; on by default like the reference
    void setRejectFilter(const gem_reject_filter& f) { filter_ = f; }
    void setRotationVariance(const std::array<float, 9>& q) { rotationVariance_ = q; }

    // What the three TF lookups of SensorProcessorBase.cpp:97-124 return (map<-sensor, base<-sensor, map<-base).
    void updateTransformations(const Mat4& mapFromSensor, const Mat4& baseFromSensor, const Mat4& mapFromBase)
    {
        transformationSensorToMap_ = mapFromSensor;
        rotationBaseToSensor_ = rotation_of(baseFromSensor);
        translationBaseToSensorInBaseFrame_ = translation_of(baseFromSensor);
        rotationMapToBase_ = rotation_of(mapFromBase);
        translationMapToBaseInMapFrame_ = translation_of(mapFromBase);
    }

    // readcomputerparam (SensorProcessorBase.cpp:270-290) + the casts of GPUPointCloudprocess (:171-184).
    gem_frame_params frameParams() const
    {
        gem_frame_params p{};
        for (int i = 0; i < 16; ++i) p.T[i] = static_cast<float>(transformationSensorToMap_[i]);           // :175-179
        p.lower = translationMapToBaseInMapFrame_[2] + ignoreLower_;                                        // :183
        p.upper = translationMapToBaseInMapFrame_[2] + ignoreUpper_;                                        // :184
        const Mat3 C_BM_T = transposed(rotationMapToBase_);
        const Mat3 C_SB_T = transposed(rotationBaseToSensor_);
        const Mat3 J = mul(C_BM_T, C_SB_T);                                                                  // :275 (double product, float cast)
        for (int j = 0; j < 3; ++j) {
            p.sensor_jacobian[j] = static_cast<float>(J[6 + j]);
            p.P_mul_C_BM_T[j] = static_cast<float>(C_BM_T[6 + j]);                                           // :281-282
        }
        for (int i = 0; i < 9; ++i) { p.C_SB_T[i] = static_cast<float>(C_SB_T[i]); p.rotation_variance[i] = rotationVariance_[i]; }
        const float bx = static_cast<float>(translationBaseToSensorInBaseFrame_[0]);
        const float by = static_cast<float>(translationBaseToSensorInBaseFrame_[1]);
        const float bz = static_cast<float>(translationBaseToSensorInBaseFrame_[2]);
        const float sk[9] = {0.f, -bz, by, bz, 0.f, -bx, -by, bx, 0.f};                                      // :284 (kindr skew)
        std::memcpy(p.B_r_BS_skew, sk, sizeof(sk));
        p.sensor_model = sensorModel();
        fillSensorParams(p.sensor_params);
        p.filter = filter_;
        p.original_width = originalWidth_;
        return p;
    }

    // SensorProcessorBase::process (SensorProcessorBase.cpp:66-94): same out-arrays as the reference,
    // caller-owned, length = cloud size.  The cloud is assumed NaN-free (cleanPointCloud, Laser.cpp:50-59).
    bool process(ElevationMap& map, const PointXYZRGBICT* cloud, int n,
                 int* point_colorR, int* point_colorG, int* point_colorB, int* point_index,
                 float* point_intensity, float* point_height, float* point_var);

    std::map<std::string, double>& sensorParameters() { return sensorParameters_; }
    void setOriginalWidth(int w) { originalWidth_ = w; }

protected:
    virtual int  sensorModel() const = 0;
    virtual void fillSensorParams(double out[8]) const = 0;
    double param(const char* k) const { auto it = sensorParameters_.find(k); return it == sensorParameters_.end() ? 0.0 : it->second; }

    Mat4 transformationSensorToMap_{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    Mat3 rotationBaseToSensor_{1, 0, 0, 0, 1, 0, 0, 0, 1};
    Vec3 translationBaseToSensorInBaseFrame_{0, 0, 0};
    Mat3 rotationMapToBase_{1, 0, 0, 0, 1, 0, 0, 0, 1};
    Vec3 translationMapToBaseInMapFrame_{0, 0, 0};
    double ignoreUpper_ = std::numeric_limits<double>::infinity();
    double ignoreLower_ = -std::numeric_limits<double>::infinity();
    gem_reject_filter filter_{1, 1.5f, 1.5f, 1.0f, 0.0f};
    std::array<float, 9> rotationVariance_{};        // zero, SensorProcessorBase.cpp:202-204
    std::map<std::string, double> sensorParameters_;
    int originalWidth_ = 0;
};

class LaserSensorProcessor : public SensorProcessorBase {        // LaserSensorProcessor.cpp:41-48
protected:
    int sensorModel() const override { return GEM_MODEL_LASER; }
    void fillSensorParams(double o[8]) const override { o[0] = param("min_radius"); o[1] = param("beam_angle"); o[2] = param("beam_constant"); }
};
class StructuredLightSensorProcessor : public SensorProcessorBase {   // StructuredLightSensorProcessor.cpp:36-50
protected:
    int sensorModel() const override { return GEM_MODEL_STRUCTURED_LIGHT; }
    void fillSensorParams(double o[8]) const override
    {
        o[0] = param("normal_factor_a"); o[1] = param("normal_factor_b"); o[2] = param("normal_factor_c");
        o[3] = param("normal_factor_d"); o[4] = param("normal_factor_e"); o[5] = param("lateral_factor");
    }
};
class StereoSensorProcessor : public SensorProcessorBase {       // StereoSensorProcessor.cpp:23-34
protected:
    int sensorModel() const override { return GEM_MODEL_STEREO; }
    void fillSensorParams(double o[8]) const override
    {
        o[0] = param("p_1"); o[1] = param("p_2"); o[2] = param("p_3"); o[3] = param("p_4"); o[4] = param("p_5");
        o[5] = param("lateral_factor"); o[6] = param("depth_to_disparity_factor");
    }
};
class PerfectSensorProcessor : public SensorProcessorBase {
protected:
    int sensorModel() const override { return GEM_MODEL_PERFECT; }
    void fillSensorParams(double*) const override {}
};

// ---------------------------------------------------------------------------------------------
// ElevationMap: the device-resident robot-centric map
// ---------------------------------------------------------------------------------------------
class ElevationMap {
public:
    // layer names of the reference's two GridMaps (ElevationMap.cpp:43-44)
    static const std::vector<std::string>& rawMapLayers()
    {
        static const std::vector<std::string> v{"elevation", "min_height", "height", "variance", "horizontal_variance_x",
            "horizontal_variance_y", "horizontal_variance_xy", "color", "timestamp", "time", "lowest_scan_point",
            "sensor_x_at_lowest_scan", "sensor_y_at_lowest_scan", "sensor_z_at_lowest_scan"};
        return v;
    }
    static const std::vector<std::string>& visualMapLayers()
    {
        static const std::vector<std::string> v{"elevation", "variance", "rough", "slope", "traver", "color_r", "color_g", "color_b", "intensity"};
        return v;
    }

    // Init_GPU_elevationmap(length, resolution, mahalanobis, obstacle_threshold)  (ElevationMapping.cpp:199)
    ElevationMap(int length, float resolution, float mahalanobisThreshold = 5.0f, float obstacleThreshold = 0.7f, int device = -1)
    {
        gem_map_config cfg{};
        cfg.length = length; cfg.resolution = resolution; cfg.mahalanobis_threshold = mahalanobisThreshold;
        cfg.variance_floor = 0.0001f; cfg.obstacle_threshold = obstacleThreshold; cfg.device = device;
        const int rc = gem_create(&cfg, &h_);
        if (rc != GEM_OK) throw Error(rc, std::string("gem_create: ") + gem_last_error(nullptr));
        length_ = length; resolution_ = resolution;
    }
    ~ElevationMap() { if (h_) gem_destroy(h_); }
    ElevationMap(const ElevationMap&) = delete;
    ElevationMap& operator=(const ElevationMap&) = delete;

    gem_handle* handle() const { return h_; }
    int length() const { return length_; }
    float resolution() const { return resolution_; }

    // ElevationMapping::updateMapLocation -> Move (ElevationMapping.cpp:1032) + ElevationMap::move (ElevationMap.cpp:172-177)
    void move(const float position[3], float center[2] = nullptr, int startIndex[2] = nullptr, float alignedShift[2] = nullptr)
    { check(gem_move(h_, position, center, startIndex, alignedShift), "gem_move"); }

    // the upstream ElevationMap::add(pointCloud, variances, ...) shape: project + bin + fuse in one call.
    void add(const gem_frame_params& frame, const float* xyzi, int n, const std::uint32_t* rgb = nullptr, const int* origIndex = nullptr)
    { check(gem_add(h_, &frame, n, xyzi, rgb, origIndex), "gem_add"); }
    void addDevice(const gem_frame_params& frame, const void* d_xyzi, int n, const void* d_rgb = nullptr, const void* d_origIndex = nullptr)
    { check(gem_add_device(h_, &frame, n, d_xyzi, d_rgb, d_origIndex), "gem_add_device"); }

    // Fuse(length, point_num, index, R, G, B, intensity, height, var)  (ElevationMapping.cpp:280)
    void fuse(int n, const int* index, const int* R, const int* G, const int* B, const float* intensity, const float* height, const float* var)
    { check(gem_fuse(h_, n, index, R, G, B, intensity, height, var), "gem_fuse"); }

    // Mapvar_update(length, var_update)  (RobotMotionMapUpdater.cpp:81)
    void update(float varianceUpdate) { check(gem_mapvar_update(h_, varianceUpdate), "gem_mapvar_update"); }

    // Map_feature(...) (ElevationMapping.cpp:410): traversability stage on the fused map; computes the ROUGH, SLOPE and
    // TRAVER layers on the device and copies out the arrays that are not null (flat storage order, length^2 floats)
    void mapFeature(float* rough = nullptr, float* slope = nullptr, float* traver = nullptr)
    { check(gem_map_feature(h_, nullptr, nullptr, nullptr, nullptr, nullptr, rough, slope, traver, nullptr), "gem_map_feature"); }

    // Raytracing(length) (ElevationMapping.cpp:421): visibility clean-up; needs trackLowest(true) while the frame is fused
    void trackLowest(bool on) { check(gem_set_lowest_tracking(h_, on ? 1 : 0), "gem_set_lowest_tracking"); }
    void raytracing() { check(gem_raytracing(h_), "gem_raytracing"); }

    // flat [storage_x * length + storage_y] array, the layout ElevationMap::show indexes (ElevationMap.cpp:98-111)
    std::vector<float> layer(int which) const
    {
        std::vector<float> v(static_cast<size_t>(length_) * length_);
        if (which >= GEM_LAYER_COLOR_R && which <= GEM_LAYER_COLOR_B) throw Error(GEM_ERR_INVALID, "colour layers are int32: use colorLayer()");
        check(gem_get_layer(h_, which, GEM_LAYOUT_STORAGE_ROWMAJOR, v.data()), "gem_get_layer");
        return v;
    }
    std::vector<int> colorLayer(int which) const
    {
        std::vector<int> v(static_cast<size_t>(length_) * length_);
        if (which < GEM_LAYER_COLOR_R || which > GEM_LAYER_COLOR_B) throw Error(GEM_ERR_INVALID, "not a colour layer");
        check(gem_get_layer(h_, which, GEM_LAYOUT_STORAGE_ROWMAJOR, v.data()), "gem_get_layer");
        return v;
    }
    // grid_map::Matrix memory (Eigen column-major, NaN for empty cells): memcpy into GridMap::get(layer).data()
    std::vector<float> gridMapLayer(int which) const
    {
        std::vector<float> v(static_cast<size_t>(length_) * length_);
        check(gem_get_layer(h_, which, GEM_LAYOUT_GRIDMAP_COLMAJOR_NAN, v.data()), "gem_get_layer");
        return v;
    }
    void synchronize() { check(gem_synchronize(h_), "gem_synchronize"); }
    // arenas for the largest pass to come (points per call, sweeps per call): no allocation inside the stream of frames afterwards
    void reserve(long long maxPoints, int maxSweeps = 1, bool withColours = false)
    { check(gem_reserve(h_, maxPoints, maxSweeps, withColours ? 1 : 0), "gem_reserve"); }
    // device inputs produced on another stream: everything enqueued from now on waits for this hipEvent_t (gem_wait_event)
    void waitEvent(void* hipEvent) { check(gem_wait_event(h_, hipEvent), "gem_wait_event"); }

    // ElevationMap::show's cell loop (ElevationMap.cpp:85-149) on the resident layers: visualMap_'s nine layers
    // (visualMapLayers() order, grid_map::Matrix memory, NaN for cells without elevation / traversability), the coloured
    // point cloud in grid_map's iteration order and the orthomosaic.  Geometry = visualMap_'s (doubles, ElevationMapping.cpp:178);
    // 0 / 0 / nullptr: length * resolution, the map's resolution and centre.
    struct Shown {
        std::vector<float> visual;            // 9 x length^2
        std::vector<float> pointsXYZ;         // n x 3
        std::vector<unsigned char> pointsRGB; // n x 3
        std::vector<unsigned char> imageBGR;  // length x length x 3
        int count = 0;
    };
    Shown show(double mapLength = 0.0, double resolution = 0.0, const double position[2] = nullptr) const
    {
        const size_t cells = static_cast<size_t>(length_) * length_;
        Shown s;
        s.visual.resize(9 * cells); s.pointsXYZ.resize(3 * cells); s.pointsRGB.resize(3 * cells); s.imageBGR.resize(3 * cells);
        check(gem_show(h_, mapLength, resolution, position, s.visual.data(), s.pointsXYZ.data(), s.pointsRGB.data(), &s.count, s.imageBGR.data()), "gem_show");
        s.pointsXYZ.resize(3 * static_cast<size_t>(s.count)); s.pointsRGB.resize(3 * static_cast<size_t>(s.count));
        return s;
    }

    // The colourisation loop of ElevationMapping::Callback (ElevationMapping.cpp:321-381) on the cloud as the node holds it:
    // P_lidar2img = Tcamera (3x4) * TLidar (4x4), every point takes the BGR pixel it lands on (b, g, r fields), draws its
    // radius-1 circle for the later points, and points outside the image get b = g = r = 0 and intensity 0.
    // image = cv::Mat::data of the BGR8 image, step = cv::Mat::step (0: width * 3); the image is not modified.
    static std::array<double, 12> lidarToImage(const std::array<double, 12>& Tcamera, const Mat4& TLidar)
    {
        std::array<double, 12> P{};
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) {
                double acc = Tcamera[4 * r] * TLidar[c];
                for (int k = 1; k < 4; ++k) acc = acc + Tcamera[4 * r + k] * TLidar[4 * k + c];
                P[4 * r + c] = acc;
            }
        return P;
    }
    void colorize(const std::array<double, 12>& lidar2img, int width, int height, const unsigned char* image, size_t step,
                  PointXYZRGBICT* cloud, int n) const
    {
        gem_camera cam{};
        for (int k = 0; k < 12; ++k) cam.lidar_to_image[k] = lidar2img[k];
        cam.width = width; cam.height = height;
        std::vector<float> xyzi(4 * static_cast<size_t>(n));
        std::vector<std::uint32_t> rgb(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i) { xyzi[4 * i] = cloud[i].x; xyzi[4 * i + 1] = cloud[i].y; xyzi[4 * i + 2] = cloud[i].z; xyzi[4 * i + 3] = cloud[i].intensity; }
        check(gem_colorize(h_, &cam, n, xyzi.data(), image, step, rgb.data()), "gem_colorize");
        for (int i = 0; i < n; ++i) {
            cloud[i].r = static_cast<std::uint8_t>(rgb[i] >> 16); cloud[i].g = static_cast<std::uint8_t>(rgb[i] >> 8); cloud[i].b = static_cast<std::uint8_t>(rgb[i]);
            cloud[i].intensity = xyzi[4 * i + 3];
        }
    }

    void check(int rc, const char* what) const { if (rc != GEM_OK) throw Error(rc, std::string(what) + ": " + gem_last_error(h_)); }

private:
    gem_handle* h_ = nullptr;
    int length_ = 0;
    float resolution_ = 0.f;
};

inline bool SensorProcessorBase::process(ElevationMap& map, const PointXYZRGBICT* cloud, int n,
                                         int* point_colorR, int* point_colorG, int* point_colorB, int* point_index,
                                         float* point_intensity, float* point_height, float* point_var)
{
    // AoS -> SoA split of GPUPointCloudprocess (SensorProcessorBase.cpp:160-169)
    std::vector<float> x(n), y(n), z(n);
    for (int i = 0; i < n; ++i) {
        x[i] = cloud[i].x; y[i] = cloud[i].y; z[i] = cloud[i].z;
        point_colorR[i] = cloud[i].r; point_colorG[i] = cloud[i].g; point_colorB[i] = cloud[i].b;
        point_intensity[i] = cloud[i].intensity;
    }
    const gem_frame_params p = frameParams();
    const int rc = gem_process_points(map.handle(), &p, n, x.data(), y.data(), z.data(), nullptr, 0,
                                      point_index, point_var, nullptr, nullptr, point_height);     // SensorProcessorBase.cpp:208
    return rc == GEM_OK;
}

// ---------------------------------------------------------------------------------------------
// RobotMotionMapUpdater (RobotMotionMapUpdater.cpp:42-145), plain arrays instead of kindr types.
// Rotations follow kindr-1.x conventions: C_IB maps base to inertial coordinates.
// ---------------------------------------------------------------------------------------------
class RobotMotionMapUpdater {
public:
    explicit RobotMotionMapUpdater(double covarianceScale = 1.0) : covarianceScale_(covarianceScale)
    {
        previousReducedCovariance_.fill(0.0);
        previousRotation_ = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        previousPosition_ = {0, 0, 0};
    }

    // returns the variance increment handed to Mapvar_update (RobotMotionMapUpdater.cpp:80-81)
    float update(ElevationMap& map, const Vec3& position, const Mat3& R_IB, const std::array<double, 36>& poseCovariance,
                 const Mat3& mapRotation = {1, 0, 0, 0, 1, 0, 0, 0, 1})
    {
        const float u = compute(position, R_IB, poseCovariance, mapRotation);
        map.update(u);
        return u;
    }

    float compute(const Vec3& position, const Mat3& R, const std::array<double, 36>& poseCovariance, const Mat3& mapRotation)
    {
        // computeReducedCovariance (:92-109)
        const double yaw = std::atan2(R[3], R[0]);
        const double pitch = std::atan2(-R[6], std::sqrt(R[0] * R[0] + R[3] * R[3]));
        const double tp = std::tan(pitch);
        double J[4][6] = {};
        J[0][0] = J[1][1] = J[2][2] = 1.0;
        J[3][3] = std::cos(yaw) * tp; J[3][4] = std::sin(yaw) * tp; J[3][5] = 1.0;
        double reduced[4][4] = {};
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) s += J[i][a] * covarianceScale_ * poseCovariance[a * 6 + b] * J[j][b];
            reduced[i][j] = s;
        }
        // computeRelativeCovariance (:111-145)
        double c = 0.5 * (R[0] + R[4] + R[8] - 1.0); c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
        const double angle = std::acos(c);
        const double wz = 0.5 * (R[3] - R[1]);
        const double rz = angle < 1e-12 ? wz : wz * angle / std::sin(angle);
        const double Rt[3][3] = {{std::cos(rz), -std::sin(rz), 0}, {std::sin(rz), std::cos(rz), 0}, {0, 0, 1}};
        const double dp[3] = {position[0] - previousPosition_[0], position[1] - previousPosition_[1], position[2] - previousPosition_[2]};
        double v[3];
        for (int i = 0; i < 3; ++i) v[i] = previousRotation_[i] * dp[0] + previousRotation_[3 + i] * dp[1] + previousRotation_[6 + i] * dp[2];
        double Rv[3];
        for (int i = 0; i < 3; ++i) Rv[i] = Rt[i][0] * v[0] + Rt[i][1] * v[1] + Rt[i][2] * v[2];
        double F[4][4] = {{1, 0, 0, -Rv[1]}, {0, 1, 0, Rv[0]}, {0, 0, 1, 0}, {0, 0, 0, 1}};
        double G[4][4] = {}, Gt[4][4] = {};
        G[3][3] = Gt[3][3] = 1.0;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { G[i][j] = Rt[j][i]; Gt[i][j] = Rt[i][j]; }
        double D[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += F[i][a] * previousReducedCovariance_[a * 4 + b] * F[j][b];
            D[i][j] = reduced[i][j] - s;
        }
        double rel[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) s += G[i][a] * D[a][b] * Gt[b][j];
            rel[i][j] = s;
        }
        // update (:58-80): R_B_M = R_I_B^T R_I_M;  J_r = -R_B_M^T;  var = (J_r Sigma J_r^T)_zz
        const Mat3 RBM = mul(transposed(R), mapRotation);
        double Jr[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Jr[i][j] = -RBM[j * 3 + i];
        double out = 0.0;
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) out += Jr[2][a] * rel[a][b] * Jr[2][b];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) previousReducedCovariance_[i * 4 + j] = reduced[i][j];
        previousPosition_ = position; previousRotation_ = R;
        return static_cast<float>(out);
    }

private:
    double covarianceScale_;
    std::array<double, 16> previousReducedCovariance_;
    Vec3 previousPosition_;
    Mat3 previousRotation_;
};

} // namespace gem
