// gem_compat_eigen.hpp -- re-creates, on top of libgem_hip.so's C ABI, the C++-linkage free functions
// that the reference's callers forward-declare and link from libgpu.so:
//
//   Init_GPU_elevationmap, Move, Fuse, Map_feature, Raytracing, Map_optmove, Map_closeloop
//        (forward-declared at elevation_mapping/src/ElevationMapping.cpp:44-50)
//   Process_points    (src/sensor_processors/SensorProcessorBase.cpp:34)
//   Mapvar_update     (src/RobotMotionMapUpdater.cpp:18 -- declared `int` there, defined `void` in
//                      gpu_process.cu:1146; the mangled name carries no return type, so one definition serves)
//
// Compile ONE translation unit that includes this header into the catkin package in place of
// cuda/gpu_process.cu: the ROS node itself stays unchanged.  Needs Eigen
// (by-value Matrix4f / Matrix3f / RowVector3f arguments), which exists in the ROS workspace but not in
// this repository's build container -- tests/cpp compiles it against a minimal stand-in.
//
// All nine run on the device: the hot-path symbols, Map_feature (the traversability stage),
// Raytracing (the visibility clean-up, 8f #3) and the loop-closure shifts Map_optmove / Map_closeloop (8f #4).
#pragma once

#include <Eigen/Core>

#include "../gem_hip.h"

#include <cstdio>
#include <cstring>
#include <vector>

namespace gem_compat {

inline gem_handle*& handle() { static gem_handle* h = nullptr; return h; }
inline gem_reject_filter& reject_filter()
{
    static gem_reject_filter f{1, 1.5f, 1.5f, 1.0f, 0.0f};       // gpu_process.cu:393, on like the reference
    return f;
}
inline void report(int rc, const char* what)
{
    // the reference prints CUDA errors to stderr and carries on (gpu_process.cu:987-992, 1127-1132)
    if (rc != GEM_OK) std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, gem_last_error(handle()));
}

} // namespace gem_compat

// gpu_process.cu:940-994
void Init_GPU_elevationmap(int length, float resolution, float h_mahalanobisDistanceThreshold_, float h_obstacle_threshold)
{
    (void)h_mahalanobisDistanceThreshold_;      // uploaded but never read by the reference; G_fuse uses the literal 5 (gpu_process.cu:504)
    if (gem_compat::handle()) { gem_destroy(gem_compat::handle()); gem_compat::handle() = nullptr; }
    gem_map_config cfg{};
    cfg.length = length; cfg.resolution = resolution;
    cfg.mahalanobis_threshold = 5.0f; cfg.variance_floor = 0.0001f;
    cfg.obstacle_threshold = h_obstacle_threshold; cfg.device = -1;
    const int rc = gem_create(&cfg, &gem_compat::handle());
    if (rc != GEM_OK) { std::fprintf(stderr, "Init_GPU_elevationmap failed (%d): %s\n", rc, gem_last_error(nullptr)); return; }
    // the node calls Raytracing every frame (ElevationMapping.cpp:421): keep the lowest scan points (gpu_process.cu:430-439)
    gem_compat::report(gem_set_lowest_tracking(gem_compat::handle(), 1), "gem_set_lowest_tracking");
}

// gpu_process.cu:1004-1083
void Move(float* current_Position, float resolution, int length, float* Central_coordinate, int* Start_indice, float* alignedPositionShift)
{
    (void)resolution; (void)length;
    gem_compat::report(gem_move(gem_compat::handle(), current_Position, Central_coordinate, Start_indice, alignedPositionShift), "Move");
}

// gpu_process.cu:1085-1144
int Process_points(int* map_index, float* point_x, float* point_y, float* point_z, float* point_var,
                   float* point_x_ts, float* point_y_ts, float* point_z_ts, Eigen::Matrix4f transform, int point_num,
                   double relativeLowerThreshold, double relativeUpperThreshold, float min_r, float beam_a, float beam_c,
                   Eigen::RowVector3f sensorJacobian, Eigen::Matrix3f rotationVariance, Eigen::Matrix3f C_SB_transpose,
                   Eigen::RowVector3f P_mul_C_BM_transpose, Eigen::Matrix3f B_r_BS_skew)
{
    gem_frame_params p{};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) p.T[i * 4 + j] = transform(i, j);
    p.lower = relativeLowerThreshold; p.upper = relativeUpperThreshold;
    p.sensor_model = GEM_MODEL_LASER;
    p.sensor_params[0] = min_r; p.sensor_params[1] = beam_a; p.sensor_params[2] = beam_c;
    for (int j = 0; j < 3; ++j) { p.sensor_jacobian[j] = sensorJacobian(0, j); p.P_mul_C_BM_T[j] = P_mul_C_BM_transpose(0, j); }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        p.rotation_variance[i * 3 + j] = rotationVariance(i, j);
        p.C_SB_T[i * 3 + j] = C_SB_transpose(i, j);
        p.B_r_BS_skew[i * 3 + j] = B_r_BS_skew(i, j);
    }
    p.filter = gem_compat::reject_filter();
    // the reference overwrites its DEVICE copies of x,y,z for rejected points, never the host arrays
    // (gpu_process.cu:443-446, no D2H of dev_x/y/z): write_back_xyz = 0
    gem_compat::report(gem_process_points(gem_compat::handle(), &p, point_num, point_x, point_y, point_z, nullptr, 0,
                                          map_index, point_var, point_x_ts, point_y_ts, point_z_ts), "Process_points");
    return 0;
}

// gpu_process.cu:1154-1193.  point_num must be the number of VALID entries: the reference passes the
// pre-cleaning cloud size here (ElevationMapping.cpp:259,280) and so reads an uninitialised tail.
void Fuse(int length, int point_num, int* point_index, int* point_colorR, int* point_colorG, int* point_colorB,
          float* point_intensity, float* point_height, float* point_var)
{
    (void)length;
    gem_compat::report(gem_fuse(gem_compat::handle(), point_num, point_index, point_colorR, point_colorG, point_colorB,
                                point_intensity, point_height, point_var), "Fuse");
}

// gpu_process.cu:1146-1152
void Mapvar_update(int length, float var_update)
{
    (void)length;
    gem_compat::report(gem_mapvar_update(gem_compat::handle(), var_update), "Mapvar_update");
}

// gpu_process.cu:1256-1302 -- traversability stage (G_Mapfeature + the Jacobi eigen-solver run on the device,
// k_map_feature), then the nine layers are copied out as the reference does (gpu_process.cu:1283-1291).
void Map_feature(int length, float* elevation, float* var, int* colorR, int* colorG, int* colorB,
                 float* rough, float* slope, float* traver, float* intensity)
{
    (void)length;
    gem_compat::report(gem_map_feature(gem_compat::handle(), elevation, var, colorR, colorG, colorB, rough, slope, traver, intensity),
                       "Map_feature");
}

// gpu_process.cu:1304-1318 -- visibility clean-up: G_Raytracing, then the lowest scan points are reset (gem_raytracing).
void Raytracing(int length)
{
    (void)length;
    gem_compat::report(gem_raytracing(gem_compat::handle()), "Raytracing");
}

// gpu_process.cu:1215-1233 -- loop-closure re-anchoring: the centre is relabelled to the optimised
// position snapped to the cell lattice and every valid elevation moves by height_update (gem_map_optmove).
void Map_optmove(float* opt_p, float height_update, float resolution, int length, float* opt_alignedPosition)
{
    (void)resolution; (void)length;
    gem_compat::report(gem_map_optmove(gem_compat::handle(), opt_p, height_update, opt_alignedPosition), "Map_optmove");
}

// gpu_process.cu:1235-1254 -- declared by the node (ElevationMapping.cpp:46) but never called.
void Map_closeloop(float* update_position, float height_update, int length, float resolution)
{
    (void)length; (void)resolution;
    gem_compat::report(gem_map_closeloop(gem_compat::handle(), update_position, height_update), "Map_closeloop");
}
