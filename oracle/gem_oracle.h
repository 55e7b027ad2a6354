/*
 * gem_oracle.h -- CPU ORACLE for the GEM point-cloud -> elevation-grid hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (gem_amd/, include/) never
 * links, imports or calls anything in oracle/.
 *
 * It is a plain-C, single-threaded restatement of the semantics of the reference's CUDA
 * kernels (there is no CPU ElevationMap::add in the reference tree).
 * Citations use GPU = elevation_mapping/elevation_mapping/cuda/gpu_process.cu,
 * SPB.cpp = .../src/sensor_processors/SensorProcessorBase.cpp, RMU.cpp = .../src/RobotMotionMapUpdater.cpp.
 *
 * PINNING: the reference ships no tests, golden vectors or fixtures, and its own build (nvcc, Eigen,
 * ROS, PCL, kindr) is not possible in this environment.  The oracle is pinned
 *   (1) against the reference's OWN gpu_process.cu compiled for the CPU (oracle/ref_build/ -> oracle/_ref/libgem_ref.so:
 *       kernels run sequentially over their grids, CUDA runtime + Eigen are stand-ins, nothing else of its text is
 *       touched) -- tests/test_reference_compiled.py: indices, variances, fused layers, Move, loop-closure shifts bit
 *       for bit; slope / traversability up to the C library's float trigonometry;
 *   (2) by the hand-derived known-answer tests in tests/test_oracle_kat.py;
 *   (3) by the golden fixtures it generated itself (tests/golden/, scripts committed).
 * Not covered by (1): Eigen's internal evaluation order (restated identically in the stand-in and here), nvcc's FMA
 * contraction (a build-flag effect; source-level arithmetic is what is replayed).
 *   (1b) the structured-light / stereo / perfect sensor models against the reference's OWN *SensorProcessor.cpp, and the
 *       motion updater against its RobotMotionMapUpdater.cpp, each compiled where it lies against stand-ins for Eigen / kindr /
 *       PCL / ROS / TF (oracle/ref_build/sensors, oracle/ref_build/motion -> oracle/_ref/libgem_ref_sensors.so,
 *       libgem_ref_motion.so): tests/test_reference_sensor_models.py, tests/test_motion_update.py.
 *
 * Build with -ffp-contract=off: every float product and sum below is individually rounded.
 */
#ifndef GEM_ORACLE_H
#define GEM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* sensor noise models: laser = the only one on the reference's GPU path (GPU:403-425);
 * the other three exist as CPU computeVariances() (SL.cpp:121-153, Stereo.cpp:72-104, Perfect.cpp:74-102). */
enum { GEMO_MODEL_LASER = 0, GEMO_MODEL_STRUCTURED_LIGHT = 1, GEMO_MODEL_STEREO = 2, GEMO_MODEL_PERFECT = 3 };

typedef struct gemo_frame {
    float  T[16];                 /* sensor->map homogeneous transform, row-major (SPB.cpp:171-179)           */
    double lower, upper;          /* absolute height window, doubles (SPB.cpp:183-184, GPU:50-51)             */
    int    sensor_model;          /* GEMO_MODEL_*                                                             */
    double sp[8];                 /* model parameters, DOUBLE like sensorParameters_ (std::map<string,double>):
                                     laser: min_r, beam_a, beam_c (cast to float, SPB.cpp:286-288) |
                                     SL: a,b,c,d,e,lateral | stereo: p1..p5, lateral, depth_to_disparity     */
    float  sensor_jacobian[3];    /* J_s (SPB.cpp:275)                                                        */
    float  rotation_variance[9];  /* Sigma_q row-major (zero in the reference, SPB.cpp:202-204)               */
    float  C_SB_T[9];             /* row-major (SPB.cpp:283)                                                  */
    float  P_mul_C_BM_T[3];       /* (SPB.cpp:281-282)                                                        */
    float  B_r_BS_skew[9];        /* row-major (SPB.cpp:284)                                                  */
    int    filter_on;             /* 1 = sensor-frame reject filter of GPU:393, 0 = off                        */
    float  filter_box_x, filter_box_y, filter_band_y, filter_plane_y;  /* reference: 1.5, 1.5, 1.0, 0.0      */
    int    original_width;        /* stereo only: image width used by getI/getJ (Stereo.cpp:108-116)          */
} gemo_frame;

typedef struct gemo_map {
    int   L;                      /* cells per side (GPU:35) */
    float res;                    /* GPU:36 */
    float mahal;                  /* literal 5 in the reference (GPU:504) */
    float var_floor;              /* literal 0.0001 in the reference (GPU:500,533) */
    float *elevation, *variance, *intensity, *traver, *lowest;   /* GPU:20-24 */
    int   *colorR, *colorG, *colorB;                             /* GPU:26-28 */
    float center[2];              /* GPU:30 */
    int   start[2];               /* GPU:31 */
    float sensor_z;               /* GPU:33 */
    float obstacle_threshold;     /* GPU:37; 0.7 unless gemo_set_obstacle_threshold() (elevation_map.yaml)     */
} gemo_map;

gemo_map* gemo_create(int length, float resolution, float mahalanobis, float var_floor);
void      gemo_destroy(gemo_map* m);

/* GPU:1004-1083 (Move).  Returns the number of clear launches the reference would have issued. */
int  gemo_move(gemo_map* m, const float pos[3], float out_center[2], int out_start[2], float out_shift[2]);

/* GPU:309-330 / GPU:332-358.  Return -1 when outside. */
int  gemo_points_to_index(const gemo_map* m, float px, float py);
int  gemo_points_to_map_index(const gemo_map* m, float px, float py);

/* GPU:384-455 (G_pointsprocess), including its map_lowest side effect (GPU:432-439) in the schedule in which point i
 * finishes before point i + 1 starts: lowest[g] = min(lowest[g], h); if (h == lowest[g]) lowest[g] += 3 * var.
 * (On a GPU the read-modify-write races; this schedule is what the reference's code produces when its grid is run
 * sequentially, and it is the one libgem_hip reproduces.)  map_lowest is indexed by the GEOGRAPHIC cell, not by the
 * circular-buffer cell (GPU:430 PointsToIndex).  x,y,z are overwritten with
 * -1 for rejected points exactly as the reference does to its device copies (GPU:443-446).
 * orig_index may be NULL (stereo model only).  Returns number of accepted points. */
int  gemo_process_points(gemo_map* m, const gemo_frame* f, int n,
                         float* x, float* y, float* z, const int* orig_index,
                         int* map_index, float* var, float* x_ts, float* y_ts, float* z_ts);

/* GPU:477-537 (G_fuse), restated as ONE sequential loop over points (each cell still sees its
 * points in ascending i, which is all the per-cell scan of the reference guarantees).
 * R,G,B,intensity may be NULL (treated as all-zero => colour/intensity layers untouched). */
void gemo_fuse(gemo_map* m, int n, const int* index, const int* R, const int* G, const int* B,
               const float* intensity, const float* height, const float* var);

/* The literal O(L^2 * N) form of G_fuse, one "thread" per cell: used only to prove the O(N)
 * restatement equivalent on small cases. */
void gemo_fuse_literal(gemo_map* m, int n, const int* index, const int* R, const int* G, const int* B,
                       const float* intensity, const float* height, const float* var);

/* GPU:540-547 (G_Mapvar_update) */
void gemo_mapvar_update(gemo_map* m, float var_update);

/* process_points + fuse on an interleaved XYZI cloud (the fused path the product calls gem_add).
 * rgb: packed 0x00RRGGBB per point or NULL.  Returns accepted count; counts[0]=accepted,
 * counts[1]=distinct touched cells (for the B_alg figure of SURVEY.md 8d). */
int  gemo_add(gemo_map* m, const gemo_frame* f, int n, const float* xyzi, const unsigned* rgb,
              const int* orig_index, long long counts[2]);

/* The all-core form (gem_oracle_mt.c): for every sweep, Mapvar_update(var_updates[s]) (if given) then add(frames[s],
 * cloud s = xyzi + 4 * offsets[s]), with the CELLS partitioned into row strips, one per thread, each thread scanning the sweep's
 * index array in input order.  Bit-identical to the sequential calls (no colours).  Returns accepted points, -1 on failure. */
long long gemo_add_batch_mt(gemo_map* m, int n_sweeps, const gemo_frame* frames, const float* xyzi, const long long* offsets,
                            const float* var_updates, int nthreads);

/* GPU:1215-1233 / 1235-1254 with G_update_mapheight GPU:1195-1202: loop-closure re-anchoring of the map. */
void gemo_map_optmove(gemo_map* m, const float opt_p[2], float height_update, float out_aligned[2]);
void gemo_map_closeloop(gemo_map* m, const float update_position[2], float height_update);

/* GPU:549-670 (G_Mapfeature) + GPU:66-187 (computerEigenvalue): 5x5-neighbourhood plane fit per cell ->
 * roughness, slope and traversability; updates m->traver like map_traver.  Outputs may be NULL. */
void gemo_map_feature(gemo_map* m, float* rough, float* slope, float* traver_out);

/* RMU.cpp:42-145 restated with plain arrays: returns the scalar var_update handed to Mapvar_update.
 * pose: position[3] + rotation matrix R_IB row-major[9]; cov: 6x6 row-major; state carries the
 * previous pose / previous reduced covariance exactly like the class members. */
typedef struct gemo_motion_state {
    double prev_reduced_cov[16];
    double prev_pos[3];
    double prev_R[9];
    double covariance_scale;
} gemo_motion_state;
void   gemo_motion_init(gemo_motion_state* s, double covariance_scale);
double gemo_motion_update(gemo_motion_state* s, const double pos[3], const double R_IB[9],
                          const double cov6x6[36], const double map_R[9]);

/* EM.cpp:85-149 (ElevationMap::show): the cell loop behind the visualMap_ layers, the point cloud and the orthomosaic.  gem_oracle_show.c */
int gemo_show(const gemo_map* m, const float* rough, const float* slope, double map_length, double resolution, const double map_position[2],
              float* visual, float* points_xyz, unsigned char* points_rgb, unsigned char* image_bgr);

/* the step in front of the path: input colourisation, EMg.cpp:349-381 (gem_oracle_color.c) */
void gemo_lidar_to_image(const double tcamera[12], const double tlidar[16], double out[12]);
int gemo_colorize(const double P[12], int width, int height, unsigned char* image_bgr, size_t stride, int n, float* xyzi, uint32_t* rgb);

/* GPU:1304-1318 (Raytracing): G_Raytracing (GPU:708-891) then G_Clear_maplowest (GPU:232-239).  gem_oracle_raytrace.c */
void gemo_raytracing(gemo_map* m);
void gemo_set_obstacle_threshold(gemo_map* m, float t);

#ifdef __cplusplus
}
#endif
#endif
