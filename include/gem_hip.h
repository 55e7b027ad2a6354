/*
 * gem_hip.h -- C ABI of libgem_hip.so: the MI355X (gfx950) implementation of GEM's
 * point-cloud -> elevation-grid hot path.  This is the drop-in boundary: plain pointers and
 * sizes, no C++ / Eigen / torch types.  Each entry point cites the reference interface it
 * replaces (GPU = elevation_mapping/elevation_mapping/cuda/gpu_process.cu, EMg.cpp =
 * .../src/ElevationMapping.cpp, SPB.cpp = .../src/sensor_processors/SensorProcessorBase.cpp,
 * RMU.cpp = .../src/RobotMotionMapUpdater.cpp).  The nine C++-linkage symbols the unmodified
 * ROS node links against are re-created on top of this ABI in include/gem/gem_compat_eigen.hpp.
 *
 * Conventions
 *   - every function returns GEM_OK (0) or a negative gem_status; gem_last_error() gives text.
 *   - one gem_handle == one robot-centric map (the reference keeps this as hidden process-global
 *     __device__ state, GPU:20-33); a handle is internally locked, so the reference's
 *     {Process_points -> Fuse} || {Mapvar_update} thread pair (EMg.cpp:391-394) is safe.
 *   - host-pointer entry points copy in/out and are synchronous w.r.t. the caller's buffers;
 *     *_device entry points take device pointers, enqueue on the handle's stream and return.
 *   - there is NO CPU fallback: without a HIP device gem_create fails with GEM_ERR_NO_DEVICE.
 */
#ifndef GEM_HIP_H
#define GEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEM_ABI_VERSION 9

typedef enum gem_status {
    GEM_OK = 0,
    GEM_ERR_INVALID = -1,      /* bad argument */
    GEM_ERR_NO_DEVICE = -2,    /* no HIP device / HIP runtime error at create */
    GEM_ERR_HIP = -3,          /* HIP runtime error (text in gem_last_error) */
    GEM_ERR_NOMEM = -4,
    GEM_ERR_COMM = -5          /* RCCL error */
} gem_status;

/* sensor noise models.  Laser is the only model on the reference's GPU path (GPU:403-425,
 * SPB.cpp:286-288); the others are the reference's CPU computeVariances() bodies
 * (StructuredLightSensorProcessor.cpp:121-153, StereoSensorProcessor.cpp:72-104,
 * PerfectSensorProcessor.cpp:74-102). */
enum { GEM_MODEL_LASER = 0, GEM_MODEL_STRUCTURED_LIGHT = 1, GEM_MODEL_STEREO = 2, GEM_MODEL_PERFECT = 3 };

/* map layers (device layout GPU:20-28; GridMap names EM.cpp:43-44) */
enum {
    GEM_LAYER_ELEVATION = 0, GEM_LAYER_VARIANCE = 1, GEM_LAYER_INTENSITY = 2, GEM_LAYER_TRAVER = 3,
    GEM_LAYER_LOWEST = 4, GEM_LAYER_COLOR_R = 5, GEM_LAYER_COLOR_G = 6, GEM_LAYER_COLOR_B = 7,
    GEM_LAYER_ROUGH = 8, GEM_LAYER_SLOPE = 9,      /* outputs of gem_map_feature (visualMap_ layers "rough", "slope", EM.cpp:44) */
    GEM_LAYER_COUNT = 10
};
/* GEM_LAYER_LOWEST is the reference's map_lowest: indexed by the GEOGRAPHIC cell [gx * L + gy] (GPU:430, 676-679), NOT by the
 * circular-buffer cell like the other layers, and not shifted by gem_move; the GRIDMAP layout does not apply to it. */
/* layouts for gem_get_layer / gem_set_layer */
enum {
    GEM_LAYOUT_STORAGE_ROWMAJOR = 0,   /* the reference's flat [storage_x * L + storage_y] arrays (EM.cpp:98-111)   */
    GEM_LAYOUT_GRIDMAP_COLMAJOR_NAN = 1 /* grid_map::Matrix (Eigen column-major, buffer order), NaN for empty cells */
};

typedef struct gem_map_config {
    int   length;                  /* cells per side, = length_in_x / resolution (EMg.cpp:195)                 */
    float resolution;              /* metres per cell (GPU:36)                                                 */
    float mahalanobis_threshold;   /* the reference uploads one (GPU:977) but uses the literal 5 (GPU:504): pass 5 */
    float variance_floor;          /* literal 0.0001 in the reference (GPU:500,533)                            */
    float obstacle_threshold;      /* GPU:940 4th argument: cells with traversability below it are ray-traced (GPU:712)     */
    int   strip_row0, strip_rows;  /* storage-row strip this handle owns (multi-GPU tiling); 0,0 = whole map   */
    int   device;                  /* HIP device ordinal; -1 = current device                                  */
} gem_map_config;

/* the hard-coded sensor-frame reject filter of GPU:393, parameterised; defaults 1.5,1.5,1.0,0.0 */
typedef struct gem_reject_filter {
    int   enabled;
    float box_x, box_y;            /* reject |x|<box_x && |y|<box_y */
    float band_y;                  /* reject |y|<band_y             */
    float plane_y;                 /* reject y>plane_y              */
} gem_reject_filter;

/* per-frame constants: exactly what SensorProcessorBase::GPUPointCloudprocess hands to
 * Process_points (SPB.cpp:171-208): transform, height window, model parameters, Jacobian pieces. */
typedef struct gem_frame_params {
    float  T[16];                  /* sensor->map, row-major (Eigen::Matrix4f Transform, SPB.cpp:175-179)      */
    double lower, upper;           /* relativeLower/UpperThreshold (SPB.cpp:183-184)                          */
    int    sensor_model;           /* GEM_MODEL_*                                                              */
    double sensor_params[8];       /* laser: min_radius, beam_angle, beam_constant (SPB.cpp:286-288);
                                      structured light: normal_factor_a..e, lateral_factor;
                                      stereo: p_1..p_5, lateral_factor, depth_to_disparity_factor            */
    float  sensor_jacobian[3];     /* SPB.cpp:275 */
    float  rotation_variance[9];   /* row-major; zero in the reference (SPB.cpp:202-204) */
    float  C_SB_T[9];              /* row-major, SPB.cpp:283 */
    float  P_mul_C_BM_T[3];        /* SPB.cpp:281-282 */
    float  B_r_BS_skew[9];         /* row-major, SPB.cpp:284 */
    gem_reject_filter filter;
    int    original_width;         /* stereo: image width for getI/getJ (StereoSensorProcessor.cpp:108-116)  */
} gem_frame_params;

typedef struct gem_handle gem_handle;

/* counters of the most recent add/fuse call (read back lazily; forces a stream sync) */
typedef struct gem_stats {
    long long points_in;
    long long points_binned;       /* accepted AND inside the map (and inside this handle's strip)             */
    long long cells_touched;       /* distinct cells that received >= 1 point                                  */
    float     ms_bin, ms_fuse;     /* accumulated kernel time of the two pipeline kernels since reset          */
    int       launches_bin, launches_fuse;
    float     ms_frame;            /* ... and of k_frame (fuse of the previous sweep + bin of the new one)     */
    int       launches_frame;
    float     ms_sort[6];          /* the sorted pipeline of big passes: count1, scan1, scatter1, count2, scan2, scatter2 */
    int       launches_sort;       /* passes through it                                                        */
    float     ms_walk;             /* ... and its k_fuse_walk                                                  */
    int       launches_walk;
} gem_stats;

/* ---- lifecycle: replaces Init_GPU_elevationmap (GPU:940-994, called EMg.cpp:199) --------------- */
int  gem_create(const gem_map_config* cfg, gem_handle** out);
void gem_destroy(gem_handle* h);
const char* gem_last_error(const gem_handle* h);      /* h may be NULL: error of the last failed gem_create */
int  gem_abi_version(void);

/* stream the handle enqueues on (a hipStream_t passed as void*); NULL = the handle's own stream */
int  gem_set_stream(gem_handle* h, void* hip_stream);
int  gem_synchronize(gem_handle* h);
/* Stream-ordered device inputs: a caller whose cloud is produced on ANOTHER stream records a hipEvent_t there and hands it over
 * before the *_device call; all work the handle enqueues afterwards (on either of its internal streams) waits for it on the
 * device -- no host synchronisation.  Without it, device buffers must be complete when a *_device entry is called.          */
int  gem_wait_event(gem_handle* h, void* hip_event);

/* ---- Move (GPU:1004-1083, called EMg.cpp:1032) -------------------------------------------------- */
int  gem_move(gem_handle* h, const float position[3], float out_center[2], int out_start[2],
              float out_aligned_shift[2]);
int  gem_get_pose(gem_handle* h, float out_center[2], int out_start[2]);

/* ---- Process_points (GPU:1085-1144, called SPB.cpp:208): host SoA arrays in, host arrays out.
 *      x,y,z are overwritten with -1 for rejected points like the reference's device copies
 *      (GPU:443-446) only if write_back_xyz != 0.  Any output pointer may be NULL.              */
int  gem_process_points(gem_handle* h, const gem_frame_params* p, int n,
                        float* x, float* y, float* z, const int* orig_index, int write_back_xyz,
                        int* map_index, float* var, float* x_ts, float* y_ts, float* z_ts);

/* ---- Fuse (GPU:1154-1193, called EMg.cpp:280): host arrays in.  R,G,B,intensity may be NULL. -- */
int  gem_fuse(gem_handle* h, int n, const int* index, const int* R, const int* G, const int* B,
              const float* intensity, const float* height, const float* var);

/* ---- the fused path: SensorProcessorBase::process + Fuse (EMg.cpp:254-283) in one call on an
 *      interleaved XYZI cloud (16 B / point); rgb = packed 0x00RRGGBB per point or NULL.
 *      Nothing returns to the host.  This is the ElevationMap::add-shaped entry.                  */
int  gem_add(gem_handle* h, const gem_frame_params* p, int n, const float* xyzi,
             const uint32_t* rgb, const int* orig_index);
/*      gem_add_device takes device pointers and only enqueues.  The buffers must be complete when the call is made -- or
 *      their producer's event must have been passed to gem_wait_event -- and stay untouched until gem_synchronize (or
 *      any call that returns map data).  The order of operations the caller issues is the order the map sees;
 *      underneath, a stream of single colourless sweeps runs as ONE launch per frame (binning of the new cloud next
 *      to the fusion of the previous frame's records), the newest frame's fusion being launched by the next call
 *      that needs it; a cloud big enough for the sorted pipeline (a depth image) likewise leaves its last kernel, the walk
 *      over its sorted records, to the next call -- which can then launch it without a stream wait.           */
int  gem_add_device(gem_handle* h, const gem_frame_params* p, int n, const void* d_xyzi,
                    const void* d_rgb, const void* d_orig_index);

/* ---- AoS ingest: the cloud as an array of point structs, e.g. pcl::PointCloud<PointXYZRGBICT>::points
 *      (PointXYZRGBICT.hpp:28-46: 32-byte structs, x y z at 0 4 8, rgb at 16 -- PCL's b g r a bytes --, intensity at 24).
 *      Replaces the host loop that pulls the fields into seven arrays (SPB.cpp:160-169) and the packing that gem_add
 *      expects: the structs are copied to the device as they are and unpacked there.  Offsets are byte offsets of 4-byte
 *      fields inside a struct of point_step bytes; off_intensity / off_rgb may be -1 (intensity 0 / no colours).
 *      Same result as gem_add on the unpacked arrays.                                                            */
int  gem_add_aos(gem_handle* h, const gem_frame_params* p, int n, const void* points_host, int point_step,
                 int off_x, int off_y, int off_z, int off_intensity, int off_rgb);

/* ---- batched sweeps (BASELINE config 4): for s in 0..n_sweeps-1:
 *        Mapvar_update(var_updates[s]) ; add(params[s], cloud s)
 *      with the map pose fixed for the batch.  Clouds are device-resident, concatenated:
 *      cloud s = d_xyzi + 16*offsets[s], offsets has n_sweeps+1 entries.                           */
int  gem_add_batch_device(gem_handle* h, int n_sweeps, const gem_frame_params* params,
                          const void* d_xyzi, const long long* offsets, const float* var_updates);

/* ... the same from HOST memory (the reference's caller owns host arrays, EMg.cpp:260-283, GPU:1096-1141):
 *      sweep s = clouds_host[s][0 .. 4 * counts[s]) floats (XYZI points).  The sweeps are copied into the handle's own
 *      device arena one behind the other -- the staging copy of a sweep beside the DMA of the one before -- and fused by
 *      one batched pass; the caller's arrays have been read when the call returns.                                */
int  gem_add_batch(gem_handle* h, int n_sweeps, const gem_frame_params* params, const float* const* clouds_host,
                   const int* counts, const float* var_updates);

/* ---- Mapvar_update (GPU:1146-1152, called RMU.cpp:81) ------------------------------------------ */
int  gem_mapvar_update(gem_handle* h, float var_update);

/* ---- layer access (what the dead G_get_mapinfo / G_set_mapinfo hinted at, GPU:457-475; feeds
 *      ElevationMap::show, EM.cpp:85-149).  dst/src hold L*L 4-byte elements (float, or int32 for
 *      the colour layers in STORAGE_ROWMAJOR; float in GRIDMAP layout).                            */
int  gem_get_layer(gem_handle* h, int layer, int layout, void* dst_host);
int  gem_set_layer(gem_handle* h, int layer, const void* src_host);     /* STORAGE_ROWMAJOR only */
int  gem_layer_device_ptr(gem_handle* h, int layer, void** out_device_ptr);

/* ---- traversability stage that follows the fusion every frame: Map_feature (GPU:1256-1302, kernel
 *      G_Mapfeature GPU:549-670 with the Jacobi eigen-solver GPU:66-187; called EMg.cpp:410).  Computes the
 *      ROUGH, SLOPE and TRAVER layers on the device from ELEVATION.  Each host pointer may be NULL (nothing is
 *      copied for it); with all NULL the call only enqueues the kernel.  Like the reference, the nine arrays
 *      hold L*L elements in STORAGE_ROWMAJOR order.  Cells without elevation report rough = slope = 0 and
 *      keep their stored traversability (the reference leaves its output arrays uninitialised there).       */
int  gem_map_feature(gem_handle* h, float* elevation, float* variance, int* colorR, int* colorG, int* colorB,
                     float* rough, float* slope, float* traver, float* intensity);

/* ---- the feed of ElevationMap::show (EM.cpp:85-149, called EMg.cpp:413 right after Map_feature): the reference copies
 *      nine L*L arrays to the host and loops over all cells there.  gem_show does that loop on the resident layers:
 *        visual      9 x L*L floats, grid_map::Matrix layout (Eigen column-major by BUFFER index) of visualMap_'s layers in the order
 *                    elevation, variance, rough, slope, traver, color_r, color_g, color_b, intensity (EM.cpp:44); NaN where the cell
 *                    has no elevation or no traversability (EM.cpp:89, 101)
 *        points_*    the coloured cloud of EM.cpp:113-122, one point per kept cell in grid_map's iteration order, compacted on the
 *                    device: xyz n x 3 floats (x, y from grid_map's getPositionFromIndex in double, z = elevation), rgb n x 3 bytes;
 *                    both arrays must hold L*L points; *out_count = n
 *        image_bgr   the L x L x 3 orthomosaic of EM.cpp:87, 124-126 (unwrapped row, column; b, g, r)
 *      map_length / resolution / position are visualMap_'s geometry (doubles, EMg.cpp:178; position = what Move returned,
 *      EM.cpp:172-177); pass 0 / 0 / NULL to use length * resolution, the handle's resolution and centre.  Any output may be NULL.
 *      Run gem_map_feature first: rough / slope / traver are its layers.                                                      */
int  gem_show(gem_handle* h, double map_length, double resolution, const double position[2],
              float* visual, float* points_xyz, unsigned char* points_rgb, int* out_count, unsigned char* image_bgr);

/* ---- input colourisation (EMg.cpp:349-381, the loop of ElevationMapping::Callback in front of the path): every
 *      point is projected into the camera image with P_lidar2img = T.camera (3x4) * T.lidar (4x4) (doubles, EMg.cpp:342-345;
 *      row-major here), takes the BGR pixel it lands on, and draws cv::circle(img, pixel, 1, that colour) into the image the
 *      later points sample -- gem_colorize reproduces that order dependence (the latest earlier point on a 4-neighbour pixel
 *      hands its colour on).  Points that fall outside the image (or behind the camera) get colour 0 and INTENSITY 0
 *      (EMg.cpp:372-377): xyzi is updated in place.  rgb[i] = 0x00RRGGBB, the `rgb` input of gem_add*.  row_stride = bytes per
 *      image row (0 = width * 3).  The image itself is left as it was (the reference's drawn-on copy is discarded, EMg.cpp:316). */
typedef struct gem_camera {
    double lidar_to_image[12];   /* row-major 3 x 4 */
    int    width, height;
} gem_camera;
int  gem_colorize(gem_handle* h, const gem_camera* cam, int n, float* xyzi, const unsigned char* image_bgr, size_t row_stride, uint32_t* rgb);
int  gem_colorize_device(gem_handle* h, const gem_camera* cam, int n, float* d_xyzi, const unsigned char* d_image_bgr, size_t row_stride,
                         uint32_t* d_rgb);           /* device pointers; only enqueues on the handle's stream */

/* ---- loop-closure re-anchoring: Map_optmove (GPU:1215-1233, called EMg.cpp:1020) relabels the
 *      map centre to opt_position snapped to the old centre's cell lattice (the circular buffer is not shifted,
 *      nothing is cleared) and adds height_update to every valid elevation (G_update_mapheight, GPU:1195-1202);
 *      Map_closeloop (GPU:1235-1254) moves the centre by the aligned shift instead.                         */
int  gem_map_optmove(gem_handle* h, const float opt_position[2], float height_update, float out_aligned_position[2]);
int  gem_map_closeloop(gem_handle* h, const float update_position[2], float height_update);

/* ---- visibility clean-up: Raytracing (GPU:1304-1318, called EMg.cpp:421) = G_Raytracing (GPU:708-891):
 *      every cell with traversability below obstacle_threshold walks away from the map centre along the centre->cell
 *      ray; crossed cells with a lowest scan point this frame bound its height by the sensor's line of sight, and the
 *      cell is deleted (elevation = -10) if elevation - 3 sigma exceeds the tightest bound -- then G_Clear_maplowest
 *      (GPU:232-239) resets the LOWEST layer to 10.
 *      The LOWEST layer is the side output of G_pointsprocess (GPU:430-439: lowest = min(lowest, h); if (h == lowest)
 *      lowest += 3 * var, per GEOGRAPHIC cell).  It is maintained -- in input order per cell, the result of running the
 *      reference's grid sequentially -- by the fuse kernels of gem_fuse / gem_add* ONLY while lowest tracking is on
 *      (default off: the LiDAR hot path is not slowed down); gem_process_points alone does not touch it.              */
int  gem_set_lowest_tracking(gem_handle* h, int enabled);
int  gem_raytracing(gem_handle* h);

/* ---- statistics / timing (bench harness) --------------------------------------------------------- */
int  gem_set_timing(gem_handle* h, int enabled);      /* record hipEvents around each pipeline kernel */
int  gem_set_counting(gem_handle* h, int enabled);    /* count binned points / touched cells on device */
int  gem_get_stats(gem_handle* h, gem_stats* out, int reset);

/* Pre-size the handle's device arenas for the largest pass that is going to come: max_points points in at most max_sweeps sweeps
 * per call (1 for gem_add / gem_add_device / gem_fuse), with or without colours.  The arenas only ever grow, but growing in the
 * middle of a stream -- the first bigger cloud after smaller ones -- waits for everything in flight and re-allocates; after
 * gem_reserve no pass within these bounds allocates, whichever pipeline it takes (the sorted forms from their thresholds on, the
 * tile pipeline below them; the sweeps of a batch below the sorted threshold are taken to be at most twice their mean length).
 * On a handle that joined a communicator with gem_comm_init_tiles the bounds are those of a gem_add_sharded_device STEP -- the
 * GLOBAL points and sweeps: the shard's sort (its W-th of the points), both sets of receive buffers (no strip gets more records
 * than the step has points) and the staging tables are sized.  Synchronous; call it once after gem_create / gem_comm_init*.    */
int  gem_reserve(gem_handle* h, long long max_points, int max_sweeps, int with_colours);

/* ---- multi-GPU: RCCL all-gather of the fused strips over xGMI ------------------------
 *      One process per GPU, one handle per process.  gem_comm_init splits the map into row strips in STORAGE coordinates
 *      (rank r owns rows [L r / W, L (r+1) / W): Move never migrates data); gem_comm_init_tiles makes the strips whole rows of
 *      32 x 32-cell tiles, which the sharded path below needs.  gem_allgather_layers completes every rank's copy of the layers:
 *      every rank's strip goes DIRECTLY to every other rank (one grouped ncclSend / ncclRecv pair per peer and layer: each peer
 *      has its own xGMI link; strips of any sizes), read from a published copy of the strip and carried by the handle's
 *      communication stream -- the call returns at once, later passes over this rank's own strip run beside the transfers, and
 *      whatever observes the whole map (gem_get_layer, gem_synchronize, gem_move, ...) waits for them.
 *      Stage A (replicated binning): every rank calls gem_add* with the WHOLE cloud and fuses only its strip.             */
int  gem_comm_unique_id(void* out_128_bytes);
int  gem_comm_init(gem_handle* h, const void* unique_id_128_bytes, int nranks, int rank);
int  gem_comm_init_tiles(gem_handle* h, const void* unique_id_128_bytes, int nranks, int rank);
int  gem_get_strip(gem_handle* h, int* out_row0, int* out_row1);
int  gem_allgather_layers(gem_handle* h, int with_attributes);   /* elevation+variance (+intensity, colours) */

/*      Stage B (points sharded): rank r holds a contiguous index range of the batch -- sweeps first_global_sweep ..
 *      first_global_sweep + n_local_sweeps - 1 of n_global_sweeps (a sweep may be split between two neighbouring ranks: both
 *      pass it, the lower rank holds its head; first_point_in_sweep = index, inside its sweep, of this rank's first point --
 *      the camera sensor models derive the pixel row / column from it).  gem_add_sharded_device = for s:
 *      Mapvar_update(var_updates_global[s]); add(sweep s) on the map tiled over the ranks: each rank projects / bins / sorts its
 *      own points for the whole map (block-sorted: the records of a strip, and of every block of 256 cells, are one contiguous
 *      range), the sorted records of every strip travel to the strip's owner together with their block ranges (ncclSend /
 *      ncclRecv, one group, on the handle's communication stream; a rank's own records stay where they are), and the owner takes
 *      every block's records source by source in rank order -- ascending index ranges, so rank order is input order and every
 *      cell sees its points exactly as on one device.  With one rank nothing is exchanged and nothing returns to the host.  With
 *      more, the call enqueues the step's sort and the all-gather of its strip boundaries (16 words per rank, copied to the host)
 *      and returns; the step's SECOND HALF -- the exchange on the communication stream, the fusion on the handle's stream, and the
 *      all-gather of the layers on a stream and communicator of their own if gem_allgather_layers followed the step -- is enqueued by
 *      the NEXT call, or by whatever observes or modifies the map (gem_synchronize, gem_get_layer, gem_move, gem_mapvar_update,
 *      ...): nothing ever waits for work the same call enqueued, and consecutive steps overlap stage by stage.  All of these are
 *      collectives: every rank makes the same sequence of calls.  var_updates_global (n_global_sweeps <= 512 values, identical on
 *      all ranks) may be NULL.  No colours, no lowest tracking on this path.  Arguments, geometry and every allocation -- the receive
 *      buffers included, sized from gem_reserve's bound or from W shares like this rank's -- are checked before the step's first
 *      collective; a rank that still cannot go on in the middle of a step (a step that brings more records than foreseen, and no
 *      memory to grow) aborts both communicators, so that its peers fail instead of waiting.
 *      The two halves are exported for hosts that carry the exchange themselves (gem_amd/tiling.py with torch.distributed):
 *      gem_shard_sort_device returns the device arrays of the sorted records {h, var} (8 bytes) / keys (4 bytes),
 *      out_bounds[nstrips + 1] = the first record of every strip, and (optional) the device array of the block ranges
 *      {first, end} (8 bytes per block of 256 cells, 4 x tiles entries, positions in the sorted arrays; empty block = {0, 0});
 *      gem_shard_fuse_device walks this handle's strip through n_src sources in input order: device pointers and record counts,
 *      and optionally (d_ranges, bases -- both or neither) every source's block ranges with entry 0 = the first block of this
 *      handle's strip, and the position in the source's own arrays that d_hv[s] / d_key[s] point at; without them the blocks'
 *      records are found by search.                                                                                        */
int  gem_add_sharded_device(gem_handle* h, int n_local_sweeps, const gem_frame_params* params, const void* d_xyzi,
                            const long long* offsets, int first_global_sweep, int n_global_sweeps, int first_point_in_sweep,
                            const float* var_updates_global);
int  gem_shard_sort_device(gem_handle* h, int n_local_sweeps, const gem_frame_params* params, const void* d_xyzi,
                           const long long* offsets, int first_global_sweep, int n_global_sweeps, int first_point_in_sweep,
                           int nstrips, const int* strip_rows,
                           uint32_t* out_bounds, const void** out_d_hv, const void** out_d_key, const void** out_d_ranges);
int  gem_shard_fuse_device(gem_handle* h, int n_src, const void* const* d_hv, const void* const* d_key, const uint32_t* counts,
                           const void* const* d_ranges, const uint32_t* bases, int n_global_sweeps, const float* var_updates_global);

#ifdef __cplusplus
}
#endif
#endif
