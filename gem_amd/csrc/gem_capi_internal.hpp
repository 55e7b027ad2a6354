// gem_capi_internal.hpp -- what the translation units of the C ABI share: the handle, its arenas and staging, the pass pipelines,
// the multi-rank step.  Internal to libgem_hip.so (nothing here is installed); the ABI itself is include/gem_hip.h.
//   gem_capi_core.cpp      errors, arenas, the pinned staging buffer and the host <-> device transfers, frame constants, the deferred
//                          launches (flush_* / settle), the process-lifetime stream pools
//   gem_capi_pipeline.cpp  run_pipeline / run_sort_pipeline: which kernels a pass takes, on which streams, with which buffers
//   gem_capi.cpp           the extern "C" entry points of include/gem_hip.h (single-device part) and include/gem_hip_debug.h
//   gem_capi_comm.cpp      communicators, the all-gather of the layers, the sharded step (gem_add_sharded_device and its halves)
#pragma once

#include "../../include/gem_hip.h"
#include "../../include/gem_hip_debug.h"
#include "gem_kernels.hpp"
#include "gem_hostcopy.hpp"
#include "gem_transport.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>


using namespace gem;

namespace gemi {

extern thread_local std::string g_create_error;


struct Arena {
    void*  p = nullptr;
    size_t cap = 0;
};

struct EventPair { hipEvent_t a, b; int kind; };

} // namespace gemi

using namespace gemi;

struct StreamSet { hipStream_t s[4] = {nullptr, nullptr, nullptr, nullptr}; };   // own, bin, bin2, tab (see acquire_streams)

struct gem_handle {
    std::mutex  mu;
    std::string err;
    int         device = 0;
    hipStream_t own_stream = nullptr;
    StreamSet streams;                  // own, bin, bin2, tab as they were taken from the pool (given back together)
    hipStream_t stream = nullptr;
    gem_map_config cfg{};
    int   L = 0, cells = 0;
    float res = 0.f;
    LayerPtrs layers{};
    float center[2] = {0.f, 0.f};
    int   start[2] = {0, 0};
    float sensor_z = 0.f;
    int   row0 = 0, row1 = 0;
    int   ts = 0, T = 0;           // tile shift (0 = per pass), tiles of the last pass

    Arena stage;        // staging of host-pointer inputs / outputs
    // Pipeline intermediates, double-buffered: k_bin of pass p+1 runs on `bin_stream` while k_fuse of
    // pass p runs on `stream` (binning does not depend on the map, only on the cloud and the pose).
    struct PassBuffers {
        Arena rec, srt, seg, flag, gflag;   // records, descriptor table, touched stamps per (tile, sweep) and per (sweep, tile, 32 units)
        Arena s_hv1, s_hv2, s_key1, s_key2, s_src1, s_src2, s_cnt1, s_cnt2, s_misc;   // the sorted pipeline of big passes (gem_sort.hip)
        bool blkcnt_dirty = false;     // k_sort_project has been asked to count into s_blkcnt and k_block_prefix has not cleared it yet
        Arena s_blkcnt;                // [4 T] records per block, zero between passes (k_sort_project adds, k_block_prefix reads and clears)
        Arena s_ranges, s_shard;       // multi-GPU shard: every block's range in the sorted records; strip ids [16] | strip bounds [16]
        Arena tables;                  // batched-call tables (frames, sweep_unit0, sweep_first, var_updates)
        void* host_tables = nullptr;   // their pinned staging copy: the upload is asynchronous, `tables_done` guards its reuse
        size_t host_cap = 0;
        hipEvent_t tables_done = nullptr;
        bool tables_recorded = false;
        // what the device copy of the tables was built from (batch_tables_key): a stream of batches with the same frames, offsets,
        // increments and map pose -- a mapping loop replaying a fixed sensor rig, the benchmarks -- skips the 2 x 32 fill_frame,
        // the memset and the upload of the call (60-100 us of host time per C4 call before, against a 95 us device period)
        std::vector<unsigned char> tab_key;
        int tab_src = 0;
        hipStream_t tab_upload_stream = nullptr;
        hipEvent_t bin_done = nullptr, fuse_done = nullptr;
        bool fuse_recorded = false;
        uint32_t epoch = 0;            // touched-flag stamp of the last pass (0 = the flag table holds no live stamps)
    } pb[4];            // the tile pipeline alternates between the first two; the sorted pipeline's overlapped passes rotate through all four
    unsigned pass = 0, sort_pass = 0;
    uint32_t sort_epoch = 0;            // a number per sorted pass (SortArgs::epoch)
    int sort_streams = 2;               // binning streams the overlapped passes of the sorted pipeline alternate between (debug knob)
    bool trace = false;                 // debug knob: one line on stderr per pass of the sorted pipeline (which streams / buffers it took)
    int sort_ring = 3;                  // buffer sets they rotate through (debug knob, 2..4).  The sort of pass p may start once the walk of pass p - ring has
                                        // read its buffers: with two sets that wait -- a host round trip and then the whole sort chain -- sat between consecutive walks
                                        // (C4 block-sorted: 120 us per batch with two sets, 104 with three, 109 with four)
    // A stream of single device-resident sweeps runs as ONE launch per frame: k_frame fuses the previous
    // frame's records next to the binning of the new cloud.  The fuse of the newest frame is therefore
    // deferred until the next gem_add_device -- or until anything observes or modifies the map.
    struct Deferred { bool valid = false; FuseArgs fa{}; int ts = 0, attr = 0; } deferred;
    // The sorted pipeline's walk of an overlapped pass is launched by the NEXT call (or by whatever observes the map): by then its sort
    // has usually completed, and a walk that need not be put behind a hipStreamWaitEvent starts 1.4 us after the walk before it
    // instead of 5-7 (tools/ubench/handover.hip: the wait costs that much even when the event completed long before).  A stream of
    // depth images is bound by exactly that chain of walks (C3: 45.7 us per frame = 38 us walk + the hand-over).
    struct DeferredWalk { bool valid = false; WalkArgs wa{}; bool block_form = false; int attr = 0; unsigned slot = 0; } dwalk;
    bool defer_walk = true;             // (debug knob "defer_walk")
    // what the LAST sharded step's exchange and the LAST all-gather moved on this rank, in bytes (gem_debug_get "step_exchange_bytes_out" / "_in",
    // "gather_bytes_out" / "_in"; bench.py --gpus N prices them against the xGMI links)
    long long xbytes_out = 0, xbytes_in = 0, gbytes_out = 0, gbytes_in = 0;
    bool roctx = false;                             // roctx ranges around the entry points (gem_debug_set "roctx"; ApiRange)
    bool walk_always_wait = false;                  // a walk left to the next call waits for its sort's event even when the host has seen it complete (flush_walk)
    long long walks_unwaited = 0, walks_left = 0;   // walks left to the next call (gem_debug_get "walks_left"); of those, launched without a stream wait ("walks_unwaited")
    bool defer = true;
    hipStream_t bin_stream = nullptr;
    hipStream_t bin_stream2 = nullptr;  // the sorted pipeline sorts consecutive big passes on two streams (see run_sort_pipeline)
    hipStream_t tab_stream = nullptr;   // uploads a batched pass's tables while the binning stream is still busy with the pass before
    hipEvent_t switch_done = nullptr;   // recorded on `stream` when a pass moves its binning to `bin_stream` after passes that did not
    bool main_reads_pb = false;         // work enqueued on `stream` since the last such switch reads the pass buffers
    bool overlap = true;
    long long overlap_min_points = 1000000;        // tile pipeline: a cross-stream event pair costs 3 us, the second stream only pays for big passes
    long long sort_overlap_min_points = 100000;    // sorted pipeline: its walk is a few long chains on a mostly idle chip; the next pass's sort fits beside it (depth image 120 -> 83 us)
    bool sort_path = true;              // passes of at least sort_min_points points run the sorted pipeline (gem_sort.hip)
    // single cloud / batch of sweeps (tools/dbg/crossover.py).  Batches: block-sorted from three LiDAR sweeps on (393 k points: 43 us
    // against the tile pipeline's 46; four sweeps 43 / 56, two 46 / 35); single clouds: a 131 k-point LiDAR sweep takes 10 us on
    // the tile pipeline and 35 sorted, a 150 k-point depth image 60 and 38
    long long sort_min_points = 200000, sort_min_points_batch = 390000;
    bool walk_permute = true;           // k_fuse_walk: blocks take the tile rows centre-first
    int  sort_passes = 0;               // 0 = by map size and form (sort_geometry); 1 / 2 / 3 force it
    int few_bins = 0;                   // k_sort_scatter's ballots per wave instruction before the LDS way (0 = the built-in 8; debug knob)
    int blk_batch = 0;                  // k_fuse_block's round: 0 = by the pass's mean block load, 512 / 2048 forced (debug knob)
    int ray_depth = 4, ray_lanes = 16;  // k_raytracing: loads in flight per lane, lanes per ray (debug knobs; 16 x 4 measured best on C2)
    bool fast_laser = true;             // frames that qualify use the zero-rotation-variance form of the laser variance (fill_frame; debug knob)
    bool rank_by_ballot = false;        // k_sort_scatter ranks by ballot in every pass (debug knob)
    bool lane_sort = true;              // k_fuse_block: cells to threads by record count (debug knob)
    bool ride_events = true;            // the sort's last dispatch carries the event the walk waits for (no marker behind it)
    int  walk_lds_pad = 0;              // k_fuse_block: extra dynamic LDS per workgroup (debug knob: fewer workgroups per CU)
    int  walk_prio = 4096;              // k_fuse_block: blocks of at least this many records run at raised issue priority (debug knob, 0 = off)
    int  fuse_count = 1;                // pass 2's counts from pass 1's scatter (SortArgs::fuse_count): 0 = never, 1 = passes of up to kFuseCountMaxPoints points, 2 = always (debug knob)
    int  sort_chunk = 0;                // records per counting-sort chunk: 0 = by the pass's size (sort_chunk_for), 1024 / 4096 forced (debug knob)
    bool light_fast = true;             // k_fuse_block's light rounds by arrival slots + sorting network (debug knob)
    bool cache_tables = true;           // batched calls: skip building / uploading tables equal to the ones the buffer set already holds (debug knob)
    std::vector<unsigned char> key_scratch;
    bool plain_loop = true;             // the walks' plain chain loop for blocks whose values are in range (debug knob: 0 = the guarded loop everywhere)
    int  sort_form = 0;                 // 0 = batches of sweeps BLOCK-sorted (k_fuse_block), single clouds CELL-sorted (k_fuse_walk); 1 / 2 force cell / block
    int dbg_sweep = 0;                  // debug stamps of the dense path: which sweep (GEM_DBG_SWEEP)
    bool track_lowest = false;          // also maintain map_lowest in the fuse kernels (gem_set_lowest_tracking, for gem_raytracing)
    unsigned dense_min = 2048;          // records of one sweep in one 16x16 tile above which the tile is counting-sorted (k_fuse_list, dense path)
    Arena scratch;      // layer export
    unsigned long long* d_counters = nullptr;

    float pending[kMaxPending] = {0, 0, 0, 0};
    int   n_pending = 0;
    bool  floor_dirty = true;          // some cell may hold variance < floor (init / clear / set_layer)

    bool  timing = false, counting = false;
    std::vector<EventPair> events;     // recorded, not yet folded
    std::vector<EventPair> pool;
    gem_stats stats{};
    hipEvent_t copy_done = nullptr;

    // ---- caller-owned pageable arrays (gem_hostcopy.hpp): the handle's pinned staging buffer, the DMA between it and the device,
    //      a few threads between it and the caller's arrays.  copy_threads 0 = the runtime's own pageable path.
    static constexpr int kStageEvents = 24;
    void*  hstage = nullptr;            // hipHostMalloc'ed
    size_t hstage_cap = 0;
    bool   hstage_failed = false;       // an allocation failed: not tried again
    size_t hstage_max = 256u << 20;     // larger transfers go through the runtime
    int    copy_threads = 4;            // the calling thread + 3 workers (gem_debug_set "copy_threads")
    int    download_groups = 8;         // pieces a download is cut into: the device writes piece g + 1 into the staging buffer while the copy threads move piece g on (gem_debug_set "download_groups")
    hipEvent_t ev_stage[kStageEvents] = {};      // host-visible: a segment's DMA into the staging buffer is done
    hipEvent_t stage_read = nullptr;    // the last DMA OUT of the staging buffer is done (it may be written again)
    bool   stage_read_pending = false;
    // deferred uploads (gem_add, gem_add_batch): the call returns once the caller's arrays have been READ into one half of the staging
    // buffer and the DMA out of it is enqueued; the next call fills the other half meanwhile.  A half is reused when the DMA that read
    // it two calls ago is done.
    hipEvent_t ev_half[2] = {nullptr, nullptr};
    bool   half_pending[2] = {false, false};
    unsigned stage_par = 0;
    long long hstage_allocations = 0;
    long long xfer_ns[5] = {0, 0, 0, 0, 0};    // host time so far: upload memcpy, upload enqueue, download enqueue, download wait, download memcpy

    // ---- multi-GPU.  Two communicators, each with a stream of its own: `tp_x` carries a step's boundary
    //      all-gather and record exchange on `comm_stream`, `tp_g` the all-gather of the fused layers on `gather_stream` -- step
    //      p + 1's exchange does not queue behind step p's 46 MB of layers.  (RCCL over xGMI; W handles of one process on one
    //      device through the loopback of gem_transport.hpp in the tests.)
    std::unique_ptr<Transport> tp_x, tp_g;
    int nranks = 1, rank = 0;
    int strip_row[kMaxRanks + 1] = {0};  // storage rows [strip_row[k], strip_row[k+1]) belong to rank k (gem_comm_init / gem_comm_init_tiles)
    bool tile_strips = false;           // strips are whole rows of 32x32 tiles (needed by the sharded path)
    // points sharded (gem_shard_sort_device / gem_shard_fuse_device / gem_add_sharded_device)
    struct Shard {
        bool valid = false;
        const uint2* hv = nullptr; const uint32_t* key = nullptr;     // this device's sorted records
        const uint2* ranges = nullptr;                                 // [4 T] where every block's records are in them (k_block_prefix)
        uint32_t bounds[kMaxRanks + 1] = {0};                          // first record of every strip in them
        const uint32_t* d_bounds = nullptr;                            // ... on the device (16 words)
        int nstrips = 0, n_global_sweeps = 0;
        long long points = 0;                                          // points this device sorted for the step
        int slot = -1;                                                 // pass-buffer set the sort ran in on a binning stream (its bin_done / fuse_done events), -1: on the handle's stream
        hipStream_t stream = nullptr;                                  // the stream the sort was enqueued on
    } shard;
    // A step of gem_add_sharded_device whose SECOND HALF -- exchange, walk, all-gather of the layers if one was asked for -- is
    // still to come: the call returns once the step's sort and the all-gather of its strip boundaries are enqueued; the next call
    // (or whatever observes the map) finishes it.  That way the host never waits for a sort it has just enqueued: when it needs the
    // boundaries of step p, the sort of step p + 1 is already queued behind it (shard_finish_locked).
    struct Step {
        bool valid = false;
        int parity = 0;                                                // which of the two sets of staging / receive buffers
        int n_global_sweeps = 0;
        bool has_vu = false; float vu[512];
        Shard sd;
        bool gather = false; int gather_attrs = 0;                     // gem_allgather_layers was called behind it
    } step;
    unsigned step_seq = 0;                                             // steps begun so far (parity = step_seq & 1)
    Arena sh_dev, sh_ranges;                                           // ids / bounds / gathered bounds / variance increments (two sets); an empty shard's block ranges
    Arena sh_recv_hv[2], sh_recv_key[2], sh_recv_rng[2];               // records and block ranges received from the other ranks, one set per parity
    hipStream_t comm_stream = nullptr, gather_stream = nullptr;
    hipEvent_t ev_sorted = nullptr, ev_exchanged = nullptr;
    hipEvent_t ev_bounds[2] = {nullptr, nullptr};                      // the gathered boundaries of that parity are on the host
    hipEvent_t ev_walked[2] = {nullptr, nullptr};                      // the walk that read that parity's receive buffers is done
    bool walk_recorded[2] = {false, false};
    hipEvent_t ev_vu[2] = {nullptr, nullptr};                          // the upload of that variance-increment staging buffer is done
    bool vu_recorded[2] = {false, false};
    unsigned vu_seq = 0;
    // all-gather of the layers: sends read a PUBLISHED COPY of this rank's strip (two, rotating), receives write the other ranks' strips
    Arena published[2];
    hipEvent_t ev_published[2] = {nullptr, nullptr}, ev_gathered[2] = {nullptr, nullptr};
    bool gather_outstanding[2] = {false, false};                       // that gather has not been waited for by the handle's stream yet
    bool gathered_recorded[2] = {false, false};
    unsigned gather_seq = 0;
    long long recv_bound = 0;                                          // gem_reserve on a communicator handle: records a step may bring to this rank at most
    void* sh_host = nullptr;            // pinned staging of the small tables (kShardHostBytes)
    // optional time stamps of the last finished step's phases (gem_set_timing; gem_debug_get "step_*_ns")
    hipEvent_t ev_t[10] = {};
    bool step_timed = false;

    Arena dbg;          // optional k_fuse phase stamps
    Arena ray;          // gem_raytracing: the cells that walk, their number (two counters in turn), the snapshot of the lowest scan points
    unsigned ray_calls = 0;
    Arena color;        // gem_colorize: its own sort arrays and tables (never shared with a pass in flight on the binning stream)
    bool  dbg_on = false;
    bool  dbg_frame = false;            // debug knob: with the stamps on, a stream of single sweeps still runs as k_frame (its tiles AND its binning blocks are stamped)
    long long sort_fallbacks = 0;      // passes whose forced sorted form / pass count did not fit the map and took the other form (gem_debug_get)
    long long arena_allocations = 0;   // hipMalloc calls of ensure() so far (gem_debug_get: a stream of frames after gem_reserve must not add any)
    int   dbg_rows = 0;  // rows of `dbg` the last pass wrote, if it was a block-sorted one (else h->T rows)
    int fuse_variant = 12;
};

namespace gemi {

int fail(gem_handle* h, int code, const char* what, hipError_t e = hipSuccess);

// Optional roctx ranges around the entry points (tracing): off by default and free when off (one load of a flag);
// gem_debug_set(h, "roctx", 1) resolves roctxRangePushA / roctxRangePop from the ROCm marker library at run time (the library does
// not link against it) -- `rocprofv3 --kernel-trace --marker-trace` then shows which call enqueued which kernels.
bool roctx_load();                                   // true when the marker library was found (process-wide, once)
void roctx_push(const char* name);
void roctx_pop();
struct ApiRange {
    bool on;
    ApiRange(const gem_handle* h, const char* name);
    ~ApiRange() { if (on) roctx_pop(); }
};

#define GEM_HIP(h, call)                                                        \
    do { hipError_t _e = (call); if (_e != hipSuccess) return fail(h, GEM_ERR_HIP, #call, _e); } while (0)

// ... inside a multi-rank step, between its collectives: a rank that fails there takes the communicators down with it, so that the
// peers' pending receives fail instead of waiting for it (step_abort)
int step_abort(gem_handle* h, int rc);
#define GEM_HIP_STEP(h, call)                                                   \
    do { hipError_t _e = (call); if (_e != hipSuccess) return step_abort(h, fail(h, GEM_ERR_HIP, #call, _e)); } while (0)

int ensure(gem_handle* h, Arena& a, size_t bytes);
int ensure_zeroed(gem_handle* h, Arena& a, size_t bytes);

struct HostXfer { void* host; void* dev; size_t bytes; };

inline long long host_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

unsigned char* host_stage(gem_handle* h, size_t bytes);
int drain_staging(gem_handle* h);
int upload_arrays(gem_handle* h, const HostXfer* x, int n, bool defer_ok = false, unsigned char** zero_copy_region = nullptr, int* zero_copy_half = nullptr);
int download_arrays(gem_handle* h, const HostXfer* x, int n, size_t stage_off);
void fill_frame(const gem_handle* h, const gem_frame_params* p, FrameConst& f);
hipEvent_t get_event(gem_handle* h);

// Optional per-kernel timing: the dispatch is time-stamped through a (start, stop) event pair
// handed to hipExtLaunchKernelGGL, so the figure is the kernel's own duration on its stream.
struct Timed {
    gem_handle* h; EventPair ep{};
    bool on;
    Timed(gem_handle* hh, int kind) : h(hh), on(hh->timing && kind >= 0)
    {
        if (!on) return;
        if (!h->pool.empty()) { ep = h->pool.back(); h->pool.pop_back(); }
        else { ep.a = get_event(h); ep.b = get_event(h); }
        ep.kind = kind;
    }
    LaunchEvents events() const { LaunchEvents e; if (on) { e.start = ep.a; e.stop = ep.b; } return e; }
    ~Timed() { if (on) h->events.push_back(ep); }
};

void fold_events(gem_handle* h);
int flush_deferred(gem_handle* h);
int flush_walk(gem_handle* h);
int flush_local(gem_handle* h);
int wait_gather(gem_handle* h);
int settle(gem_handle* h);
int flush_pending(gem_handle* h, bool with_floor);
int index_to_range(int index, int L);          // gpu_process.cu:914-919

// One pipeline pass over up to `n` points that are already on the device.
struct PassInput {
    int src = 0;                       // 0 = XYZI cloud, 1 = Fuse() arrays
    bool device_input = false;         // the caller's device buffers are read directly (no staging copy)
    bool caller_device = false;        // ... and they ARE the caller's (gem_add_device, gem_add_batch_device: untouched until gem_synchronize by contract), not an arena or a staging half of the handle that the next call refills
    int n_sweeps = 1;
    long long n = 0;
    const gem_frame_params* params = nullptr;      // [n_sweeps] (src 0)
    const long long* offsets = nullptr;            // [n_sweeps+1] (batched)
    const float* var_updates = nullptr;            // [n_sweeps]  (batched, host)
    const int* sweep_orig0 = nullptr;              // [n_sweeps]  (batched, host, optional) index inside its sweep of each sweep's first point here
    const float4* xyzi = nullptr; const uint32_t* rgb = nullptr; const int* orig = nullptr;
    const int* f_index = nullptr; const float* f_height = nullptr; const float* f_var = nullptr;
    const int* f_R = nullptr; const int* f_G = nullptr; const int* f_B = nullptr; const float* f_I = nullptr;
};

// Events between the handle's OWN streams on its own device (a pass's sort -> its walk, a walk -> the sort that reuses its buffers)
// carry no system-scope fence: the kernel boundary already writes the producer's L2 lines back for the consumer's XCDs, and
// nothing on the host or on another device reads data behind them (round 4: C3 -1.8 us, C4 -2.2 us per call).  Whatever a PEER
// device or the host reads -- the record exchange and the all-gathers of the multi-rank step -- is ordered by events WITH the
// fence (ev_sorted and the other step events, comm_attach), never by these.
constexpr unsigned  kDeviceEventFlags = hipEventDisableTiming | hipEventDisableSystemFence;
constexpr int       kUnit = 64;                      // points per unit (one wave of k_bin_wave)
constexpr long long kSweepPoints = 2048ll * kUnit;   // a single cloud longer than this is processed as a batch of sweeps of this size

hipError_t acquire_streams(int device, StreamSet& out);
void release_streams(int device, const StreamSet& set);
hipError_t acquire_comm_stream(int device, hipStream_t* out);
void release_comm_stream(int device, hipStream_t st);
inline int ceil_log2(int v) { int b = 0; while ((1 << b) < v) ++b; return b; }

// The key geometry of the sorted pipelines for this map, and whether a pass of `n_sweeps` sweeps fits the 32-bit record key.
// block_form: the digits cover the BLOCK id (id >> 8) only and k_fuse_block orders a block's records by cell itself; otherwise
// they cover the whole id and k_fuse_walk streams every cell's run (gem_kernels.hpp).
struct SortGeometry { int tiles_per_row, T, id_bits, n_passes, dshift[3], dbits[3], dbins[3]; bool block_form, ok; };
SortGeometry sort_geometry(const gem_handle* h, int n_sweeps, bool block_form);

// pinned host staging of the sharded path: per parity (4096 B each) strip ids at word 0, own bounds at word 32, the gathered bounds
// [W][16] at word 64; the variance increments' two buffers at byte 8192 + 2048 b.  The device twin has the same layout.
constexpr size_t kShardHostBytes = 8192 + 2 * 2048, kShardDevBytes = 8192 + 2 * 2048;
struct ShardOpts { int sweep_id0; int nstrips; const int* strip_rows; bool bounds_stay_on_device; };   // sort only: the walk happens on the strip owners

int run_sort_pipeline(gem_handle* h, const PassInput& in, int attr, const SortGeometry& geo, const ShardOpts* shard = nullptr);
int run_pipeline(gem_handle* h, const PassInput& in0);

// gem_capi_comm.cpp
int shard_finish_locked(gem_handle* h);       // the second half of a pending gem_add_sharded_device step
int ensure_recv(gem_handle* h, int parity, size_t records);
size_t strip_blocks_of(const gem_handle* h, int p);
bool shard_sort_rotates(const gem_handle* h, long long n);
int shard_checks(gem_handle* h, int n_global_sweeps, SortGeometry* geo);

} // namespace gemi
