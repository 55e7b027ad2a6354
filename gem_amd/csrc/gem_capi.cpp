// gem_capi.cpp -- the C ABI of libgem_hip.so (see include/gem_hip.h): handle, persistent device
// arenas, host-side Move logic and launch orchestration for the gfx950 kernels.
//
// The reference keeps the map as hidden process-global __device__ state and does 15 cudaMalloc +
// 15 cudaFree + 15 cudaMemcpy per frame on this path (gpu_process.cu:1096-1141, 1165-1192).
// Here a handle owns persistent arenas that only ever grow, everything is enqueued on one HIP
// stream, and nothing returns to the host unless the caller asks for it.
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

namespace {

thread_local std::string g_create_error;

struct Arena {
    void*  p = nullptr;
    size_t cap = 0;
};

struct EventPair { hipEvent_t a, b; int kind; };

} // namespace

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

namespace {

int fail(gem_handle* h, int code, const char* what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof(buf), "%s", what);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define GEM_HIP(h, call)                                                        \
    do { hipError_t _e = (call); if (_e != hipSuccess) return fail(h, GEM_ERR_HIP, #call, _e); } while (0)

// ... inside a multi-rank step, between its collectives: a rank that fails there takes the communicators down with it, so that the
// peers' pending receives fail instead of waiting for it (step_abort)
int step_abort(gem_handle* h, int rc);
#define GEM_HIP_STEP(h, call)                                                   \
    do { hipError_t _e = (call); if (_e != hipSuccess) return step_abort(h, fail(h, GEM_ERR_HIP, #call, _e)); } while (0)

int ensure(gem_handle* h, Arena& a, size_t bytes)
{
    if (bytes <= a.cap) return GEM_OK;
    // arenas may still be in use by enqueued work
    GEM_HIP(h, hipStreamSynchronize(h->stream));
    if (h->bin_stream) GEM_HIP(h, hipStreamSynchronize(h->bin_stream));
    if (h->bin_stream2) GEM_HIP(h, hipStreamSynchronize(h->bin_stream2));
    if (h->tab_stream) GEM_HIP(h, hipStreamSynchronize(h->tab_stream));
    if (h->comm_stream) GEM_HIP(h, hipStreamSynchronize(h->comm_stream));
    if (h->gather_stream) GEM_HIP(h, hipStreamSynchronize(h->gather_stream));
    if (a.p) GEM_HIP(h, hipFree(a.p));
    a.p = nullptr; a.cap = 0;
    size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipMalloc(&a.p, want);
    if (e != hipSuccess) return fail(h, GEM_ERR_NOMEM, "hipMalloc(arena)", e);
    a.cap = want;
    ++h->arena_allocations;
    return GEM_OK;
}

// ... for tables the kernels keep all-zero between passes: cleared when (re)allocated (allocation synchronises anyway)
int ensure_zeroed(gem_handle* h, Arena& a, size_t bytes)
{
    if (bytes <= a.cap) return GEM_OK;
    const int rc = ensure(h, a, bytes);
    if (rc) return rc;
    GEM_HIP(h, hipMemsetAsync(a.p, 0, a.cap, h->stream));
    GEM_HIP(h, hipStreamSynchronize(h->stream));
    return GEM_OK;
}

// ---- caller-owned host arrays <-> device arenas ---------------------------------------------------------------------------------
struct HostXfer { void* host; void* dev; size_t bytes; };

static inline long long host_ns()
{
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// the pinned staging buffer, at least `bytes` large -- or nullptr: switched off, too large, or the allocation failed (the callers
// then hand the arrays to the runtime, which stages pageable memory itself: slower, never wrong)
unsigned char* host_stage(gem_handle* h, size_t bytes)
{
    if (h->copy_threads <= 0 || h->hstage_failed || bytes > h->hstage_max) return nullptr;
    if (bytes > h->hstage_cap) {
        if (h->hstage) {
            if (hipStreamSynchronize(h->stream) != hipSuccess) return nullptr;       // a DMA may still read it
            h->stage_read_pending = false; h->half_pending[0] = h->half_pending[1] = false;
            hipHostFree(h->hstage); h->hstage = nullptr; h->hstage_cap = 0;
        }
        const size_t want = bytes + bytes / 4 + 4096;
        if (hipHostMalloc(&h->hstage, want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->hstage = nullptr; h->hstage_failed = true; return nullptr; }
        h->hstage_cap = want;
        ++h->hstage_allocations;
    }
    for (auto& ev : h->ev_stage) if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ev = nullptr; return nullptr; }
    if (!h->stage_read && hipEventCreateWithFlags(&h->stage_read, hipEventDisableTiming) != hipSuccess) { h->stage_read = nullptr; return nullptr; }
    return static_cast<unsigned char*>(h->hstage);
}

// every DMA out of the staging buffer that a deferred upload left in flight is done (whoever uses the buffer from its start comes here first)
int drain_staging(gem_handle* h)
{
    for (int p = 0; p < 2; ++p) if (h->half_pending[p]) { GEM_HIP(h, hipEventSynchronize(h->ev_half[p])); h->half_pending[p] = false; }
    if (h->stage_read_pending) { GEM_HIP(h, hipEventSynchronize(h->stage_read)); h->stage_read_pending = false; }
    return GEM_OK;
}

// Host arrays -> device, on h->stream.  On return the caller's arrays have been READ (they may be stack arrays that die with the
// call, EMg.cpp:260-267); the device copies are enqueued.  defer_ok: the caller enqueues its kernels behind the copies ON h->stream
// (or on streams ordered behind it: main_reads_pb) and nothing on the host needs them done -- the call then does not wait for the
// DMA: it leaves it reading one half of the staging buffer while the next call's arrays are copied into the other half (a stream of
// gem_add calls: staging copy, link and kernels of consecutive sweeps overlap; before, each call ran them back to back).
int upload_arrays(gem_handle* h, const HostXfer* x, int n, bool defer_ok = false, unsigned char** zero_copy_region = nullptr, int* zero_copy_half = nullptr)
{
    if (zero_copy_region) *zero_copy_region = nullptr;
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (x[i].bytes + 255) & ~(size_t)255;
    // (tens of megabytes -- a batch of sweeps, an aggregated cloud -- come from DRAM, not from the caller's cache, and the runtime's own
    //  pageable path, which pins the pages where they lie, moves them faster than any number of copy threads through the staging
    //  buffer: 67 MB in 1.7 ms against 2.1-2.5, tools/dbg/host_batch.py)
    unsigned char* stg = (total >= (128u << 10) && total < (16u << 20)) ? host_stage(h, defer_ok ? 2 * total + 512 : total) : nullptr;
    if (!stg) {
        for (int i = 0; i < n; ++i) if (x[i].bytes) GEM_HIP(h, hipMemcpyAsync(x[i].dev, x[i].host, x[i].bytes, hipMemcpyHostToDevice, h->stream));
        GEM_HIP(h, hipEventRecord(h->copy_done, h->stream));
        GEM_HIP(h, hipEventSynchronize(h->copy_done));
        return GEM_OK;
    }
    int par = -1;
    if (defer_ok) {
        for (auto& ev : h->ev_half) if (!ev) GEM_HIP(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        if (h->stage_read_pending) { GEM_HIP(h, hipEventSynchronize(h->stage_read)); h->stage_read_pending = false; }
        par = (int)(h->stage_par++ & 1u);
        // the halves are the two halves of the BUFFER (not of this call's bytes: calls of different sizes must not overlap)
        const size_t half_at = (h->hstage_cap / 2) & ~(size_t)255;
        if (h->half_pending[par]) { GEM_HIP(h, hipEventSynchronize(h->ev_half[par])); h->half_pending[par] = false; }
        // a half that does not hold the call (the buffer was sized by a smaller deferred call and has not grown): the other half's reader first
        if (half_at < total || h->hstage_cap - half_at < total) { const int rcd = drain_staging(h); if (rcd) return rcd; par = 0; }
        else stg += (size_t)par * half_at;
        if (zero_copy_region) {
            // ZERO COPY: the arrays go into the half at the stride they would have on the device and the pass's kernels read them
            // THERE, over the link (the buffer is device-visible pinned memory) -- no DMA command, hence no hand-over between the
            // copy engine and the compute queue on either side of it (a H2D command between two kernels of one stream cost
            // ~10 us each way: 80 us per 2 MB sweep where link + kernel are 57).  The caller records the half's event behind its kernels.
            size_t off = 0;
            gem::CopySeg segs[16]; int ns = 0;
            const long long t0 = host_ns();
            for (int i = 0; i < n; ++i) {
                if (x[i].bytes) segs[ns++] = {stg + off, x[i].host, x[i].bytes};
                off += (x[i].bytes + 255) & ~(size_t)255;
                if (ns == 16 || i == n - 1) { if (ns) gem::CopyPool::get().run(segs, ns, h->copy_threads); ns = 0; }
            }
            h->xfer_ns[0] += host_ns() - t0;
            *zero_copy_region = stg; *zero_copy_half = par;
            return GEM_OK;
        }
    } else { const int rcd = drain_staging(h); if (rcd) return rcd; }
    // Arrays that follow each other on the device at the staging buffer's own 256-byte stride form one region, copied by DMA
    // commands that ignore the array boundaries (a command costs ~9 us before its first byte: seven 0.5 MB arrays one by one run at
    // 29 GB/s, as two commands at 43); the DMA of one group runs under the memcpy of the next.
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t off = 0;
    for (int i = 0; i < n;) {
        int e = i + 1;
        size_t len = pad(x[i].bytes);
        while (e < n && e - i < 16 && static_cast<unsigned char*>(x[e].dev) == static_cast<unsigned char*>(x[i].dev) + len) len += pad(x[e++].bytes);   // (a region has at most 16 members: segs[])
        // the DMA ends with the region's last BYTE: the padding behind the last member belongs to nobody (the callers size their
        // device arrays by what they hold, not by the staging buffer's stride)
        const size_t real_len = len - pad(x[e - 1].bytes) + x[e - 1].bytes;
        // (deferred: the DMA of this call runs beside the NEXT call's copy, so a region of up to 8 MB goes as ONE command -- a command
        //  costs ~9 us before its first byte; not deferred: in two, the second copy beside the first command)
        const size_t group = pad(len <= (8u << 20) ? (par >= 0 ? len : std::max<size_t>(len / 2, 512u << 10)) : (4u << 20));
        for (size_t a = 0; a < len; a += group) {
            const size_t b = std::min(len, a + group);
            gem::CopySeg segs[16]; int ns = 0;
            size_t m_off = 0;
            for (int m = i; m < e; ++m) {                                           // the members' parts inside [a, b)
                const size_t lo = std::max(a, m_off), hi = std::min(b, m_off + x[m].bytes);
                if (lo < hi) segs[ns++] = {stg + off + lo, static_cast<const unsigned char*>(x[m].host) + (lo - m_off), hi - lo};
                m_off += pad(x[m].bytes);
            }
            const long long t0 = host_ns();
            gem::CopyPool::get().run(segs, ns, h->copy_threads);
            const long long t1 = host_ns();
            const size_t b_real = std::min(b, real_len);
            if (a < b_real) GEM_HIP(h, hipMemcpyAsync(static_cast<unsigned char*>(x[i].dev) + a, stg + off + a, b_real - a, hipMemcpyHostToDevice, h->stream));
            h->xfer_ns[0] += t1 - t0; h->xfer_ns[1] += host_ns() - t1;
        }
        off += len;
        i = e;
    }
    if (par >= 0) {
        // deferred: the DMA stays in flight; the pass's kernels follow it on h->stream, and the binning streams of an overlapped pass
        // are put behind h->stream before they read the arena (main_reads_pb: run_pipeline / run_sort_pipeline)
        GEM_HIP(h, hipEventRecord(h->ev_half[par], h->stream));
        h->half_pending[par] = true;
        h->main_reads_pb = true;
        return GEM_OK;
    }
    // The pipeline reads the arena on its binning streams too, which are not ordered behind h->stream by anything but the host:
    // the copies are waited for, as they were when the runtime staged the arrays.
    GEM_HIP(h, hipEventRecord(h->stage_read, h->stream));
    GEM_HIP(h, hipEventSynchronize(h->stage_read));
    h->stage_read_pending = false;
    return GEM_OK;
}

// Device -> host arrays, after everything enqueued on h->stream so far.  Returns when the caller's arrays hold the data (and
// h->stream is idle).  `stage_off`: the staging bytes before it may still be read by this call's own uploads.
int download_arrays(gem_handle* h, const HostXfer* x, int n, size_t stage_off)
{
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (x[i].bytes + 255) & ~(size_t)255;
    stage_off = (stage_off + 255) & ~(size_t)255;
    { const int rcd = drain_staging(h); if (rcd) return rcd; }
    unsigned char* stg = total >= (128u << 10) ? host_stage(h, stage_off + total) : nullptr;
    if (!stg) {
        for (int i = 0; i < n; ++i) if (x[i].bytes) GEM_HIP(h, hipMemcpyAsync(x[i].host, x[i].dev, x[i].bytes, hipMemcpyDeviceToHost, h->stream));
        GEM_HIP(h, hipStreamSynchronize(h->stream));
        h->stage_read_pending = false;
        return GEM_OK;
    }
    stg += stage_off;
    const long long t_begin = host_ns();
    // groups of pieces, a launch of k_copy_list + an event each: while the device writes group g + 1 into the staging buffer the
    // copy threads move group g on to the caller's arrays.  At most kStageEvents groups of at most kCopyListMax pieces.
    // Eight groups (at least 768 KB each).  Measured alternatives on the 13 MB of Map_feature: two streams taking turns, to hide the
    // ~6 us the link idles at the event between two launches: 322 -> 338 us; few large groups first and small ones last: 322 -> 404 us
    // (the copy threads fall behind on a 5 MB group: 84 -> 54 GB/s).
    const size_t n_groups = (size_t)std::max(1, std::min(h->download_groups, (int)gem_handle::kStageEvents - 10));     // (nine arrays can add nine part groups)
    size_t group = std::max<size_t>(n_groups > 8 ? (256u << 10) : (768u << 10), (total + n_groups - 1) / n_groups);
    group = (group + 255) & ~(size_t)255;
    gem::CopySeg segs[gem_handle::kStageEvents][kCopyListMax];
    int nseg[gem_handle::kStageEvents] = {};
    int ng = 0;
    {
        CopyList cl; cl.n = 0;
        size_t in_group = 0, off = 0;
        auto flush = [&]() -> int {
            if (!cl.n) return GEM_OK;
            if (ng == gem_handle::kStageEvents) return fail(h, GEM_ERR_INVALID, "download_arrays: too many groups");      // (nine arrays: at most 9 + 9)
            GEM_HIP(h, launch_copy_list(h->stream, cl));
            GEM_HIP(h, hipEventRecord(h->ev_stage[ng], h->stream));
            nseg[ng++] = cl.n; cl.n = 0; in_group = 0;
            return GEM_OK;
        };
        for (int i = 0; i < n; ++i) {
            for (size_t o = 0; o < x[i].bytes;) {
                const size_t b = std::min(x[i].bytes - o, group - in_group);
                cl.piece[cl.n] = {stg + off + o, static_cast<const unsigned char*>(x[i].dev) + o, b};
                segs[ng][cl.n] = {static_cast<unsigned char*>(x[i].host) + o, stg + off + o, b};
                ++cl.n; in_group += b; o += b;
                if (in_group >= group || cl.n == kCopyListMax) { const int rcf = flush(); if (rcf) return rcf; }
            }
            off += (x[i].bytes + 255) & ~(size_t)255;
        }
        { const int rcf = flush(); if (rcf) return rcf; }
    }
    long long t0 = host_ns();
    h->xfer_ns[2] += t0 - t_begin;
    for (int g = 0; g < ng; ++g) {
        GEM_HIP(h, hipEventSynchronize(h->ev_stage[g]));
        const long long t1 = host_ns();
        gem::CopyPool::get().run(segs[g], nseg[g], h->copy_threads);
        const long long t2 = host_ns();
        h->xfer_ns[3] += t1 - t0; h->xfer_ns[4] += t2 - t1; t0 = t2;
    }
    h->stage_read_pending = false;                      // the last group's event followed everything on the stream
    return GEM_OK;
}

void fill_frame(const gem_handle* h, const gem_frame_params* p, FrameConst& f)
{
    memset(&f, 0, sizeof(f));
    if (p) {
        for (int i = 0; i < 12; ++i) f.T[i] = p->T[i];
        f.lower = p->lower; f.upper = p->upper;
        // GPU:397 compares (double)h with the double bounds.  For a float h, (double)h > lower  <=>  h > the largest float <= lower
        // (no float lies strictly between that one and its successor, which is above `lower`), and (double)h < upper  <=>
        // h < the smallest float >= upper; NaN bounds stay NaN (never inside).  The kernels compare floats.
        f.lower_f = (float)p->lower; if ((double)f.lower_f > p->lower) f.lower_f = nextafterf(f.lower_f, -INFINITY);
        f.upper_f = (float)p->upper; if ((double)f.upper_f < p->upper) f.upper_f = nextafterf(f.upper_f, INFINITY);
        for (int i = 0; i < 8; ++i) f.sp[i] = p->sensor_params[i];
        for (int i = 0; i < 3; ++i) { f.Js[i] = p->sensor_jacobian[i]; f.P[i] = p->P_mul_C_BM_T[i]; }
        for (int i = 0; i < 9; ++i) { f.Q[i] = p->rotation_variance[i]; f.C[i] = p->C_SB_T[i]; f.Bs[i] = p->B_r_BS_skew[i]; }
        f.filter_on = p->filter.enabled;
        f.fbx = p->filter.box_x; f.fby = p->filter.box_y; f.fband = p->filter.band_y; f.fplane = p->filter.plane_y;
        f.model = p->sensor_model;
        f.orig_width = p->original_width;
    }
    f.cx = h->center[0]; f.cy = h->center[1];
    f.sx = h->start[0];  f.sy = h->start[1];
    f.L = h->L; f.res = h->res;
    f.row0 = h->row0; f.row1 = h->row1;
    // kModelLaserFast (gem_device.hpp, height_variance): the variance's rotation term vanishes and its last addend is a constant
    f.beam_a = (float)f.sp[1]; f.beam_c = (float)f.sp[2];
    f.t2 = 0.f; f.fast_laser = 0;
    if (p && f.model == GEM_MODEL_LASER && h->fast_laser) {
        const float min_r = (float)f.sp[0], vn = min_r * min_r;
        const float c0 = f.Js[0] * 0.0f, c1 = f.Js[1] * 0.0f, c2 = f.Js[2] * vn;     // b2 = dot3(Js0, 0, Js1, 0, Js2, vn), GPU:293-298
        const float b2 = c0 + (c1 + c2);
        const float t2 = b2 * f.Js[2];
        auto small = [](float v, float bound) { return std::isfinite(v) && std::fabs(v) <= bound; };
        bool ok = t2 > 0.f && std::isfinite(t2) && small(f.beam_a, 1e6f) && small(f.beam_c, 1e6f);
        for (int i = 0; i < 9; ++i) ok = ok && f.Q[i] == 0.0f && small(f.C[i], 1e6f) && small(f.Bs[i], 1e6f);
        for (int i = 0; i < 3; ++i) ok = ok && small(f.Js[i], 1e6f) && small(f.P[i], 1e6f);
        // the frame bounds the points it accepts: rows of T orthonormal within 1 %, translation / window / map extent below 1e9
        for (int i = 0; i < 3 && ok; ++i)
            for (int j = i; j < 3; ++j) {
                double d = 0.0;
                for (int k = 0; k < 3; ++k) d += (double)f.T[4 * i + k] * (double)f.T[4 * j + k];
                ok = ok && std::fabs(d - (i == j ? 1.0 : 0.0)) <= 0.01;
            }
        for (int i = 0; i < 3; ++i) ok = ok && small(f.T[4 * i + 3], 1e9f);
        ok = ok && std::isfinite(f.lower) && std::isfinite(f.upper) && std::fabs(f.lower) <= 1e9 && std::fabs(f.upper) <= 1e9;
        ok = ok && small(f.cx, 1e9f) && small(f.cy, 1e9f) && (double)f.L * (double)f.res <= 1e9;
        // the straight-line binning divides by the resolution through its refined reciprocal (gem_device.hpp, div_binning)
        ok = ok && f.res >= 9.5367431640625e-7f && f.res <= 1048576.0f && f.L >= 2;
        if (ok) { f.t2 = t2; f.fast_laser = 1; }
    }
}

hipEvent_t get_event(gem_handle* h)
{
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    (void)h;
    return e;
}

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

// (Pinning caller-owned pageable arrays for the duration of a call -- hipHostRegister ... hipHostUnregister around the copies --
//  was built and measured in round 4: Process_points 345 -> 235 us.  It is NOT in the product: the randomised soak died with GPU
//  memory faults on host heap addresses a few hundred scenarios in, every time, and ran clean for 4600 scenarios without it.
//  Registrations of heap memory that is freed and reused between calls are not something this runtime tolerates.)
void fold_events(gem_handle* h)
{
    for (auto& ep : h->events) {
        float ms = 0.f;
        if (hipEventSynchronize(ep.b) == hipSuccess && hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
            if (ep.kind == 0)      { h->stats.ms_bin += ms; h->stats.launches_bin++; }
            else if (ep.kind == 1) { h->stats.ms_fuse += ms; h->stats.launches_fuse++; }
            else if (ep.kind == 2) { h->stats.ms_frame += ms; h->stats.launches_frame++; }
            else if (ep.kind == 9) { h->stats.ms_walk += ms; h->stats.launches_walk++; }
            else                   { h->stats.ms_sort[ep.kind - 3] += ms; if (ep.kind == 3) h->stats.launches_sort++; }
        }
        else (void)hipGetLastError();      // (a pair that was never recorded: not an error of the next launch)
        h->pool.push_back(ep);
    }
    h->events.clear();
}

// the fuse of the newest frame, if it is still pending (see gem_handle::deferred)
int flush_deferred(gem_handle* h)
{
    if (!h->deferred.valid) return GEM_OK;
    h->deferred.valid = false;
    h->main_reads_pb = true;
    if (h->dbg_frame) h->deferred.fa.dbg = nullptr;            // (the stamps of the last k_frame stay readable: this flush is not the launch being profiled)
    Timed t(h, 1);
    // A deferred list is one k_frame would have fused beside the next sweep's binning (16x16 tiles, one sweep, no attributes):
    // the same kernel without a binning half -- six workgroups per CU hold every tile of a 600^2 map at once, k_fuse_list's four
    // take two tile lifetimes (10.0-10.4 us against ~6.5 for the C2 sweep; this launch ends every synchronised run of sweeps)
    gem::BinArgs no_bin{};
    GEM_HIP(h, launch_frame(h->stream, h->deferred.fa, no_bin, h->deferred.attr, t.events()));
    return GEM_OK;
}

// The walk a sorted pass left to its successor (gem_handle::dwalk): behind its sort -- by an event wait only if the sort is still running.
int flush_walk(gem_handle* h)
{
    if (!h->dwalk.valid) return GEM_OK;
    h->dwalk.valid = false;
    gem_handle::PassBuffers& pb = h->pb[h->dwalk.slot];
    // ORDERING ASSUMPTION (stated, not proven by the API): once hipEventQuery reports bin_done complete, the sort's stores are visible
    // to a kernel launched afterwards on ANOTHER stream of this device -- bin_done carries no system-scope fence, so this rests on the
    // release at the end of the sort's last dispatch (L2 write-back of the device's own XCDs) and the acquire at the start of the
    // walk's, which is what ROCm 7.x does for every kernel boundary.  The "walk_always_wait" knob states the edge instead (5 us of
    // the walk's stream, profiles/r05_ubench_handover.txt); the soak and tests/test_parity_gpu.py run both.
    if (!h->walk_always_wait && hipEventQuery(pb.bin_done) == hipSuccess) ++h->walks_unwaited;
    else { (void)hipGetLastError(); GEM_HIP(h, hipStreamWaitEvent(h->stream, pb.bin_done, 0)); }
    {
        Timed t(h, 9);
        GEM_HIP(h, h->dwalk.block_form ? launch_block_walk(h->stream, h->dwalk.wa, h->dwalk.attr, t.events()) : launch_walk(h->stream, h->dwalk.wa, h->dwalk.attr, t.events()));
    }
    GEM_HIP(h, hipEventRecord(pb.fuse_done, h->stream)); pb.fuse_recorded = true;
    return GEM_OK;
}

// Every launch this handle has put off on its OWN stream, oldest first: the walk a sorted pass left to its successor, then the fuse
// of the newest single sweep.  Whatever fuses, publishes or observes the map calls this (or settle, which does) -- never one of the two alone.
int flush_local(gem_handle* h)
{
    { const int rc = flush_walk(h); if (rc) return rc; }
    return flush_deferred(h);
}

// An all-gather of the fused strips still in flight on the gather stream writes the other ranks' strips: whatever observes or
// modifies the whole map on the handle's stream comes after it.
int wait_gather(gem_handle* h)
{
    for (int g = 0; g < 2; ++g) {
        if (!h->gather_outstanding[g]) continue;
        GEM_HIP(h, hipStreamWaitEvent(h->stream, h->ev_gathered[g], 0));
        h->gather_outstanding[g] = false;
    }
    return GEM_OK;
}

int shard_finish_locked(gem_handle* h);       // the second half of a pending gem_add_sharded_device step (below)
int ensure_recv(gem_handle* h, int parity, size_t records);

// Everything the handle has put off -- the second half of a sharded step (COLLECTIVE: every rank gets here with the same sequence
// of calls), the fuse of the newest single sweep, the transfers of an all-gather -- before something observes or modifies the map.
int settle(gem_handle* h)
{
    { const int rc = shard_finish_locked(h); if (rc) return rc; }
    { const int rc = flush_local(h); if (rc) return rc; }
    return wait_gather(h);
}

// standalone dense pass: queued Mapvar_update increments (+ optionally the variance floor)
int flush_pending(gem_handle* h, bool with_floor)
{
    { const int rc = settle(h); if (rc) return rc; }
    if (h->n_pending == 0 && !with_floor) return GEM_OK;
    GEM_HIP(h, launch_dense_variance(h->stream, h->layers.variance, h->cells, h->n_pending, h->pending, with_floor ? 1 : 0,
                                     h->cfg.variance_floor));
    h->n_pending = 0;
    if (with_floor) h->floor_dirty = false;
    return GEM_OK;
}

int index_to_range(int index, int L)          // gpu_process.cu:914-919
{
    if (index < 0) index += ((-index / L) + 1) * L;
    return index % L;
}

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

// The second stream carries the map-independent half of a big pass (projection / binning / sorting) next to the fusion of the
// pass before.  (A high-priority stream for it was measured: no effect.)
// Streams live as long as the process and pass from handle to handle, four at a time, each keeping its role: the runtime maps
// every stream onto one of a few hardware queues when it is created, streams that share a queue run one after the other, and
// which queues overlap well with which is a matter of their creation order.  Measured on C4 with all 24 assignments of four
// consecutively created streams s0..s3 to (own, bin, bin2, tab) (`tools/dbg/roles.py`): 125 us per batch when `own` and one
// binning stream are among {s0, s1} and the other binning stream among {s2, s3}; 140 us for the other split assignments; 185 us
// with `own` among {s2, s3} and both binning streams among {s0, s1}.  And a handle whose streams were created after another
// handle's had been DESTROYED found its binning streams on the queue of its own stream: no overlap at all (C5 371 -> 391 us,
// C4 135 -> 185).  Hence: sets of four created together, roles by creation order, never destroyed.
static std::mutex g_stream_pool_mu;
static std::vector<StreamSet> g_stream_pool[64];

static hipError_t acquire_streams(int device, StreamSet& out)
{
    {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        if (device >= 0 && device < 64 && !g_stream_pool[device].empty()) {
            out = g_stream_pool[device].back(); g_stream_pool[device].pop_back();
            return hipSuccess;
        }
    }
    // The handle's own stream -- where the fusion kernels run -- is created with the HIGHEST priority: when a pass's walk and the
    // next passes' sort kernels are in flight together, the walk's workgroups are dispatched first.  A block-sorted batch ends with
    // the chains of the blocks under the sensor, and every microsecond those wait for a slot is a microsecond of the batch (C4:
    // 106 -> 99 us per batch); the cell-sorted aggregated cloud, whose three-pass sort is the longer chain, pays 3 % for it
    // (C5: 346 -> 356 us).
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) { prio_lo = prio_hi = 0; (void)hipGetLastError(); }
    for (int i = 0; i < 4; ++i) {
        const hipError_t e = i == 0 ? hipStreamCreateWithPriority(&out.s[i], hipStreamNonBlocking, prio_hi)
                                    : hipStreamCreateWithFlags(&out.s[i], hipStreamNonBlocking);
        if (e != hipSuccess) { for (int j = 0; j < i; ++j) hipStreamDestroy(out.s[j]); out = StreamSet{}; return e; }
    }
    return hipSuccess;
}

// the communication stream of a handle that joined a communicator (gem_comm_init*): pooled like the others, never destroyed
static std::vector<hipStream_t> g_comm_pool[64];

static hipError_t acquire_comm_stream(int device, hipStream_t* out)
{
    {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        if (device >= 0 && device < 64 && !g_comm_pool[device].empty()) { *out = g_comm_pool[device].back(); g_comm_pool[device].pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

static void release_comm_stream(int device, hipStream_t st)
{
    if (!st) return;
    hipStreamSynchronize(st);
    if (device < 0 || device >= 64) { hipStreamDestroy(st); return; }
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    g_comm_pool[device].push_back(st);
}

static void release_streams(int device, const StreamSet& set)
{
    if (!set.s[0]) return;
    for (hipStream_t st : set.s) hipStreamSynchronize(st);
    if (device < 0 || device >= 64) { for (hipStream_t st : set.s) hipStreamDestroy(st); return; }
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    g_stream_pool[device].push_back(set);
}

static int ceil_log2(int v) { int b = 0; while ((1 << b) < v) ++b; return b; }

// The key geometry of the sorted pipelines for this map, and whether a pass of `n_sweeps` sweeps fits the 32-bit record key.
// block_form: the digits cover the BLOCK id (id >> 8) only and k_fuse_block orders a block's records by cell itself; otherwise
// they cover the whole id and k_fuse_walk streams every cell's run (gem_kernels.hpp).
struct SortGeometry { int tiles_per_row, T, id_bits, n_passes, dshift[3], dbits[3], dbins[3]; bool block_form, ok; };
SortGeometry sort_geometry(const gem_handle* h, int n_sweeps, bool block_form)
{
    SortGeometry g{};
    g.block_form = block_form;
    g.tiles_per_row = (h->L + 31) / 32;
    g.T = g.tiles_per_row * g.tiles_per_row;
    g.id_bits = 10 + std::max(1, ceil_log2(g.T));                 // id = tile << 10 | cell in tile
    const int lo = block_form ? 8 : 0;                            // first bit the digits cover
    const long long values = (((long long)g.T) << 10) >> lo;      // ids / block ids in use: 0 .. values - 1
    // Digits of about equal width, at most ten bits: the records of a (chunk, bin) leave k_sort_scatter as one run, and with
    // thousands of bins a 4096-record chunk has one or two records per run -- no coalescing left (cell-sorted, the 2400^2 map in
    // two passes of 2048 / 2813 bins: 206 + 158 us; in three passes of 256 / 256 / 88 bins: see DESIGN.md).  Block ids are
    // different: consecutive points of a scan fall into few blocks, the runs are long whatever the number of bins, and a map of
    // up to kOnePassMaxBins blocks (600^2: 1444) is sorted by ONE pass.
    if (h->sort_passes) g.n_passes = h->sort_passes;
    else if (block_form) g.n_passes = values <= kOnePassMaxBins ? 1 : (g.id_bits - lo <= 20 ? 2 : 3);
    else g.n_passes = g.id_bits <= 20 ? 2 : 3;
    int shift = lo;
    for (int i = 0; i < g.n_passes; ++i) {
        const int left = g.n_passes - i;
        // (rounded down: the lowest digit sees the records in input order -- every bin in use, a run per bin and chunk -- and pays
        //  for its bins; the higher digits see them sorted by the lower ones, longer runs.  Cell-sorted 600^2: 512 x 722 bins
        //  28.7 + 26.4 us, 1024 x 361 37.7 + 21.4, 256 x 1444 27.6 + 38.8)
        int bits = (g.id_bits - shift) / left;
        if (i == 0 && !block_form) bits = std::max(bits, 8);      // the 256 cells of a k_fuse_walk workgroup never straddle a bin of the last pass
        if (i == g.n_passes - 1) bits = g.id_bits - shift;
        bits = std::max(bits, 1);
        g.dshift[i] = shift; g.dbits[i] = bits;
        g.dbins[i] = i == g.n_passes - 1 ? (int)(((((long long)g.T) << 10) - 1) >> shift) + 1 : 1 << bits;
        shift += bits;
    }
    const long long max_sweeps = std::min<long long>(512, (1ll << (32 - g.id_bits)) - 1);     // the sweep field is never all ones
    g.ok = g.id_bits <= 26 && n_sweeps <= max_sweeps && g.dshift[g.n_passes - 1] >= 8 && shift == g.id_bits;
    for (int i = 0; i < g.n_passes; ++i) {
        g.ok = g.ok && g.dbins[i] <= kSortMaxBins && g.dbits[i] >= 1;
        g.ok = g.ok && sort_shape(g.dbins[i], true, kSortChunkRecords).lds <= 160 * 1024;   // what launch_sort checks (a forced pass count may not fit; the big chunk needs the most)
    }
    return g;
}

// Everything a batched pass's device tables are a function of, as bytes: equal keys = equal tables.
void batch_tables_key(const gem_handle* h, const PassInput& in, int kind, const void* device_tables, const std::vector<int>& first_of_sweep, std::vector<unsigned char>& key)
{
    key.clear();
    auto put = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); key.insert(key.end(), b, b + n); };
    const int head[8] = {kind, in.n_sweeps, in.var_updates ? 1 : 0, in.sweep_orig0 ? 1 : 0, h->L, h->row0, h->row1, h->fast_laser ? 1 : 0};
    put(head, sizeof(head));
    put(&device_tables, sizeof(device_tables));
    put(h->center, sizeof(h->center)); put(h->start, sizeof(h->start)); put(&h->res, sizeof(h->res));
    put(in.params, sizeof(gem_frame_params) * in.n_sweeps);
    put(in.offsets, sizeof(long long) * (in.n_sweeps + 1));
    if (in.var_updates) put(in.var_updates, sizeof(float) * in.n_sweeps);
    if (in.sweep_orig0) put(in.sweep_orig0, sizeof(int) * in.n_sweeps);
    put(first_of_sweep.data(), sizeof(int) * first_of_sweep.size());
}

// One pass through the sorted pipeline (gem_sort.hip): six sort kernels on the binning stream, k_fuse_walk on the handle's.
// pinned host staging of the sharded path: per parity (4096 B each) strip ids at word 0, own bounds at word 32, the gathered bounds
// [W][16] at word 64; the variance increments' two buffers at byte 8192 + 2048 b.  The device twin has the same layout.
constexpr size_t kShardHostBytes = 8192 + 2 * 2048, kShardDevBytes = 8192 + 2 * 2048;
struct ShardOpts { int sweep_id0; int nstrips; const int* strip_rows; bool bounds_stay_on_device; };   // sort only: the walk happens on the strip owners

int run_sort_pipeline(gem_handle* h, const PassInput& in, int attr, const SortGeometry& geo, const ShardOpts* shard = nullptr)
{
    const bool batched = in.n_sweeps > 1;
    const bool with_src = (attr & 3) != 0;
    const int chunk = sort_chunk_for(in.n, h->sort_chunk);              // 1024-record chunks for passes that 4096-record ones would leave on a third of the chip
    const SortShape sh1 = sort_shape(geo.dbins[0], with_src, chunk);
    std::vector<int> chunk0(in.n_sweeps + 1, 0);
    for (int s = 0; s < in.n_sweeps; ++s) {
        const long long cnt = batched ? in.offsets[s + 1] - in.offsets[s] : in.n;
        chunk0[s + 1] = chunk0[s] + (int)((cnt + sh1.chunk - 1) / sh1.chunk);
    }
    const int NC1 = chunk0[in.n_sweeps];
    const bool dense = h->n_pending > 0 || h->floor_dirty || (batched && in.var_updates != nullptr);
    const int T = geo.T;
    h->T = T;

    bool overlap = h->overlap && in.n >= std::min(h->overlap_min_points, h->sort_overlap_min_points) && h->stream == h->own_stream && !h->counting &&
                   (!shard || shard->bounds_stay_on_device);          // (the halves' sort returns its strip boundaries to the host: nothing to overlap)
    { const int rcd = flush_deferred(h); if (rcd) return rcd; }
    // this pass leaves its walk to the next call (gem_handle::dwalk) -- and then launches the previous pass's walk late, behind its own sort's launches
    const bool leave_walk = overlap && !shard && h->defer_walk && in.caller_device && attr == 0 && !h->timing && !h->dbg_on;
    if (!leave_walk) { const int rcd = flush_walk(h); if (rcd) return rcd; }
    // a shard bins into the WHOLE map (its records go to the strip owners); the frames carry the strip
    const int keep_row0 = h->row0, keep_row1 = h->row1;
    struct RestoreRows { gem_handle* h; int r0, r1; ~RestoreRows() { h->row0 = r0; h->row1 = r1; } } restore{h, keep_row0, keep_row1};
    if (shard) { h->row0 = 0; h->row1 = h->L; }
    // Consecutive overlapped passes sort on TWO binning streams in turn: the sort of a pass is a chain of six dependent kernels
    // that keep the chip's VALUs busy less than half of the time, so the tail of one pass's chain runs next
    // to the head of the next one's -- and next to the walk of the pass before, which alone has to follow the walk before it
    // (C4 150 -> 125 us per batch, C5 395 -> 355; a third stream: 129 / 365).
    const unsigned seq = overlap ? h->sort_pass++ : 0u;
    const unsigned slot = overlap ? seq % (unsigned)h->sort_ring : 0u;
    gem_handle::PassBuffers& pb = h->pb[slot];
    hipStream_t sbin = overlap ? (((seq & 1u) && h->sort_streams > 1 && h->bin_stream2) ? h->bin_stream2 : h->bin_stream) : h->stream;
    if (overlap && h->main_reads_pb) {                   // see run_pipeline
        GEM_HIP(h, hipEventRecord(h->switch_done, h->stream));
        GEM_HIP(h, hipStreamWaitEvent(h->bin_stream, h->switch_done, 0));
        if (h->bin_stream2) GEM_HIP(h, hipStreamWaitEvent(h->bin_stream2, h->switch_done, 0));
        if (h->tab_stream) GEM_HIP(h, hipStreamWaitEvent(h->tab_stream, h->switch_done, 0));      // (the batch tables of a pass buffer are uploaded there)
        h->main_reads_pb = false;
        for (auto& b : h->pb) b.fuse_recorded = false;
    }
    if (!overlap) h->main_reads_pb = true;
    if (h->trace)
        fprintf(stderr, "[gem] sorted pass: n=%lld sweeps=%d overlap=%d (knob %d, min %lld, own stream %d, counting %d, shard %d) slot=%u stream=%s\n",
                (long long)in.n, in.n_sweeps, (int)overlap, (int)h->overlap, (long long)std::min(h->overlap_min_points, h->sort_overlap_min_points), (int)(h->stream == h->own_stream),
                (int)h->counting, (int)(shard != nullptr), slot, sbin == h->stream ? "main" : (sbin == h->bin_stream ? "bin" : "bin2"));

    const long long nc2max = (in.n + sh1.chunk - 1) / sh1.chunk;
    const size_t N = (size_t)in.n;
    int rc;
    // (+64 bytes: k_fuse_walk fetches whole groups of four records; a cell's last group may reach past the last record)
    if ((rc = ensure(h, pb.s_hv1, N * 8 + 64))) return rc;
    if ((rc = ensure(h, pb.s_hv2, N * 8 + 64))) return rc;
    if ((rc = ensure(h, pb.s_key1, N * 4 + 64))) return rc;
    if ((rc = ensure(h, pb.s_key2, N * 4 + 64))) return rc;
    if (with_src) {
        if ((rc = ensure(h, pb.s_src1, N * 4 + 64))) return rc;
        if ((rc = ensure(h, pb.s_src2, N * 4 + 64))) return rc;
    }
    int bins_hi = 1;                                                  // the later passes share one count table
    for (int i = 1; i < geo.n_passes; ++i) bins_hi = std::max(bins_hi, geo.dbins[i]);
    if ((rc = ensure(h, pb.s_cnt1, (size_t)NC1 * geo.dbins[0] * 4))) return rc;
    if ((rc = ensure(h, pb.s_cnt2, (size_t)nc2max * bins_hi * 4 + 16))) return rc;
    // segment sums [pass][4][bins] | record count | bin bases of the last pass [bins + 1]
    size_t o_seg[3] = {0, 0, 0}, o_next = 0;
    for (int i = 0; i < geo.n_passes; ++i) { o_seg[i] = o_next; o_next += (size_t)geo.dbins[i] * 16; }
    const size_t o_total = o_next, o_base = (o_total + 4 + 15) & ~(size_t)15;
    const size_t o_segcnt = (o_base + ((size_t)geo.dbins[geo.n_passes - 1] + 1) * 4 + 15) & ~(size_t)15;
    if ((rc = ensure(h, pb.s_misc, o_segcnt + (size_t)NC1 * kSortSegsPerChunk * 4))) return rc;
    // the walk of pass p-2 has read these buffers (host-side wait, see run_pipeline)
    if (overlap && pb.fuse_recorded) GEM_HIP(h, hipEventSynchronize(pb.fuse_done));

    SortArgs sa{};
    WalkArgs wa{};
    int batch_src = -1;                                               // which k_sort_project instantiation the batch's frames take (cached with the tables)
    if (batched) {
        // tables: frames | chunk0 | first | var_updates
        const size_t o_frames = 0;
        const size_t o_chunk0 = o_frames + sizeof(FrameConst) * in.n_sweeps;
        const size_t o_first = (o_chunk0 + sizeof(int) * (in.n_sweeps + 1) + 15) & ~(size_t)15;
        const size_t o_var = o_first + sizeof(long long) * (in.n_sweeps + 1);
        const size_t o_orig = o_var + sizeof(float) * in.n_sweeps;
        const size_t total = o_orig + sizeof(int) * in.n_sweeps;
        if (total > pb.tables.cap) pb.tab_key.clear();               // (a new allocation holds nothing, even at the old address)
        if ((rc = ensure(h, pb.tables, total))) return rc;
        batch_tables_key(h, in, 0, pb.tables.p, chunk0, h->key_scratch);
        const bool tables_cached = h->cache_tables && h->key_scratch == pb.tab_key;
        if (!tables_cached) {
            if (total > pb.host_cap) {
                if (pb.tables_recorded) GEM_HIP(h, hipEventSynchronize(pb.tables_done));
                if (pb.host_tables) GEM_HIP(h, hipHostFree(pb.host_tables));
                pb.host_tables = nullptr; pb.host_cap = 0;
                GEM_HIP(h, hipHostMalloc(&pb.host_tables, total * 2, hipHostMallocDefault));
                pb.host_cap = total * 2;
            }
            if (!pb.tables_done) GEM_HIP(h, hipEventCreateWithFlags(&pb.tables_done, hipEventDisableTiming));
            if (pb.tables_recorded) GEM_HIP(h, hipEventSynchronize(pb.tables_done));     // the previous upload from this buffer has been read
            unsigned char* host = static_cast<unsigned char*>(pb.host_tables);
            memset(host, 0, total);
            // clouds whose frames all use the laser model (the reference's only GPU model, GPU:403-408) take the instantiation without
            // the camera models' double-precision code: 2; with every frame's rotation variance zero (height_variance, kModelLaserFast): 4
            bool laser = true, fast = true;
            for (int s = 0; s < in.n_sweeps; ++s) {
                FrameConst& fc = reinterpret_cast<FrameConst*>(host + o_frames)[s];
                fill_frame(h, &in.params[s], fc);
                laser = laser && in.params[s].sensor_model == GEM_MODEL_LASER;
                fast = fast && fc.fast_laser != 0;
            }
            pb.tab_src = laser ? (fast ? 4 : 2) : 0;
            memcpy(host + o_chunk0, chunk0.data(), sizeof(int) * (in.n_sweeps + 1));
            memcpy(host + o_first, in.offsets, sizeof(long long) * (in.n_sweeps + 1));
            if (in.var_updates) memcpy(host + o_var, in.var_updates, sizeof(float) * in.n_sweeps);
            if (in.sweep_orig0) memcpy(host + o_orig, in.sweep_orig0, sizeof(int) * in.n_sweeps);
            // on a stream of its own when the passes overlap: the upload (a 5 us blit + two kernel boundaries) then runs while the
            // binning stream is still sorting the pass before, instead of at the head of this pass's chain (the buffer's last
            // readers -- the pass before the previous one -- are done: fuse_done above)
            hipStream_t stab = sbin;
            if (overlap && h->tab_stream) stab = h->tab_stream;
            pb.tab_key.clear();                                        // (not valid until the upload is enqueued)
            GEM_HIP(h, hipMemcpyAsync(pb.tables.p, host, total, hipMemcpyHostToDevice, stab));
            GEM_HIP(h, hipEventRecord(pb.tables_done, stab)); pb.tables_recorded = true;
            if (stab != sbin) GEM_HIP(h, hipStreamWaitEvent(sbin, pb.tables_done, 0));
            pb.tab_key = h->key_scratch;
            pb.tab_upload_stream = stab;
        } else if (pb.tab_upload_stream != sbin && pb.tables_recorded) {
            // the cached upload ran on another stream than this pass's sort (the upload stream, or the other binning stream): long
            // done -- passes of this buffer set have run since -- but the order is stated, not assumed
            GEM_HIP(h, hipStreamWaitEvent(sbin, pb.tables_done, 0));
        }
        batch_src = pb.tab_src;
        unsigned char* d = static_cast<unsigned char*>(pb.tables.p);
        sa.frames = reinterpret_cast<const FrameConst*>(d + o_frames);
        sa.sweep_chunk0 = reinterpret_cast<const int*>(d + o_chunk0);
        sa.sweep_first = reinterpret_cast<const long long*>(d + o_first);
        sa.sweep_orig0 = in.sweep_orig0 ? reinterpret_cast<const int*>(d + o_orig) : nullptr;
        wa.var_updates = in.var_updates ? reinterpret_cast<const float*>(d + o_var) : nullptr;
    } else {
        fill_frame(h, in.src == 0 ? in.params : nullptr, sa.frame0);
        sa.orig0_single = in.sweep_orig0 ? in.sweep_orig0[0] : 0;
    }
    sa.n_sweeps = in.n_sweeps; sa.n = in.n; sa.sweep_id0 = shard ? shard->sweep_id0 : 0;
    sa.xyzi = in.xyzi; sa.rgb = in.rgb; sa.orig = in.orig;
    sa.f_index = in.f_index; sa.f_height = in.f_height; sa.f_var = in.f_var;
    sa.f_R = in.f_R; sa.f_G = in.f_G; sa.f_B = in.f_B; sa.f_I = in.f_I;
    sa.keep_sentinel = h->track_lowest ? 1 : 0;
    sa.rank_by_ballot = h->rank_by_ballot ? 1 : 0; sa.few_bins = h->few_bins;
    sa.tiles_per_row = geo.tiles_per_row; sa.T = T;
    sa.id_bits = geo.id_bits; sa.n_passes = geo.n_passes;
    for (int i = 0; i < 3; ++i) { sa.dshift[i] = geo.dshift[i]; sa.dbits[i] = geo.dbits[i]; sa.dbins[i] = geo.dbins[i]; }
    sa.n_chunks1 = NC1; sa.chunk = chunk;
    // Small two-pass sorts (a depth image: 300 k points, six launches of 5-10 us each) let pass 1's scatter count pass 2's digit with
    // atomics: one launch and one pass over the keys less (4.7 us of the chip per frame; the frame's period is its walk and does not
    // move).  Big passes keep k_sort_count: ten million device-scope atomics cost more than its 8 us (k_sort_project's block counts
    // were 4.4 ns each).
    constexpr long long kFuseCountMaxPoints = 600000;
    sa.fuse_count = (geo.n_passes >= 2 && (h->fuse_count == 2 || (h->fuse_count == 1 && in.n <= kFuseCountMaxPoints))) ? 1 : 0;
    unsigned char* misc = static_cast<unsigned char*>(pb.s_misc.p);
    for (int i = 0; i < geo.n_passes; ++i) {
        sa.cnt[i] = static_cast<uint32_t*>(i == 0 ? pb.s_cnt1.p : pb.s_cnt2.p);
        sa.segtot[i] = reinterpret_cast<uint32_t*>(misc + o_seg[i]);
    }
    sa.total = reinterpret_cast<uint32_t*>(misc + o_total); sa.bin_base = reinterpret_cast<uint32_t*>(misc + o_base);
    // (word 1 behind the record count: k_sort_project stores the pass's epoch there when a record is outside the plain range of
    //  the walks' chain loops; epochs never repeat, so the word needs no clearing)
    sa.odd_flag = sa.total + 1; sa.epoch = ++h->sort_epoch; if (sa.epoch == 0u) sa.epoch = ++h->sort_epoch;
    wa.odd_flag = sa.odd_flag; wa.epoch = sa.epoch;
    sa.blk_cnt = nullptr;
    if ((geo.block_form && geo.n_passes > 1) || shard) {             // the walk will want every block's range (the last pass's bins are not the blocks)
        if ((rc = ensure_zeroed(h, pb.s_blkcnt, (size_t)4 * T * sizeof(uint32_t))) || (rc = ensure(h, pb.s_ranges, (size_t)4 * T * sizeof(uint2)))) return rc;
        sa.blk_cnt = static_cast<uint32_t*>(pb.s_blkcnt.p);
        // the counts are zero between passes because k_block_prefix leaves them so; a pass that failed between the two leaves them
        // dirty: cleared here before the next one counts
        if (pb.blkcnt_dirty) GEM_HIP(h, hipMemsetAsync(pb.s_blkcnt.p, 0, pb.s_blkcnt.cap, sbin));
        pb.blkcnt_dirty = true;
    }
    sa.seg_cnt = reinterpret_cast<uint32_t*>(misc + o_segcnt);
    // arrays a: the projected records in input order, later the final order; arrays b: the order after pass 1
    sa.hv_a = static_cast<uint2*>(pb.s_hv2.p); sa.hv_b = static_cast<uint2*>(pb.s_hv1.p);
    sa.key_a = static_cast<uint32_t*>(pb.s_key2.p); sa.key_b = static_cast<uint32_t*>(pb.s_key1.p);
    sa.src_a = with_src ? static_cast<uint32_t*>(pb.s_src2.p) : nullptr; sa.src_b = with_src ? static_cast<uint32_t*>(pb.s_src1.p) : nullptr;
    sa.counters = h->counting ? h->d_counters : nullptr;

    const bool final_b = (geo.n_passes & 1) != 0;                     // the passes ping-pong between the arrays: a -> b -> a (-> b)
    wa.hv = final_b ? sa.hv_b : sa.hv_a; wa.key = final_b ? sa.key_b : sa.key_a; wa.src = final_b ? sa.src_b : sa.src_a; wa.bin_base = sa.bin_base;
    // centre rows first while (nearly) all of the walk's waves are resident at once: the start order then decides when the long
    // chains under the sensor begin (C4: 62 -> 52 us); a walk of many rounds reads its records front to back instead (C5:
    // 91 us in memory order, 100-120 us in any other)
    wa.walk_order = (h->walk_permute && 4ll * T <= 4096) ? 1 : 0;
    wa.center_tr = ((h->L / 2 + h->start[0]) % h->L) >> 5;
    wa.T = T; wa.tiles_per_row = geo.tiles_per_row; wa.L = h->L; wa.row0 = h->row0; wa.row1 = h->row1;
    wa.id_bits = geo.id_bits; wa.bin_shift = geo.dshift[geo.n_passes - 1]; wa.n_sweeps = in.n_sweeps;
    wa.exact_bins = (geo.block_form && geo.n_passes == 1) ? 1 : 0;
    wa.lane_sort = h->lane_sort ? 1 : 0;
    wa.light_blocks = h->blk_batch ? (h->blk_batch <= 512 ? 1 : 0) : ((long long)in.n <= 768ll * 4 * T ? 1 : 0);   // (by the mean: a heavy block just takes more rounds)
    if (wa.light_blocks) wa.lane_sort = 0;                             // (handing the busiest cells to wave 0 pays for blocks of thousands of records: C5 118 -> 114 us without)
    wa.mahal = h->cfg.mahalanobis_threshold; wa.var_floor = h->cfg.variance_floor;
    wa.dense = dense ? 1 : 0;
    wa.n_pending = h->n_pending;
    for (int i = 0; i < kMaxPending; ++i) wa.pending[i] = h->pending[i];
    wa.plain_env = h->plain_loop ? walk_plain_env(wa.var_floor, wa.mahal, h->pending, h->n_pending, batched ? in.var_updates : nullptr, in.n_sweeps) : 0;
    wa.prio_records = h->walk_prio; wa.lds_pad = h->walk_lds_pad; wa.light_fast = h->light_fast ? 1 : 0;
    wa.elevation = h->layers.elevation; wa.variance = h->layers.variance; wa.lowest = h->layers.lowest;
    wa.start0 = h->start[0]; wa.start1 = h->start[1];
    wa.intensity = h->layers.intensity; wa.colorR = h->layers.colorR; wa.colorG = h->layers.colorG; wa.colorB = h->layers.colorB;
    wa.xyzi = in.xyzi; wa.rgb = in.rgb; wa.f_R = in.f_R; wa.f_G = in.f_G; wa.f_B = in.f_B; wa.f_I = in.f_I;
    wa.counters = sa.counters;
    wa.count_per_pass = batched ? 0 : 1;

    if (h->counting) GEM_HIP(h, hipMemsetAsync(h->d_counters, 0, 2 * sizeof(unsigned long long), h->stream));
    bool ride_bin = false;
    {
        // (a third pass is accounted with the second: count / scan / scatter of the higher digits)
        // (event pairs only for the kernels that are launched: the elapsed time of a pair that was never recorded is an error)
        const bool two = geo.n_passes >= 2, three = geo.n_passes == 3;
        Timed t0(h, 3), t1(h, 4), t2(h, 5), t3(h, two && !sa.fuse_count ? 6 : -1), t4(h, two ? 7 : -1), t5(h, two ? 8 : -1), t6(h, three ? 6 : -1), t7(h, three ? 7 : -1), t8(h, three ? 8 : -1);
        LaunchEvents ev[9] = {t0.events(), t1.events(), t2.events(), t3.events(), t4.events(), t5.events(), t6.events(), t7.events(), t8.events()};
        // The walk waits for the sort across streams: as the STOP EVENT of the sort's last dispatch the event is seen 3 us earlier
        // than a marker recorded behind it (tools/ubench/handover.hip: 7 against 10 us) -- when that kernel is the last thing on the
        // sort's stream before the walk (no k_block_prefix, no strip search behind it) and nothing is being timed.
        ride_bin = overlap && h->ride_events && !h->timing && !shard && !(geo.block_form && geo.n_passes > 1);
        if (ride_bin) ev[3 * geo.n_passes - 1].stop = pb.bin_done;
        int src = in.src;
        if (src == 0 && batched) { if (batch_src > 0) src = batch_src; }
        else if (src == 0) {
            if (in.params[0].sensor_model == GEM_MODEL_LASER) src = sa.frame0.fast_laser ? 4 : 2;      // 4: the rotation variance is zero (height_variance, kModelLaserFast)
        }
        GEM_HIP(h, launch_sort(sbin, sa, src, with_src, ev));
    }
    if (shard) {
        // where the strips begin in the sorted records (one 32-ary search per boundary) and where every block's records are
        // (k_block_prefix): behind the sort, on its stream
        gem_handle::Shard& sd = h->shard;
        sd.valid = false;
        if (!h->sh_host) GEM_HIP(h, hipHostMalloc(&h->sh_host, kShardHostBytes, hipHostMallocDefault));
        if ((rc = ensure(h, pb.s_shard, 64 * sizeof(uint32_t)))) return rc;
        uint32_t* host = static_cast<uint32_t*>(h->sh_host);
        for (int k = 0; k <= shard->nstrips; ++k) {
            const int tile_row = shard->strip_rows[k] >= h->L ? geo.tiles_per_row : shard->strip_rows[k] / 32;
            host[k] = (uint32_t)(tile_row * geo.tiles_per_row) << 10;                // first cell id of the strip (the same every call)
        }
        uint32_t* d_ids = static_cast<uint32_t*>(pb.s_shard.p), *d_bounds = d_ids + 16;
        const uint32_t* keys = final_b ? sa.key_b : sa.key_a;
        GEM_HIP(h, hipMemcpyAsync(d_ids, host, sizeof(uint32_t) * (shard->nstrips + 1), hipMemcpyHostToDevice, sbin));
        GEM_HIP(h, launch_strip_bounds(sbin, keys, sa.total, geo.id_bits, d_ids, d_bounds, shard->nstrips + 1));
        GEM_HIP(h, launch_block_prefix(sbin, sa.blk_cnt, 4 * T, static_cast<uint2*>(pb.s_ranges.p)));
        pb.blkcnt_dirty = false;
        sd.hv = final_b ? sa.hv_b : sa.hv_a; sd.key = keys; sd.ranges = static_cast<const uint2*>(pb.s_ranges.p);
        sd.d_bounds = d_bounds; sd.nstrips = shard->nstrips; sd.slot = overlap ? (int)slot : -1; sd.stream = sbin;
        h->stats.points_in = in.n;
        if (shard->bounds_stay_on_device) {                  // gem_add_sharded_device all-gathers them from where they are
            for (int k = 0; k <= shard->nstrips; ++k) sd.bounds[k] = 0;
            if (overlap) GEM_HIP(h, hipEventRecord(pb.bin_done, sbin));
            sd.valid = true;
            return GEM_OK;
        }
        GEM_HIP(h, hipMemcpyAsync(host + 32, d_bounds, sizeof(uint32_t) * (shard->nstrips + 1), hipMemcpyDeviceToHost, sbin));
        GEM_HIP(h, hipStreamSynchronize(sbin));
        for (int k = 0; k <= shard->nstrips; ++k) sd.bounds[k] = host[32 + k];
        sd.valid = true;
        return GEM_OK;
    }
    if (geo.block_form && geo.n_passes > 1) {
        // the last digit's bins hold several blocks: where every block's records are (the prefix of the per-block counts
        // k_sort_project took), behind the sort on its stream, instead of a search by every workgroup of the walk
        GEM_HIP(h, launch_block_prefix(sbin, sa.blk_cnt, 4 * T, static_cast<uint2*>(pb.s_ranges.p)));
        pb.blkcnt_dirty = false;
        wa.ranges = static_cast<const uint2*>(pb.s_ranges.p);
    }
    { const int rcd = flush_walk(h); if (rcd) return rcd; }           // the pass before: its sort has had this call's launches to finish
    if (leave_walk) {
        if (!ride_bin) GEM_HIP(h, hipEventRecord(pb.bin_done, sbin));
        h->dwalk.wa = wa; h->dwalk.block_form = geo.block_form; h->dwalk.attr = attr; h->dwalk.slot = slot; h->dwalk.valid = true;
        ++h->walks_left;
        h->dbg_rows = 0;
        h->n_pending = 0;
        h->floor_dirty = false;
        h->stats.points_in = in.n;
        return GEM_OK;
    }
    if (overlap) {
        if (!ride_bin) GEM_HIP(h, hipEventRecord(pb.bin_done, sbin));
        GEM_HIP(h, hipStreamWaitEvent(h->stream, pb.bin_done, 0));
    }
    h->dbg_rows = 0;
    if (h->dbg_on && geo.block_form) {
        if ((rc = ensure(h, h->dbg, (size_t)T * 4 * 16 * 8))) return rc;
        GEM_HIP(h, hipMemsetAsync(h->dbg.p, 0, (size_t)T * 4 * 16 * 8, h->stream));
        wa.dbg = static_cast<unsigned long long*>(h->dbg.p);
        h->dbg_rows = T * 4;
    }
    { Timed t(h, 9); GEM_HIP(h, geo.block_form ? launch_block_walk(h->stream, wa, attr, t.events()) : launch_walk(h->stream, wa, attr, t.events())); }
    if (overlap) { GEM_HIP(h, hipEventRecord(pb.fuse_done, h->stream)); pb.fuse_recorded = true; }
    h->n_pending = 0;
    h->floor_dirty = false;
    h->stats.points_in = in.n;
    return GEM_OK;
}

int run_pipeline(gem_handle* h, const PassInput& in0)
{
    // Big passes (batches of sweeps, aggregated clouds, depth images) go through the sorted pipeline: a global two-digit counting
    // sort of the in-map points by (tile, cell), then one walk per cell (gem_sort.hip).  Small ones -- a single LiDAR sweep -- keep
    // the tile pipeline below, whose one or two launches cost less than the sort's seven.
    // Measured crossover (tools/dbg/crossover.py): batches of LiDAR sweeps -- a few points per cell and sweep -- are faster on the tile
    // pipeline up to about 8 sweeps (1 M points); a single dense cloud (a depth image: hundreds of points per cell) from ~150 k points.
    const long long sort_from = in0.n_sweeps > 1 ? h->sort_min_points_batch : h->sort_min_points;
    if (h->sort_path && in0.n >= sort_from && in0.n < (1ll << 31)) {
        int attr = 0;
        if (in0.src == 0 && in0.rgb) attr = 1;
        if (in0.src == 1 && in0.f_R && in0.f_G && in0.f_B && in0.f_I) attr = 2;
        if (h->track_lowest) attr |= 4;
        // Batches of sweeps -- a few records per cell and sweep, every batch of a block's records spread over its cells -- take the
        // block-sorted form (one counting-sort pass for the 600^2 map instead of two, no per-cell order in HBM at all); a single
        // dense cloud (a depth image: a quarter of its points in one block, hundreds per cell, image row by image row) needs the
        // whole chip to order it by cell: the cell-sorted form.
        // (Maps of more than kOnePassMaxBins blocks take two passes over the block id and k_block_prefix; with k_fuse_block's rounds
        //  of 512 records for light blocks that is still the shorter way -- C5, 2400^2, same box: 351-365 us cell-sorted in three
        //  passes, 333-340 block-sorted in two.)
        const bool block_form = h->sort_form == 2 || (h->sort_form == 0 && in0.n_sweeps > 1);
        SortGeometry geo = sort_geometry(h, in0.n_sweeps, block_form);
        if (!geo.ok) { geo = sort_geometry(h, in0.n_sweeps, !block_form); ++h->sort_fallbacks; }    // (a forced form / pass count that does not fit this map: counted, gem_debug_get)
        if (geo.ok) return run_sort_pipeline(h, in0, attr, geo);
    }
    { const int rcw = flush_walk(h); if (rcw) return rcw; }           // (a sorted pass's walk still to be launched: before anything of this pass fuses)
    // A big single cloud becomes a batch of sweeps with one frame: every tile then only reads the descriptor
    // rows of the sweeps that reach it (flag[tile][sweep]) instead of one row over all units.  The
    // recurrence is unchanged: the per-sweep variance floor is idempotent with the floor at the start of every
    // step (GPU:500-501), and no variance increment is applied between these sweeps.
    PassInput in = in0;
    std::vector<gem_frame_params> cut_params;
    std::vector<long long> cut_offsets;
    std::vector<int> orig0;
    // (Fuse's arrays too, src == 1: a descriptor row holds the units of ONE sweep, k_fuse_list reads one chunk of kChunkUnits of it --
    //  until round 4 the cut was only made for clouds, and a Fuse of more than 131 072 points that stayed below the sorted pipeline's
    //  threshold lost every point behind the first 131 072.)
    if (in.n_sweeps == 1 && in.n > kSweepPoints) {
        const int ns = (int)((in.n + kSweepPoints - 1) / kSweepPoints);
        if (in.src == 0) cut_params.assign(ns, *in.params);
        cut_offsets.resize(ns + 1); orig0.resize(ns);
        for (int s = 0; s <= ns; ++s) cut_offsets[s] = std::min<long long>(in.n, (long long)s * kSweepPoints);
        for (int s = 0; s < ns; ++s) orig0[s] = (int)cut_offsets[s];
        in.n_sweeps = ns; in.params = in.src == 0 ? cut_params.data() : nullptr; in.offsets = cut_offsets.data(); in.var_updates = nullptr;
    }
    const bool batched = in.n_sweeps > 1;
    const int U = kUnit;

    // units per sweep
    std::vector<int> unit0(in.n_sweeps + 1, 0);
    int bpad = 0;
    for (int s = 0; s < in.n_sweeps; ++s) {
        const long long cnt = batched ? in.offsets[s + 1] - in.offsets[s] : in.n;
        long long units = (cnt + U - 1) / U;
        units = (units + 31) & ~31ll;               // descriptor rows are flagged in groups of 32 units (64 B)
        if (units > 0x3fffffff) return fail(h, GEM_ERR_INVALID, "cloud too large");
        unit0[s + 1] = unit0[s] + (int)units;
        bpad = std::max(bpad, (int)units);
    }
    const int B = unit0[in.n_sweeps];
    const bool dense = h->n_pending > 0 || h->floor_dirty || (batched && in.var_updates != nullptr);

    if (B == 0) {
        // Fuse with zero points still runs the floor pass (gpu_process.cu:533-534)
        if (batched && in.var_updates)
            for (int s = 0; s < in.n_sweeps; ++s) {
                if (h->n_pending == kMaxPending) { int rc = flush_pending(h, true); if (rc) return rc; }
                h->pending[h->n_pending++] = in.var_updates[s];
            }
        return (h->n_pending || h->floor_dirty) ? flush_pending(h, true) : GEM_OK;
    }
    int attr = 0;
    if (in.src == 0 && in.rgb) attr = 1;
    if (in.src == 1 && in.f_R && in.f_G && in.f_B && in.f_I) attr = 2;
    if (h->track_lowest) attr |= 4;                  // the kernel variants that also maintain map_lowest (16x16 tiles)
    // tile size of this pass: 16x16 cells (more, lighter workgroups: better balance and latency hiding)
    // unless the [sweep][tile][unit] descriptor table would get too big, then 32x32
    int ts = h->ts;
    {
        const long long tpr4 = (h->L + 15) / 16;
        const long long table4 = tpr4 * tpr4 * (long long)bpad * in.n_sweeps * (long long)sizeof(uint16_t);
        if (ts == 0) ts = table4 <= (1ll << 29) ? 4 : 5;
        // the kernel variants that maintain map_lowest exist for 16x16 tiles only: the choice is made HERE, before the tile
        // geometry (te, tiles_per_row, T, table sizes) is derived from it
        if (h->track_lowest) {
            if (table4 > (16ll << 30)) return fail(h, GEM_ERR_INVALID, "lowest tracking: the pass is too large for 16x16 tiles (cut it into smaller calls)");
            ts = 4;
        }
    }
    const int te = 1 << ts;
    const int tiles_per_row = (h->L + te - 1) / te;
    const int T = tiles_per_row * tiles_per_row;
    h->T = T;
    if (fuse_lds_bytes(ts, h->fuse_variant, attr & 3) > 160 * 1024) return fail(h, GEM_ERR_INVALID, "fuse kernel geometry exceeds the LDS");

    // k_bin of this pass may run on its own stream, concurrently with the k_fuse of the previous pass
    // (it depends on the cloud and the pose, not on the map).  Only with the handle's own stream:
    // a caller-provided stream keeps everything in order on that stream.  Device-resident inputs
    // must be complete when the call is made (they are not ordered against the handle's streams).
    // The cross-stream event pair costs ~3 us per pass (measured), so it only pays for big passes
    // (batches / aggregated clouds: C4 379 -> 313 us); single sweeps stay on one stream.
    bool overlap = h->overlap && in.n >= h->overlap_min_points && h->stream == h->own_stream && !h->counting && !h->dbg_on;
    // one launch per frame for a stream of single sweeps (k_frame): needs the other half of the double buffer
    const bool defer = h->defer && in.device_input && in.src == 0 && !batched && (attr & 3) == 0 && ts == 4 && !overlap &&
                       !h->counting && (!h->dbg_on || h->dbg_frame);
    if (!defer) { const int rcd = flush_deferred(h); if (rcd) return rcd; }
    gem_handle::PassBuffers& pb = h->pb[(overlap || defer) ? (h->pass++ & 1u) : 0u];
    hipStream_t sbin = overlap ? h->bin_stream : h->stream;
    if (overlap && h->main_reads_pb) {
        // Passes that ran entirely on the handle's stream (single sweeps, k_frame, a flushed deferred fuse) read either half of
        // the double buffer without recording a per-half event.  Before k_bin on the other stream may overwrite a half, that
        // stream waits for everything enqueued on the handle's stream so far (one event at the switch, none per frame).
        GEM_HIP(h, hipEventRecord(h->switch_done, h->stream));
        GEM_HIP(h, hipStreamWaitEvent(h->bin_stream, h->switch_done, 0));
        if (h->bin_stream2) GEM_HIP(h, hipStreamWaitEvent(h->bin_stream2, h->switch_done, 0));
        if (h->tab_stream) GEM_HIP(h, hipStreamWaitEvent(h->tab_stream, h->switch_done, 0));      // (the batch tables of a pass buffer are uploaded there)
        h->main_reads_pb = false;
        for (auto& b : h->pb) b.fuse_recorded = false;      // covered by the wait above
    }
    if (!overlap) h->main_reads_pb = true;
    int rc;
    if ((rc = ensure(h, pb.rec, (size_t)B * U * sizeof(uint4)))) return rc;
    if ((rc = ensure(h, pb.srt, (size_t)B * U * sizeof(uint4) + 16))) return rc;     // sorted arena + its bump pointer (last 16 bytes)
    // k_fuse of pass p-2 has read these buffers.  Waited for on the HOST: a hipStreamWaitEvent on an event that is still
    // far from complete delayed the start of k_bin behind it (C5: 1.52 -> 1.64-1.81 ms per pass, the overlap mostly lost);
    // the host stays at most two (big) passes ahead of the device, which costs nothing.
    if (overlap && pb.fuse_recorded) GEM_HIP(h, hipEventSynchronize(pb.fuse_done));
    {   // descriptor table [sweep][tile][unit in sweep]: k_fuse_list zeroes what it consumes, so the table only
        // has to be cleared when it is (re)allocated
        const size_t need = (size_t)in.n_sweeps * T * bpad * sizeof(uint16_t);
        if (need > pb.seg.cap) {
            if ((rc = ensure(h, pb.seg, need))) return rc;
            GEM_HIP(h, hipMemsetAsync(pb.seg.p, 0, pb.seg.cap, sbin));
        }
        // touched flags [tile][sweep]: stamped with the pass's epoch instead of being cleared
        const size_t need_flag = (size_t)T * in.n_sweeps * sizeof(uint32_t);
        const size_t need_gflag = (size_t)in.n_sweeps * T * (bpad / 32) * sizeof(uint32_t);
        const bool grow_flag = need_flag > pb.flag.cap || need_gflag > pb.gflag.cap;
        if ((rc = ensure(h, pb.flag, need_flag))) return rc;
        if ((rc = ensure(h, pb.gflag, need_gflag))) return rc;
        if (grow_flag || pb.epoch >= kFlagEpochMax) {
            GEM_HIP(h, hipMemsetAsync(pb.flag.p, 0, pb.flag.cap, sbin));
            GEM_HIP(h, hipMemsetAsync(pb.gflag.p, 0, pb.gflag.cap, sbin));
            pb.epoch = 0;
        }
        ++pb.epoch;
    }

    BinArgs ba{};
    FuseArgs fa{};
    if (batched) {
        // tables: frames | unit0 | first | orig0 | var_updates
        const size_t o_frames = 0;
        const size_t o_unit0 = o_frames + sizeof(FrameConst) * in.n_sweeps;
        const size_t o_first = (o_unit0 + sizeof(int) * (in.n_sweeps + 1) + 15) & ~(size_t)15;
        const size_t o_orig = o_first + sizeof(long long) * (in.n_sweeps + 1);
        const size_t o_var = o_orig + sizeof(int) * in.n_sweeps;
        const size_t total = o_var + sizeof(float) * in.n_sweeps;
        pb.tab_key.clear();                                          // (the sorted pipeline's cached tables of this buffer set are overwritten below)
        if ((rc = ensure(h, pb.tables, total))) return rc;
        // staged in pinned memory so that the upload does not make the host wait for the stream (a pageable source would:
        // the call then cost a whole k_bin of host time, 240 us per C4 batch)
        if (total > pb.host_cap) {
            if (pb.tables_recorded) GEM_HIP(h, hipEventSynchronize(pb.tables_done));
            if (pb.host_tables) GEM_HIP(h, hipHostFree(pb.host_tables));
            pb.host_tables = nullptr; pb.host_cap = 0;
            GEM_HIP(h, hipHostMalloc(&pb.host_tables, total * 2, hipHostMallocDefault));
            pb.host_cap = total * 2;
        }
        if (!pb.tables_done) GEM_HIP(h, hipEventCreateWithFlags(&pb.tables_done, hipEventDisableTiming));
        if (pb.tables_recorded) GEM_HIP(h, hipEventSynchronize(pb.tables_done));     // the previous upload from this buffer has been read
        unsigned char* host = static_cast<unsigned char*>(pb.host_tables);
        memset(host, 0, total);
        for (int s = 0; s < in.n_sweeps; ++s) fill_frame(h, in.src == 0 ? &in.params[s] : nullptr, reinterpret_cast<FrameConst*>(host + o_frames)[s]);
        memcpy(host + o_unit0, unit0.data(), sizeof(int) * (in.n_sweeps + 1));
        memcpy(host + o_first, in.offsets, sizeof(long long) * (in.n_sweeps + 1));
        if (!orig0.empty()) memcpy(host + o_orig, orig0.data(), sizeof(int) * in.n_sweeps);
        if (in.var_updates) memcpy(host + o_var, in.var_updates, sizeof(float) * in.n_sweeps);
        pb.tab_key.clear();                                  // (the sorted pipeline's cached tables of this buffer set are overwritten)
        GEM_HIP(h, hipMemcpyAsync(pb.tables.p, host, total, hipMemcpyHostToDevice, sbin));
        GEM_HIP(h, hipEventRecord(pb.tables_done, sbin)); pb.tables_recorded = true;
        unsigned char* d = static_cast<unsigned char*>(pb.tables.p);
        ba.frames = reinterpret_cast<const FrameConst*>(d + o_frames);
        ba.sweep_unit0 = reinterpret_cast<const int*>(d + o_unit0);
        ba.sweep_first = reinterpret_cast<const long long*>(d + o_first);
        ba.sweep_orig0 = orig0.empty() ? nullptr : reinterpret_cast<const int*>(d + o_orig);
        fa.sweep_unit0 = ba.sweep_unit0;
        fa.var_updates = in.var_updates ? reinterpret_cast<const float*>(d + o_var) : nullptr;
    } else {
        fill_frame(h, in.src == 0 ? in.params : nullptr, ba.frame0);
    }
    ba.n_sweeps = in.n_sweeps; ba.n = in.n;
    ba.xyzi = in.xyzi; ba.rgb = in.rgb; ba.orig = in.orig;
    ba.f_index = in.f_index; ba.f_height = in.f_height; ba.f_var = in.f_var;
    ba.f_R = in.f_R; ba.f_G = in.f_G; ba.f_B = in.f_B; ba.f_I = in.f_I;
    ba.T = T; ba.tiles_per_row = tiles_per_row; ba.B = B; ba.Bpad = bpad;
    ba.tile_bits = 0; while ((1 << ba.tile_bits) < T) ++ba.tile_bits;
    ba.epoch = pb.epoch;
    ba.rec_words = (attr & 3) != 0 ? 4 : 3;
    ba.rec = static_cast<uint4*>(pb.rec.p); ba.seg = static_cast<uint16_t*>(pb.seg.p); ba.flag = static_cast<uint32_t*>(pb.flag.p); ba.gflag = static_cast<uint32_t*>(pb.gflag.p);
    ba.counters = h->counting ? h->d_counters : nullptr;
    ba.keep_sentinel = h->track_lowest ? 1 : 0;
    ba.srt_top = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(pb.srt.p) + pb.srt.cap - 16);

    fa.epoch = pb.epoch;
    fa.rec = ba.rec; fa.seg = ba.seg; fa.flag = ba.flag; fa.gflag = ba.gflag; fa.B_total = B; fa.U = U; fa.n_sweeps = in.n_sweeps; fa.Bpad = bpad;
    fa.T = T; fa.tiles_per_row = tiles_per_row; fa.L = h->L; fa.row0 = h->row0; fa.row1 = h->row1;
    fa.center_tr = ((h->L / 2 + h->start[0]) % h->L) >> ts; fa.center_tc = ((h->L / 2 + h->start[1]) % h->L) >> ts;
    fa.mahal = h->cfg.mahalanobis_threshold; fa.var_floor = h->cfg.variance_floor;
    fa.dense = dense ? 1 : 0;
    fa.n_pending = h->n_pending;
    for (int i = 0; i < kMaxPending; ++i) fa.pending[i] = h->pending[i];
    fa.elevation = h->layers.elevation; fa.variance = h->layers.variance;
    fa.intensity = h->layers.intensity; fa.colorR = h->layers.colorR; fa.colorG = h->layers.colorG; fa.colorB = h->layers.colorB;
    fa.xyzi = in.xyzi; fa.rgb = in.rgb; fa.f_R = in.f_R; fa.f_G = in.f_G; fa.f_B = in.f_B; fa.f_I = in.f_I;
    fa.counters = ba.counters;
    fa.srt = static_cast<uint4*>(pb.srt.p); fa.srt_top = ba.srt_top; fa.dense_min = h->dense_min;
    fa.lowest = h->layers.lowest; fa.start0 = h->start[0]; fa.start1 = h->start[1];
    fa.count_per_pass = orig0.empty() ? 0 : 1;
    fa.dbg = nullptr;
    fa.dbg_sweep = h->dbg_sweep;
    if (h->dbg_on) {
        // rows [0, T): the tiles' stamps; [T, T + binning blocks): the binning blocks' (k_frame with "dbg_frame": both halves of one launch)
        const int nbin = (B + 3) / 4;
        if ((rc = ensure(h, h->dbg, (size_t)(T + nbin) * 16 * 8))) return rc;
        GEM_HIP(h, hipMemsetAsync(h->dbg.p, 0, (size_t)(T + nbin) * 16 * 8, h->stream));
        fa.dbg = static_cast<unsigned long long*>(h->dbg.p);
        if (h->dbg_frame) { ba.dbg = fa.dbg + (size_t)T * 16; h->dbg_rows = T + nbin; }
    }

    if (defer) {
        if (h->deferred.valid && h->deferred.attr != attr) { const int rcd = flush_deferred(h); if (rcd) return rcd; }   // (cannot happen: toggling the tracking flushes)
        if (h->deferred.valid) { Timed t(h, 2); GEM_HIP(h, launch_frame(h->stream, h->deferred.fa, ba, attr, t.events())); }
        else                   { Timed t(h, 0); GEM_HIP(h, launch_bin(h->stream, ba, in.src, ts, t.events())); }
        h->deferred.fa = fa; h->deferred.ts = ts; h->deferred.attr = attr; h->deferred.valid = true;
        h->n_pending = 0;
        h->floor_dirty = false;
        h->stats.points_in = in.n;
        return GEM_OK;
    }
    if (h->counting) GEM_HIP(h, hipMemsetAsync(h->d_counters, 0, 2 * sizeof(unsigned long long), h->stream));
    { Timed t(h, 0); GEM_HIP(h, launch_bin(sbin, ba, in.src, ts, t.events())); }
    if (overlap) {
        GEM_HIP(h, hipEventRecord(pb.bin_done, sbin));
        GEM_HIP(h, hipStreamWaitEvent(h->stream, pb.bin_done, 0));
    }
    { Timed t(h, 1); GEM_HIP(h, launch_fuse(h->stream, fa, ts, attr, h->fuse_variant, t.events())); }
    if (overlap) { GEM_HIP(h, hipEventRecord(pb.fuse_done, h->stream)); pb.fuse_recorded = true; }
    h->n_pending = 0;
    h->floor_dirty = false;
    h->stats.points_in = in.n;
    return GEM_OK;
}

} // namespace

// ================================================================================================
extern "C" {

int gem_abi_version(void) { return GEM_ABI_VERSION; }

const char* gem_last_error(const gem_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int gem_create(const gem_map_config* cfg, gem_handle** out)
{
    if (!cfg || !out) return fail(nullptr, GEM_ERR_INVALID, "gem_create: null argument");
    *out = nullptr;
    if (cfg->length <= 0 || cfg->length > 32768 || !(cfg->resolution > 0.f))
        return fail(nullptr, GEM_ERR_INVALID, "gem_create: bad length / resolution");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, GEM_ERR_NO_DEVICE, "gem_create: no HIP device (libgem_hip has no CPU fallback)", e);
    int dev = cfg->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev) return fail(nullptr, GEM_ERR_INVALID, "gem_create: device ordinal out of range");
    if ((e = hipSetDevice(dev)) != hipSuccess) return fail(nullptr, GEM_ERR_NO_DEVICE, "hipSetDevice", e);

    gem_handle* h = new gem_handle();
    h->device = dev;
    h->cfg = *cfg;
    if (!(h->cfg.variance_floor > 0.f)) h->cfg.variance_floor = 0.0001f;          // gpu_process.cu:500
    if (!(h->cfg.mahalanobis_threshold > 0.f)) h->cfg.mahalanobis_threshold = 5.f; // gpu_process.cu:504
    h->L = cfg->length; h->cells = cfg->length * cfg->length; h->res = cfg->resolution;
    h->row0 = 0; h->row1 = h->L;
    if (cfg->strip_rows > 0) {
        if (cfg->strip_row0 < 0 || cfg->strip_row0 + cfg->strip_rows > h->L) { delete h; return fail(nullptr, GEM_ERR_INVALID, "gem_create: bad strip"); }
        h->row0 = cfg->strip_row0; h->row1 = cfg->strip_row0 + cfg->strip_rows;
    }
    h->fuse_variant = 12;                       // k_fuse_list geometry on 32x32 tiles: 10 = 256 threads, 11 = 512, 12 = 512 with 2048-record batches (2 per CU)
    h->ts = 0;                                  // 0: chosen per pass (run_pipeline)

    auto bail = [&](const char* what, hipError_t err) { int rc = fail(nullptr, GEM_ERR_HIP, what, err); gem_destroy(h); return rc; };
    if ((e = acquire_streams(dev, h->streams)) != hipSuccess) return bail("hipStreamCreate", e);
    h->own_stream = h->streams.s[0]; h->bin_stream = h->streams.s[1]; h->bin_stream2 = h->streams.s[2]; h->tab_stream = h->streams.s[3];
    h->stream = h->own_stream;
    if ((e = hipEventCreateWithFlags(&h->copy_done, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&h->switch_done, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    // (the handle's four streams come from the process-wide pool as a set, see acquire_streams: ROCm maps streams onto a few
    //  hardware queues, and streams that share one serialise)
    for (auto& b : h->pb) {
        if ((e = hipEventCreateWithFlags(&b.bin_done, kDeviceEventFlags)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&b.fuse_done, kDeviceEventFlags)) != hipSuccess) return bail("hipEventCreate", e);
    }
    // one allocation for the 8 layers (gpu_process.cu:954-961 uses 8 cudaMalloc)
    void* base = nullptr;
    const size_t layer_bytes = ((size_t)h->cells * 4 + 255) & ~(size_t)255;
    if ((e = hipMalloc(&base, layer_bytes * GEM_LAYER_COUNT)) != hipSuccess) return bail("hipMalloc(layers)", e);
    unsigned char* b = static_cast<unsigned char*>(base);
    h->layers.elevation = reinterpret_cast<float*>(b + layer_bytes * GEM_LAYER_ELEVATION);
    h->layers.variance  = reinterpret_cast<float*>(b + layer_bytes * GEM_LAYER_VARIANCE);
    h->layers.intensity = reinterpret_cast<float*>(b + layer_bytes * GEM_LAYER_INTENSITY);
    h->layers.traver    = reinterpret_cast<float*>(b + layer_bytes * GEM_LAYER_TRAVER);
    h->layers.lowest    = reinterpret_cast<float*>(b + layer_bytes * GEM_LAYER_LOWEST);
    h->layers.colorR    = reinterpret_cast<int*>(b + layer_bytes * GEM_LAYER_COLOR_R);
    h->layers.colorG    = reinterpret_cast<int*>(b + layer_bytes * GEM_LAYER_COLOR_G);
    h->layers.colorB    = reinterpret_cast<int*>(b + layer_bytes * GEM_LAYER_COLOR_B);
    h->layers.rough     = reinterpret_cast<float*>(b + layer_bytes * GEM_LAYER_ROUGH);
    h->layers.slope     = reinterpret_cast<float*>(b + layer_bytes * GEM_LAYER_SLOPE);
    if ((e = hipMalloc(reinterpret_cast<void**>(&h->d_counters), 2 * sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc(counters)", e);
    if ((e = launch_init(h->stream, h->layers, h->cells, 1)) != hipSuccess) return bail("k_init", e);   // G_Init_map
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return bail("hipStreamSynchronize", e);
    h->floor_dirty = true;
    *out = h;
    return GEM_OK;
}

void gem_destroy(gem_handle* h)
{
    if (!h) return;
    hipSetDevice(h->device);
    if (h->step.valid) {
        // a step whose second half never came (the caller did not synchronise): dropped, and the communicators go down with it so
        // that peers that do finish theirs fail instead of waiting for this rank
        h->step.valid = false;
        if (h->tp_x) h->tp_x->abort();
        if (h->tp_g) h->tp_g->abort();
    }
    if (h->stream) { settle(h); hipStreamSynchronize(h->stream); }
    if (h->bin_stream) hipStreamSynchronize(h->bin_stream);
    if (h->bin_stream2) hipStreamSynchronize(h->bin_stream2);
    if (h->tab_stream) hipStreamSynchronize(h->tab_stream);
    if (h->comm_stream) hipStreamSynchronize(h->comm_stream);
    if (h->gather_stream) hipStreamSynchronize(h->gather_stream);
    h->tp_g.reset(); h->tp_x.reset();                  // (the gather communicator first: it may be a view of the exchange communicator)
    release_comm_stream(h->device, h->comm_stream);
    release_comm_stream(h->device, h->gather_stream);
    for (hipEvent_t e : {h->ev_sorted, h->ev_exchanged, h->ev_bounds[0], h->ev_bounds[1], h->ev_walked[0], h->ev_walked[1], h->ev_vu[0], h->ev_vu[1],
                         h->ev_published[0], h->ev_published[1], h->ev_gathered[0], h->ev_gathered[1]}) if (e) hipEventDestroy(e);
    for (hipEvent_t e : h->ev_t) if (e) hipEventDestroy(e);
    fold_events(h);
    for (auto& ep : h->pool) { hipEventDestroy(ep.a); hipEventDestroy(ep.b); }
    if (h->layers.elevation) hipFree(h->layers.elevation);      // base of the single layer allocation
    if (h->d_counters) hipFree(h->d_counters);
    for (Arena* a : {&h->stage, &h->scratch, &h->dbg, &h->color, &h->ray, &h->sh_dev, &h->sh_recv_hv[0], &h->sh_recv_key[0], &h->sh_recv_rng[0],
                     &h->sh_recv_hv[1], &h->sh_recv_key[1], &h->sh_recv_rng[1], &h->sh_ranges, &h->published[0], &h->published[1]}) if (a->p) hipFree(a->p);
    if (h->sh_host) hipHostFree(h->sh_host);
    for (auto& b : h->pb) {
        for (Arena* a : {&b.rec, &b.srt, &b.seg, &b.flag, &b.gflag, &b.tables, &b.s_hv1, &b.s_hv2, &b.s_key1, &b.s_key2, &b.s_src1, &b.s_src2,
                         &b.s_cnt1, &b.s_cnt2, &b.s_misc, &b.s_ranges, &b.s_shard, &b.s_blkcnt}) if (a->p) hipFree(a->p);
        if (b.host_tables) hipHostFree(b.host_tables);
        if (b.tables_done) hipEventDestroy(b.tables_done);
        if (b.bin_done) hipEventDestroy(b.bin_done);
        if (b.fuse_done) hipEventDestroy(b.fuse_done);
    }
    if (h->copy_done) hipEventDestroy(h->copy_done);
    if (h->hstage) hipHostFree(h->hstage);
    for (auto& ev : h->ev_stage) if (ev) hipEventDestroy(ev);
    if (h->stage_read) hipEventDestroy(h->stage_read);
    for (auto& ev : h->ev_half) if (ev) hipEventDestroy(ev);
    if (h->switch_done) hipEventDestroy(h->switch_done);
    release_streams(h->device, h->streams);            // back to the pool, as a set
    delete h;
}

int gem_set_stream(gem_handle* h, void* hip_stream)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    GEM_HIP(h, hipStreamSynchronize(h->stream));
    if (h->bin_stream) GEM_HIP(h, hipStreamSynchronize(h->bin_stream));
    if (h->bin_stream2) GEM_HIP(h, hipStreamSynchronize(h->bin_stream2));
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return GEM_OK;
}

// Stream-ordered inputs: everything this handle enqueues from now on -- on its own stream and on its binning stream -- waits
// for `hip_event` (recorded by the caller on whatever stream produces the device buffers it is about to pass).
int gem_wait_event(gem_handle* h, void* hip_event)
{
    if (!h || !hip_event) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    hipEvent_t ev = static_cast<hipEvent_t>(hip_event);
    GEM_HIP(h, hipStreamWaitEvent(h->stream, ev, 0));
    if (h->bin_stream) GEM_HIP(h, hipStreamWaitEvent(h->bin_stream, ev, 0));
    if (h->bin_stream2) GEM_HIP(h, hipStreamWaitEvent(h->bin_stream2, ev, 0));
    // (a binning stream created later starts behind an event recorded on the handle's stream: see main_reads_pb)
    h->main_reads_pb = true;
    return GEM_OK;
}

int gem_synchronize(gem_handle* h)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    GEM_HIP(h, hipStreamSynchronize(h->stream));
    return GEM_OK;
}

int gem_get_pose(gem_handle* h, float out_center[2], int out_start[2])
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    if (out_center) { out_center[0] = h->center[0]; out_center[1] = h->center[1]; }
    if (out_start) { out_start[0] = h->start[0]; out_start[1] = h->start[1]; }
    return GEM_OK;
}

// Move, gpu_process.cu:1004-1083.  Centre / start live on the host (the handle is their only
// writer), so the reference's two cudaMemcpyFromSymbol round trips per frame disappear.
int gem_move(gem_handle* h, const float position[3], float out_center[2], int out_start[2], float out_aligned_shift[2])
{
    if (!h || !position) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }            // (the clears below touch every rank's strip: behind an all-gather in flight)
    const int L = h->L; const float res = h->res;
    h->sensor_z = position[2];
    int shift[2]; float aligned[2];
    for (int i = 0; i < 2; ++i) {
        const float d = position[i] - h->center[i];
        shift[i] = static_cast<int>(static_cast<double>(d / res) + 0.5 * (d > 0 ? 1 : -1));     // :897
        aligned[i] = static_cast<float>(shift[i]) * res;                                        // :909
    }
    for (int i = 0; i < 2; ++i) {
        if (shift[i] != 0) {
            h->floor_dirty = true;
            if (shift[i] >= L || shift[i] <= -L) {
                // :1034-1038 G_Clear_allmap.  (For shift <= -L the reference indexes past the arrays; we clear all.)
                GEM_HIP(h, launch_init(h->stream, h->layers, h->cells, 0));
            } else {
                const int sign = shift[i] > 0 ? 1 : -1;
                const int start_index = h->start[i] - (sign > 0 ? 1 : 0);
                const int end_index = start_index + sign - shift[i];
                const int n_cells = std::abs(shift[i]);
                int index = index_to_range(sign < 0 ? start_index : end_index, L);
                if (index + n_cells <= L) {
                    GEM_HIP(h, launch_clear_strip(h->stream, h->layers, L, index, n_cells, i == 0));
                } else {
                    const int first_n = L - index;
                    GEM_HIP(h, launch_clear_strip(h->stream, h->layers, L, index, first_n, i == 0));
                    GEM_HIP(h, launch_clear_strip(h->stream, h->layers, L, 0, n_cells - first_n, i == 0));
                }
            }
        }
        h->start[i] = index_to_range(h->start[i] - shift[i], L);
        // PositionToRange, :996-1002
        const int p_index = static_cast<int>(roundf(h->center[i] / res));
        const int s_index = static_cast<int>(roundf(aligned[i] / res));
        h->center[i] = static_cast<float>(p_index + s_index) * res;
    }
    if (out_center) { out_center[0] = h->center[0]; out_center[1] = h->center[1]; }
    if (out_start) { out_start[0] = h->start[0]; out_start[1] = h->start[1]; }
    if (out_aligned_shift) { out_aligned_shift[0] = aligned[0]; out_aligned_shift[1] = aligned[1]; }
    return GEM_OK;
}

int gem_process_points(gem_handle* h, const gem_frame_params* p, int n, float* x, float* y, float* z,
                       const int* orig_index, int write_back_xyz,
                       int* map_index, float* var, float* x_ts, float* y_ts, float* z_ts)
{
    if (!h || !p || n < 0 || (n > 0 && (!x || !y || !z))) return h ? fail(h, GEM_ERR_INVALID, "gem_process_points: bad argument") : GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    if (n == 0) return GEM_OK;
    const size_t N = (size_t)n, S = N * 4;
    int rc;
    FrameConst fc; fill_frame(h, p, fc);
    // Nothing of this call stays on the device: with the pinned staging buffer the kernel reads the cloud from it and writes its
    // five arrays into it, over the link, both directions at once (no device copy of either), in ranges of kRange points -- the
    // copy threads fill range c + 1 and empty range c - 1 while the device works on range c.
    constexpr int kRange = 65536;
    const size_t SP = (S + 255) & ~(size_t)255;
    unsigned char* stg = S >= (64u << 10) ? host_stage(h, SP * 9) : nullptr;
    if (stg && (n + kRange - 1) / kRange <= gem_handle::kStageEvents) {
        { const int rcd = drain_staging(h); if (rcd) return rcd; }
        float* sx = reinterpret_cast<float*>(stg);            float* sy = reinterpret_cast<float*>(stg + SP);
        float* sz = reinterpret_cast<float*>(stg + 2 * SP);   int* sorig = reinterpret_cast<int*>(stg + 3 * SP);
        int* sidx = reinterpret_cast<int*>(stg + 4 * SP);     float* svar = reinterpret_cast<float*>(stg + 5 * SP);
        float* sxt = reinterpret_cast<float*>(stg + 6 * SP);  float* syt = reinterpret_cast<float*>(stg + 7 * SP);
        float* szt = reinterpret_cast<float*>(stg + 8 * SP);
        const int ranges = (n + kRange - 1) / kRange;
        auto out_of = [&](int c) {
            const int first = c * kRange, cnt = std::min(kRange, n - first);
            const size_t b = (size_t)cnt * 4;
            gem::CopySeg segs[8]; int ns = 0;
            if (map_index) segs[ns++] = {map_index + first, sidx + first, b};
            if (var)  segs[ns++] = {var + first, svar + first, b};
            if (x_ts) segs[ns++] = {x_ts + first, sxt + first, b};
            if (y_ts) segs[ns++] = {y_ts + first, syt + first, b};
            if (z_ts) segs[ns++] = {z_ts + first, szt + first, b};
            if (write_back_xyz) { segs[ns++] = {x + first, sx + first, b}; segs[ns++] = {y + first, sy + first, b}; segs[ns++] = {z + first, sz + first, b}; }
            gem::CopyPool::get().run(segs, ns, h->copy_threads);
        };
        for (int c = 0; c < ranges; ++c) {
            const int first = c * kRange, cnt = std::min(kRange, n - first);
            const size_t b = (size_t)cnt * 4;
            gem::CopySeg in[4] = {{sx + first, x + first, b}, {sy + first, y + first, b}, {sz + first, z + first, b}, {sorig + first, orig_index ? orig_index + first : nullptr, orig_index ? b : 0}};
            long long t0 = host_ns();
            gem::CopyPool::get().run(in, orig_index ? 4 : 3, h->copy_threads);
            long long t1 = host_ns();
            GEM_HIP(h, launch_project(h->stream, fc, first, cnt, sx + first, sy + first, sz + first, orig_index ? sorig + first : nullptr, write_back_xyz,
                                      sidx + first, svar + first, sxt + first, syt + first, szt + first));
            GEM_HIP(h, hipEventRecord(h->ev_stage[c], h->stream));
            long long t2 = host_ns();
            h->xfer_ns[0] += t1 - t0; h->xfer_ns[1] += t2 - t1;
            if (c > 0) {
                GEM_HIP(h, hipEventSynchronize(h->ev_stage[c - 1]));
                t0 = host_ns();
                out_of(c - 1);
                h->xfer_ns[3] += t0 - t2; h->xfer_ns[4] += host_ns() - t0;
            }
        }
        long long t0 = host_ns();
        GEM_HIP(h, hipEventSynchronize(h->ev_stage[ranges - 1]));
        long long t1 = host_ns();
        out_of(ranges - 1);
        h->xfer_ns[3] += t1 - t0; h->xfer_ns[4] += host_ns() - t1;
        return GEM_OK;
    }
    if ((rc = ensure(h, h->stage, S * 9))) return rc;
    unsigned char* d = static_cast<unsigned char*>(h->stage.p);
    float* dx = reinterpret_cast<float*>(d);           float* dy = reinterpret_cast<float*>(d + S);
    float* dz = reinterpret_cast<float*>(d + 2 * S);   int* dorig = reinterpret_cast<int*>(d + 3 * S);
    int* didx = reinterpret_cast<int*>(d + 4 * S);     float* dvar = reinterpret_cast<float*>(d + 5 * S);
    float* dxt = reinterpret_cast<float*>(d + 6 * S);  float* dyt = reinterpret_cast<float*>(d + 7 * S);
    float* dzt = reinterpret_cast<float*>(d + 8 * S);
    HostXfer up[4] = {{x, dx, S}, {y, dy, S}, {z, dz, S}, {const_cast<int*>(orig_index), dorig, S}};
    HostXfer down[8]; int nd = 0;
    if (map_index) down[nd++] = {map_index, didx, S};
    if (var)  down[nd++] = {var, dvar, S};
    if (x_ts) down[nd++] = {x_ts, dxt, S};
    if (y_ts) down[nd++] = {y_ts, dyt, S};
    if (z_ts) down[nd++] = {z_ts, dzt, S};
    if (write_back_xyz) { down[nd++] = {x, dx, S}; down[nd++] = {y, dy, S}; down[nd++] = {z, dz, S}; }
    if ((rc = upload_arrays(h, up, orig_index ? 4 : 3))) return rc;
    GEM_HIP(h, launch_project(h->stream, fc, 0, n, dx, dy, dz, orig_index ? dorig : nullptr, write_back_xyz, didx, dvar, dxt, dyt, dzt));
    if (nd) return download_arrays(h, down, nd, SP * 4);
    GEM_HIP(h, hipStreamSynchronize(h->stream));
    return GEM_OK;
}

int gem_fuse(gem_handle* h, int n, const int* index, const int* R, const int* G, const int* B,
             const float* intensity, const float* height, const float* var)
{
    if (!h || n < 0 || (n > 0 && (!index || !height || !var))) return h ? fail(h, GEM_ERR_INVALID, "gem_fuse: bad argument") : GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcs = shard_finish_locked(h); if (rcs) return rcs; }
    const bool attr = R && G && B && intensity;
    const size_t S = (size_t)n * 4;
    PassInput in; in.src = 1; in.n = n;
    if (n > 0) {
        int rc;
        const size_t P = (S + 255) & ~(size_t)255;                  // the arrays' stride on the device = in the staging buffer (upload_arrays)
        if ((rc = ensure(h, h->stage, P * 7))) return rc;
        unsigned char* d = static_cast<unsigned char*>(h->stage.p);
        HostXfer up[7] = {{const_cast<int*>(index), d, S}, {const_cast<float*>(height), d + P, S}, {const_cast<float*>(var), d + 2 * P, S},
                          {const_cast<int*>(R), d + 3 * P, S}, {const_cast<int*>(G), d + 4 * P, S}, {const_cast<int*>(B), d + 5 * P, S},
                          {const_cast<float*>(intensity), d + 6 * P, S}};
        // the caller's arrays are only valid for the call (they are stack VLAs in the reference, EMg.cpp:260-267): read before it returns
        // -- into a half of the staging buffer, where the pass's kernels read them over the link (zero copy, see upload_arrays), or,
        // when the staging buffer does not take them, into the arena by the runtime's copies
        unsigned char* region = nullptr; int half = -1;
        if ((rc = upload_arrays(h, up, attr ? 7 : 3, true, &region, &half))) return rc;
        if (region) d = region;
        in.f_index = reinterpret_cast<const int*>(d); in.f_height = reinterpret_cast<const float*>(d + P);
        in.f_var = reinterpret_cast<const float*>(d + 2 * P);
        if (attr) {
            in.f_R = reinterpret_cast<const int*>(d + 3 * P); in.f_G = reinterpret_cast<const int*>(d + 4 * P);
            in.f_B = reinterpret_cast<const int*>(d + 5 * P); in.f_I = reinterpret_cast<const float*>(d + 6 * P);
        }
        if (region) {
            rc = run_pipeline(h, in);
            const hipError_t e = hipEventRecord(h->ev_half[half], h->stream);       // (the half is free when the pass's kernels have read it)
            if (e != hipSuccess && rc == GEM_OK) rc = fail(h, GEM_ERR_HIP, "hipEventRecord(staging half)", e);
            h->half_pending[half] = true;
            return rc;
        }
    }
    return run_pipeline(h, in);
}

int gem_add_device(gem_handle* h, const gem_frame_params* p, int n, const void* d_xyzi, const void* d_rgb, const void* d_orig_index)
{
    if (!h || !p || n < 0 || (n > 0 && !d_xyzi)) return h ? fail(h, GEM_ERR_INVALID, "gem_add_device: bad argument") : GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcs = shard_finish_locked(h); if (rcs) return rcs; }
    PassInput in; in.src = 0; in.n = n; in.params = p; in.device_input = true; in.caller_device = true;
    in.xyzi = static_cast<const float4*>(d_xyzi); in.rgb = static_cast<const uint32_t*>(d_rgb); in.orig = static_cast<const int*>(d_orig_index);
    return run_pipeline(h, in);
}

int gem_add(gem_handle* h, const gem_frame_params* p, int n, const float* xyzi, const uint32_t* rgb, const int* orig_index)
{
    if (!h || !p || n < 0 || (n > 0 && !xyzi)) return h ? fail(h, GEM_ERR_INVALID, "gem_add: bad argument") : GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcs = shard_finish_locked(h); if (rcs) return rcs; }
    PassInput in; in.src = 0; in.n = n; in.params = p;
    if (n > 0) {
        const size_t S = (size_t)n * 4;
        int rc;
        const size_t P4 = (S * 4 + 255) & ~(size_t)255, P = (S + 255) & ~(size_t)255;      // strides as in the staging buffer (upload_arrays)
        if ((rc = ensure(h, h->stage, P4 + 2 * P))) return rc;
        unsigned char* d = static_cast<unsigned char*>(h->stage.p);
        HostXfer up[3] = {{const_cast<float*>(xyzi), d, S * 4}, {nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
        int nu = 1;
        in.xyzi = reinterpret_cast<const float4*>(d);
        unsigned char* next = d + P4;
        if (rgb) { up[nu++] = {const_cast<uint32_t*>(rgb), next, S}; in.rgb = reinterpret_cast<const uint32_t*>(next); next += P; }
        if (orig_index) { up[nu++] = {const_cast<int*>(orig_index), next, S}; in.orig = reinterpret_cast<const int*>(next); }
        // (the cloud is in the handle's own memory when the kernels run: like a device-resident one, it may take the one-launch-per-frame
        //  path whose deferred fuse reads the binned records only)
        unsigned char* region = nullptr; int half = -1;
        if ((rc = upload_arrays(h, up, nu, true, &region, &half))) return rc;
        in.device_input = true;
        if (region) {
            // the kernels read the staging half itself (upload_arrays): same strides as the arena's
            in.xyzi = reinterpret_cast<const float4*>(region);
            unsigned char* nxt = region + P4;
            if (rgb) { in.rgb = reinterpret_cast<const uint32_t*>(nxt); nxt += P; }
            if (orig_index) in.orig = reinterpret_cast<const int*>(nxt);
            rc = run_pipeline(h, in);
            // the half is free again when everything enqueued so far has run (the pass's kernels read it; a deferred fuse does not)
            // (a pass that put its binning on another stream: that stream's work is ordered before the walk / fuse on h->stream)
            const hipError_t e = hipEventRecord(h->ev_half[half], h->stream);
            if (e != hipSuccess && rc == GEM_OK) rc = fail(h, GEM_ERR_HIP, "hipEventRecord(staging half)", e);
            h->half_pending[half] = true;
            return rc;
        }
    }
    return run_pipeline(h, in);
}

// BASELINE config 4 from HOST memory (gem_add_batch): sweep s = clouds[s][0 .. counts[s]) XYZI points, frames and
// increments as gem_add_batch_device.  The sweeps are copied into the handle's arena one behind the other -- staging copy of sweep
// s + 1 beside the DMA of sweep s -- and fused by ONE batched pass; the caller's arrays have been read when the call returns.
int gem_add_batch(gem_handle* h, int n_sweeps, const gem_frame_params* params, const float* const* clouds, const int* counts, const float* var_updates)
{
    if (!h || n_sweeps <= 0 || !params || !clouds || !counts) return h ? fail(h, GEM_ERR_INVALID, "gem_add_batch: bad argument") : GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcs = shard_finish_locked(h); if (rcs) return rcs; }
    std::vector<long long> offsets(n_sweeps + 1, 0);
    for (int s = 0; s < n_sweeps; ++s) {
        if (counts[s] < 0 || (counts[s] > 0 && !clouds[s])) return fail(h, GEM_ERR_INVALID, "gem_add_batch: bad sweep");
        offsets[s + 1] = offsets[s] + counts[s];
    }
    const long long N = offsets[n_sweeps];
    if (N >= (1ll << 31)) return fail(h, GEM_ERR_INVALID, "gem_add_batch: batch too large");
    int rc;
    if ((rc = ensure(h, h->stage, (size_t)N * 16 + 256))) return rc;
    unsigned char* d = static_cast<unsigned char*>(h->stage.p);
    if (N > 0) {
        std::vector<HostXfer> up;
        for (int s = 0; s < n_sweeps; ++s)
            if (counts[s] > 0) up.push_back({const_cast<float*>(clouds[s]), d + (size_t)offsets[s] * 16, (size_t)counts[s] * 16});
        // (in pieces of at most 64 MB of staging: the pinned buffer stays modest, and a piece's DMA runs beside the next piece's copy)
        size_t i0 = 0;
        while (i0 < up.size()) {
            size_t i1 = i0, bytes = 0;
            while (i1 < up.size() && (i1 == i0 || bytes + up[i1].bytes <= (64u << 20))) bytes += (up[i1++].bytes + 255) & ~(size_t)255;
            if ((rc = upload_arrays(h, up.data() + i0, (int)(i1 - i0), true))) return rc;
            i0 = i1;
        }
    }
    if (n_sweeps == 1) {
        if (var_updates) {
            if (!(var_updates[0] >= 0.f)) h->floor_dirty = true;
            if (h->n_pending == kMaxPending) { rc = flush_pending(h, false); if (rc) return rc; }
            h->pending[h->n_pending++] = var_updates[0];
        }
        PassInput in; in.src = 0; in.n = N; in.params = params; in.device_input = true;
        in.xyzi = reinterpret_cast<const float4*>(d);
        return run_pipeline(h, in);
    }
    PassInput in; in.src = 0; in.n_sweeps = n_sweeps; in.n = N; in.params = params;
    in.offsets = offsets.data(); in.var_updates = var_updates;
    in.xyzi = reinterpret_cast<const float4*>(d);
    return run_pipeline(h, in);
}

int gem_add_aos(gem_handle* h, const gem_frame_params* p, int n, const void* points, int point_step,
                int off_x, int off_y, int off_z, int off_intensity, int off_rgb)
{
    if (!h || !p || n < 0 || (n > 0 && !points)) return h ? fail(h, GEM_ERR_INVALID, "gem_add_aos: bad argument") : GEM_ERR_INVALID;
    auto field_ok = [&](int o, bool optional) { return (optional && o < 0) || (o >= 0 && (o & 3) == 0 && o + 4 <= point_step); };
    if (point_step < 12 || (point_step & 3) || !field_ok(off_x, false) || !field_ok(off_y, false) || !field_ok(off_z, false) ||
        !field_ok(off_intensity, true) || !field_ok(off_rgb, true))
        return fail(h, GEM_ERR_INVALID, "gem_add_aos: fields must be 4-byte aligned inside point_step");
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcs = shard_finish_locked(h); if (rcs) return rcs; }
    PassInput in; in.src = 0; in.n = n; in.params = p;
    if (n > 0) {
        const size_t raw = ((size_t)n * point_step + 15) & ~(size_t)15, S = (size_t)n * 4;
        int rc;
        if ((rc = ensure(h, h->stage, raw + S * 5))) return rc;
        unsigned char* d = static_cast<unsigned char*>(h->stage.p);
        HostXfer up{const_cast<void*>(points), d, (size_t)n * point_step};
        if ((rc = upload_arrays(h, &up, 1))) return rc;              // (returns once the caller's buffer has been read)
        float4* xyzi = reinterpret_cast<float4*>(d + raw);
        uint32_t* rgb = off_rgb >= 0 ? reinterpret_cast<uint32_t*>(d + raw + S * 4) : nullptr;
        GEM_HIP(h, launch_unpack_aos(h->stream, d, n, point_step, off_x, off_y, off_z, off_intensity, off_rgb, xyzi, rgb));
        in.xyzi = xyzi; in.rgb = rgb;
    }
    return run_pipeline(h, in);
}

int gem_add_batch_device(gem_handle* h, int n_sweeps, const gem_frame_params* params, const void* d_xyzi,
                         const long long* offsets, const float* var_updates)
{
    if (!h || n_sweeps <= 0 || !params || !offsets) return h ? fail(h, GEM_ERR_INVALID, "gem_add_batch_device: bad argument") : GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcs = shard_finish_locked(h); if (rcs) return rcs; }
    for (int s = 0; s < n_sweeps; ++s) if (offsets[s + 1] < offsets[s]) return fail(h, GEM_ERR_INVALID, "gem_add_batch_device: offsets not monotone");
    if (n_sweeps == 1) {
        if (var_updates) {
            if (!(var_updates[0] >= 0.f)) h->floor_dirty = true;
            if (h->n_pending == kMaxPending) { int rc = flush_pending(h, false); if (rc) return rc; }
            h->pending[h->n_pending++] = var_updates[0];
        }
        PassInput in; in.src = 0; in.n = offsets[1] - offsets[0]; in.params = params; in.device_input = true; in.caller_device = true;
        in.xyzi = static_cast<const float4*>(d_xyzi) + offsets[0];
        return run_pipeline(h, in);
    }
    PassInput in; in.src = 0; in.n_sweeps = n_sweeps; in.n = offsets[n_sweeps]; in.params = params; in.caller_device = true;
    in.offsets = offsets; in.var_updates = var_updates;
    in.xyzi = static_cast<const float4*>(d_xyzi);
    return run_pipeline(h, in);
}

// Arenas for the largest pass the caller is going to make, allocated NOW: the arenas only ever grow, but growing means waiting for
// everything in flight, hipFree and hipMalloc -- in the middle of a stream of frames that is a stall of a millisecond or more the
// first time a bigger cloud arrives (measured: tools/bench_configs.py --configs reserve).  max_points points in at most max_sweeps
// sweeps per call (1 for gem_add*); colours as they will be passed.  Sizes follow run_pipeline / run_sort_pipeline, for EVERY
// pipeline a pass within the bounds can take: the sorted forms (cell-sorted for single clouds, block-sorted for batches) from their
// thresholds on, the tile pipeline below them.  On a handle that joined a communicator with tile strips, max_points / max_sweeps
// bound the GLOBAL points / sweeps of a gem_add_sharded_device step: the shard's sort (its W-th of the points), both sets of
// receive buffers (no strip gets more records than the step has points) and the published copies are sized as well.
int gem_reserve(gem_handle* h, long long max_points, int max_sweeps, int with_colours)
{
    if (!h || max_points < 0 || max_sweeps < 1 || max_points >= (1ll << 31)) return h ? fail(h, GEM_ERR_INVALID, "gem_reserve: bad argument") : GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    if (max_points == 0) return GEM_OK;
    int rc;
    const long long blocks = 4ll * ((h->L + 31) / 32) * ((h->L + 31) / 32);
    // staging of host-pointer inputs (gem_add: XYZI + rgb + orig; gem_fuse: seven arrays; gem_process_points: nine)
    if ((rc = ensure(h, h->stage, ((size_t)max_points * 4 + 256) * 9))) return rc;
    {   // ... and its pinned counterpart for callers with host arrays (gem_process_points: nine arrays; gem_map_feature: nine layers),
        // where that is a modest amount: larger ones grow on first use
        constexpr size_t kReserveMax = 64u << 20;
        const size_t a = ((size_t)max_points * 4 + 256) * 9, b = ((size_t)h->cells * 4 + 256) * 9;
        // the deferred / zero-copy uploads (upload_arrays) keep TWO calls' arrays in the buffer, a half each: gem_fuse's seven arrays
        // are the largest (28 B per point; gem_add with rgb + orig_index: 24), as long as one call stays below the 16 MB from which
        // uploads go to the runtime's pageable path
        const size_t one = ((size_t)max_points * 4 + 256) * 7;
        const size_t c = one < (16u << 20) ? 2 * one + 512 : 0;
        const size_t want = std::max(std::max(a <= kReserveMax ? a : 0, b <= kReserveMax ? b : 0), c);
        if (want) (void)host_stage(h, want);
    }
    auto reserve_tables = [&](int sweeps) -> int {                       // the batched calls' tables and their pinned staging copies
        const size_t tables = sizeof(FrameConst) * sweeps + (sizeof(int) + sizeof(long long)) * (sweeps + 1) + (sizeof(float) + sizeof(int)) * sweeps + 64;
        for (auto& pb : h->pb) {
            int r;
            if (tables > pb.tables.cap) pb.tab_key.clear();          // (the cached tables of this buffer set go with the old allocation)
            if ((r = ensure(h, pb.tables, tables))) return r;
            if (tables > pb.host_cap) {
                if (pb.tables_recorded) GEM_HIP(h, hipEventSynchronize(pb.tables_done));
                if (pb.host_tables) GEM_HIP(h, hipHostFree(pb.host_tables));
                pb.host_tables = nullptr; pb.host_cap = 0;
                GEM_HIP(h, hipHostMalloc(&pb.host_tables, tables * 2, hipHostMallocDefault));
                pb.host_cap = tables * 2;
            }
        }
        return GEM_OK;
    };
    auto reserve_sorted = [&](long long points, int sweeps, bool block_form, bool shard) -> int {
        SortGeometry geo = sort_geometry(h, sweeps, block_form);
        if (!geo.ok) geo = sort_geometry(h, sweeps, !block_form);
        if (!geo.ok) return GEM_OK;                                      // (such a pass takes the tile pipeline)
        // chunks: a pass of fewer points may take the small chunk (sort_chunk_for) and then has MORE chunks than the largest pass
        const size_t N = (size_t)points;
        size_t nc_max = 0;
        for (const long long pts : {points, std::min<long long>(points, 2ll * 256 * kSortChunkRecords - 1)}) {
            const size_t c = (size_t)sort_chunk_for(pts, h->sort_chunk);
            nc_max = std::max(nc_max, ((size_t)pts + c - 1) / c);
        }
        const size_t NC1 = nc_max + (size_t)sweeps, nc2 = nc_max;
        int bins_hi = 1;
        for (int i = 1; i < geo.n_passes; ++i) bins_hi = std::max(bins_hi, geo.dbins[i]);
        size_t misc = 0;
        for (int i = 0; i < geo.n_passes; ++i) misc += (size_t)geo.dbins[i] * 16;
        misc = ((misc + 4 + 15) & ~(size_t)15) + (((size_t)geo.dbins[geo.n_passes - 1] + 1) * 4 + 15) + NC1 * kSortSegsPerChunk * 4 + 64;
        const int slots = std::max(1, std::min(h->sort_ring, 4));
        for (int k = 0; k < slots; ++k) {
            gem_handle::PassBuffers& pb = h->pb[k];
            int r;
            if ((r = ensure(h, pb.s_hv1, N * 8 + 64)) || (r = ensure(h, pb.s_hv2, N * 8 + 64)) ||
                (r = ensure(h, pb.s_key1, N * 4 + 64)) || (r = ensure(h, pb.s_key2, N * 4 + 64))) return r;
            if (with_colours && ((r = ensure(h, pb.s_src1, N * 4 + 64)) || (r = ensure(h, pb.s_src2, N * 4 + 64)))) return r;
            if ((r = ensure(h, pb.s_cnt1, NC1 * geo.dbins[0] * 4)) || (r = ensure(h, pb.s_cnt2, nc2 * bins_hi * 4 + 16)) ||
                (r = ensure(h, pb.s_misc, misc))) return r;
            if (((geo.block_form && geo.n_passes > 1) || shard) &&
                ((r = ensure(h, pb.s_ranges, (size_t)blocks * sizeof(uint2))) || (r = ensure_zeroed(h, pb.s_blkcnt, (size_t)blocks * sizeof(uint32_t))))) return r;
            if (shard && (r = ensure(h, pb.s_shard, 64 * sizeof(uint32_t)))) return r;
        }
        return GEM_OK;
    };
    auto reserve_tiles = [&](long long points, int sweeps) -> int {
        // units of 64 points, every sweep rounded up to 32 units; the descriptor table is [sweep][tile][units of the longest sweep].
        // A single cloud beyond kSweepPoints is cut into sweeps of that size (run_pipeline); the sweeps of a batch are taken to be at
        // most twice their mean length (or kSweepPoints) -- the table for "all points in one of 32 sweeps" would be 32 times the useful one.
        auto units_of = [](long long pts) { return ((pts + kUnit - 1) / kUnit + 31) & ~31ll; };
        long long units1, B;
        if (sweeps == 1 && points > kSweepPoints) { sweeps = (int)((points + kSweepPoints - 1) / kSweepPoints); units1 = units_of(kSweepPoints); B = units1 * sweeps; }
        else if (sweeps == 1) { units1 = units_of(points); B = units1; }
        else { units1 = units_of(std::min(points, std::max(kSweepPoints, 2 * points / sweeps))); B = units_of(points) + 32ll * (sweeps - 1); }
        const int ts = h->ts ? h->ts : 4;
        const long long tpr = (h->L + (1 << ts) - 1) >> ts, T = tpr * tpr;
        const size_t seg = (size_t)sweeps * T * units1 * sizeof(uint16_t);
        if (seg > ((size_t)1 << 31)) return GEM_OK;           // (not a shape the tile pipeline is meant for: such a pass sizes its own table, or fails there)
        for (int k = 0; k < 2; ++k) {
            gem_handle::PassBuffers& pb = h->pb[k];
            int r;
            if ((r = ensure(h, pb.rec, (size_t)B * kUnit * sizeof(uint4))) || (r = ensure(h, pb.srt, (size_t)B * kUnit * sizeof(uint4) + 16))) return r;
            if (seg > pb.seg.cap) {                          // (the table is all-zero between passes: cleared when it is (re)allocated)
                if ((r = ensure(h, pb.seg, seg))) return r;
                GEM_HIP(h, hipMemsetAsync(pb.seg.p, 0, pb.seg.cap, h->stream));
            }
            const size_t flag = (size_t)T * sweeps * sizeof(uint32_t), gflag = (size_t)sweeps * T * (units1 / 32) * sizeof(uint32_t);
            if (flag > pb.flag.cap || gflag > pb.gflag.cap) {
                if ((r = ensure(h, pb.flag, flag)) || (r = ensure(h, pb.gflag, gflag))) return r;
                GEM_HIP(h, hipMemsetAsync(pb.flag.p, 0, pb.flag.cap, h->stream));
                GEM_HIP(h, hipMemsetAsync(pb.gflag.p, 0, pb.gflag.cap, h->stream));
                pb.epoch = 0;
            }
        }
        return GEM_OK;
    };
    if (h->tp_x && h->tile_strips) {
        // a step of the sharded path: this rank sorts its W-th of the points, block-sorted; every strip's owner receives at most all of them
        const int W = h->nranks;
        const long long share = (max_points + W - 1) / W + 1;
        if ((rc = reserve_sorted(share, max_sweeps, true, true)) || (rc = reserve_tables(max_sweeps))) return rc;
        if (!h->sh_host) GEM_HIP(h, hipHostMalloc(&h->sh_host, kShardHostBytes, hipHostMallocDefault));
        if ((rc = ensure(h, h->sh_dev, kShardDevBytes)) || (rc = ensure(h, h->sh_ranges, (size_t)blocks * sizeof(uint2)))) return rc;
        h->recv_bound = max_points;
        if (W > 1 && ((rc = ensure_recv(h, 0, (size_t)max_points + 4 * W)) || (rc = ensure_recv(h, 1, (size_t)max_points + 4 * W)))) return rc;
        GEM_HIP(h, hipStreamSynchronize(h->stream));
        return GEM_OK;                                                   // (a handle of the sharded path: its steps are what the bounds describe)
    }
    const bool sorted_single = h->sort_path && max_points >= h->sort_min_points;
    const bool sorted_batch = h->sort_path && max_sweeps > 1 && max_points >= h->sort_min_points_batch;
    if (sorted_single && (rc = reserve_sorted(max_points, 1, h->sort_form == 2, false))) return rc;
    if (sorted_batch && ((rc = reserve_sorted(max_points, max_sweeps, h->sort_form != 1, false)))) return rc;
    if (max_sweeps > 1 && (rc = reserve_tables(max_sweeps))) return rc;
    // the tile pipeline takes what stays below the thresholds (and everything when the sorted forms are off)
    if ((rc = reserve_tiles(sorted_single ? std::min(max_points, h->sort_min_points - 1) : max_points, 1))) return rc;
    if (max_sweeps > 1 && (rc = reserve_tiles(sorted_batch ? std::min(max_points, h->sort_min_points_batch - 1) : max_points, max_sweeps))) return rc;
    if (h->track_lowest && !h->ray.p) {                                  // gem_raytracing's list of walking cells, counters and snapshot
        if ((rc = ensure(h, h->ray, ((size_t)h->cells * 2 + 4) * sizeof(uint32_t)))) return rc;
        GEM_HIP(h, hipMemsetAsync(static_cast<uint32_t*>(h->ray.p) + h->cells, 0, 4 * sizeof(uint32_t), h->stream));
    }
    GEM_HIP(h, hipStreamSynchronize(h->stream));
    return GEM_OK;
}

int gem_mapvar_update(gem_handle* h, float var_update)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcs = shard_finish_locked(h); if (rcs) return rcs; }       // (the increment belongs behind a pending step's walk)
    // queued and folded into the next fuse's single pass over the tiles; a negative (or NaN)
    // increment can push a variance under the floor, which the next Fuse must repair everywhere
    if (!(var_update >= 0.f)) h->floor_dirty = true;
    if (h->n_pending == kMaxPending) { int rc = flush_pending(h, false); if (rc) return rc; }
    h->pending[h->n_pending++] = var_update;
    return GEM_OK;
}

static void* layer_ptr(gem_handle* h, int layer)
{
    switch (layer) {
    case GEM_LAYER_ELEVATION: return h->layers.elevation;
    case GEM_LAYER_VARIANCE:  return h->layers.variance;
    case GEM_LAYER_INTENSITY: return h->layers.intensity;
    case GEM_LAYER_TRAVER:    return h->layers.traver;
    case GEM_LAYER_LOWEST:    return h->layers.lowest;
    case GEM_LAYER_COLOR_R:   return h->layers.colorR;
    case GEM_LAYER_COLOR_G:   return h->layers.colorG;
    case GEM_LAYER_COLOR_B:   return h->layers.colorB;
    case GEM_LAYER_ROUGH:     return h->layers.rough;
    case GEM_LAYER_SLOPE:     return h->layers.slope;
    default: return nullptr;
    }
}

int gem_get_layer(gem_handle* h, int layer, int layout, void* dst_host)
{
    if (!h || !dst_host) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    void* src = layer_ptr(h, layer);
    if (!src) return fail(h, GEM_ERR_INVALID, "gem_get_layer: bad layer");
    int rc = flush_pending(h, false);
    if (rc) return rc;
    const size_t bytes = (size_t)h->cells * 4;
    if (layout == GEM_LAYOUT_STORAGE_ROWMAJOR) {
        HostXfer down{dst_host, src, bytes};
        return download_arrays(h, &down, 1, 0);
    } else if (layout == GEM_LAYOUT_GRIDMAP_COLMAJOR_NAN) {
        if (layer == GEM_LAYER_LOWEST) return fail(h, GEM_ERR_INVALID, "gem_get_layer: the LOWEST layer is indexed by geographic cell, it has no grid_map layout");
        if ((rc = ensure(h, h->scratch, bytes))) return rc;
        const int is_int = layer >= GEM_LAYER_COLOR_R && layer <= GEM_LAYER_COLOR_B;
        GEM_HIP(h, launch_export_gridmap(h->stream, src, h->layers.elevation, static_cast<float*>(h->scratch.p), h->L, is_int));
        HostXfer down{dst_host, h->scratch.p, bytes};
        return download_arrays(h, &down, 1, 0);
    }
    return fail(h, GEM_ERR_INVALID, "gem_get_layer: bad layout");
}

int gem_set_layer(gem_handle* h, int layer, const void* src_host)
{
    if (!h || !src_host) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    void* dst = layer_ptr(h, layer);
    if (!dst) return fail(h, GEM_ERR_INVALID, "gem_set_layer: bad layer");
    int rc = flush_pending(h, false);
    if (rc) return rc;
    HostXfer up{const_cast<void*>(src_host), dst, (size_t)h->cells * 4};
    if ((rc = upload_arrays(h, &up, 1))) return rc;
    GEM_HIP(h, hipStreamSynchronize(h->stream));
    if (layer == GEM_LAYER_VARIANCE) h->floor_dirty = true;
    return GEM_OK;
}

int gem_layer_device_ptr(gem_handle* h, int layer, void** out_device_ptr)
{
    if (!h || !out_device_ptr) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    // the caller is about to read or write the layer: the pending fuse AND the queued Mapvar_update increments go first
    { const int rcd = flush_pending(h, false); if (rcd) return rcd; }
    *out_device_ptr = layer_ptr(h, layer);
    return *out_device_ptr ? GEM_OK : fail(h, GEM_ERR_INVALID, "gem_layer_device_ptr: bad layer");
}

// Map_optmove (gpu_process.cu:1215-1233, alignedPosition :1203-1213): after a loop closure the map centre is
// relabelled to the optimised position snapped to the old centre's cell lattice -- the circular buffer is not
// shifted, nothing is cleared -- and every valid elevation moves by height_update (G_update_mapheight :1195-1202).
int gem_map_optmove(gem_handle* h, const float opt_position[2], float height_update, float out_aligned_position[2])
{
    if (!h || !opt_position) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    for (int i = 0; i < 2; ++i) {
        const float d = opt_position[i] - h->center[i];
        const int shift = static_cast<int>(static_cast<double>(d / h->res) + 0.5 * (d > 0 ? 1 : -1));     // :1210
        h->center[i] = h->center[i] + h->res * static_cast<float>(shift);                                  // :1211
    }
    if (out_aligned_position) { out_aligned_position[0] = h->center[0]; out_aligned_position[1] = h->center[1]; }
    GEM_HIP(h, launch_update_height(h->stream, h->layers.elevation, h->cells, height_update));
    return GEM_OK;
}

// Map_closeloop (gpu_process.cu:1235-1254; declared by the node, never called): the centre moves by the aligned
// shift through PositionToRange like Move's, the buffer stays, plus the height shift.
int gem_map_closeloop(gem_handle* h, const float update_position[2], float height_update)
{
    if (!h || !update_position) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    for (int i = 0; i < 2; ++i) {
        const float d = update_position[i] - h->center[i];
        const int shift = static_cast<int>(static_cast<double>(d / h->res) + 0.5 * (d > 0 ? 1 : -1));     // :897
        const float aligned = static_cast<float>(shift) * h->res;                                          // :909
        const int p_index = static_cast<int>(roundf(h->center[i] / h->res));                               // :996-1002
        const int s_index = static_cast<int>(roundf(aligned / h->res));
        h->center[i] = static_cast<float>(p_index + s_index) * h->res;
    }
    GEM_HIP(h, launch_update_height(h->stream, h->layers.elevation, h->cells, height_update));
    return GEM_OK;
}

// Map_feature (gpu_process.cu:1256-1302): the reference mallocs nine device arrays, runs G_Mapfeature and
// copies all nine back every frame; here the kernel writes three resident layers and only the arrays
// the caller asks for are copied.
int gem_map_feature(gem_handle* h, float* elevation, float* variance, int* colorR, int* colorG, int* colorB,
                    float* rough, float* slope, float* traver, float* intensity)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    int rc = flush_pending(h, false);
    if (rc) return rc;
    GEM_HIP(h, launch_map_feature(h->stream, h->layers.elevation, h->layers.traver, h->layers.rough, h->layers.slope,
                                  h->L, h->res, h->start[0], h->start[1], h->row0, h->row1));
    const size_t bytes = (size_t)h->cells * 4;
    struct { void* dst; const void* src; } out[9] = {
        {elevation, h->layers.elevation}, {variance, h->layers.variance}, {colorR, h->layers.colorR}, {colorG, h->layers.colorG},
        {colorB, h->layers.colorB}, {rough, h->layers.rough}, {slope, h->layers.slope}, {traver, h->layers.traver},
        {intensity, h->layers.intensity}};
    HostXfer down[9]; int nd = 0;
    for (auto& o : out) if (o.dst) down[nd++] = {o.dst, const_cast<void*>(o.src), bytes};
    return nd ? download_arrays(h, down, nd, 0) : GEM_OK;
}

// ElevationMap::show's cell loop (ElevationMap.cpp:85-149) on the resident layers: visualMap_'s nine layers in grid_map's own
// layout, the coloured point cloud (compacted on the device, in the reference's iteration order) and the orthomosaic.
int gem_show(gem_handle* h, double map_length, double resolution, const double position[2],
             float* visual, float* points_xyz, unsigned char* points_rgb, int* out_count, unsigned char* image_bgr)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    int rc = flush_pending(h, false);
    if (rc) return rc;
    const size_t cells = (size_t)h->cells, blocks = (cells + 1023) / 1024;
    // scratch: block counts | total | visual 9 L^2 floats | xyz 3 L^2 floats | rgb 3 L^2 bytes | image 3 L^2 bytes
    const size_t o_cnt = 0, o_total = o_cnt + blocks * 4, o_vis = (o_total + 4 + 255) & ~(size_t)255, o_xyz = o_vis + cells * 36,
                 o_rgb = o_xyz + cells * 12, o_img = (o_rgb + cells * 3 + 255) & ~(size_t)255, total_bytes = o_img + cells * 3;
    if ((rc = ensure(h, h->scratch, total_bytes))) return rc;
    unsigned char* d = static_cast<unsigned char*>(h->scratch.p);
    const double res = resolution > 0.0 ? resolution : (double)h->res;
    const double len = map_length > 0.0 ? map_length : (double)h->L * res;
    const double px = position ? position[0] : (double)h->center[0], py = position ? position[1] : (double)h->center[1];
    if (image_bgr) GEM_HIP(h, hipMemsetAsync(d + o_img, 0, cells * 3, h->stream));          // cv::Mat(..., Scalar(0, 0, 0)), EM.cpp:87
    GEM_HIP(h, launch_show(h->stream, h->layers, h->L, h->start[0], h->start[1], len, res, px, py, reinterpret_cast<uint32_t*>(d + o_cnt),
                           visual ? reinterpret_cast<float*>(d + o_vis) : nullptr, (points_xyz || points_rgb) ? reinterpret_cast<float*>(d + o_xyz) : nullptr,
                           points_rgb ? d + o_rgb : nullptr, image_bgr ? d + o_img : nullptr, reinterpret_cast<uint32_t*>(d + o_total)));
    uint32_t n = 0;
    {   // (through the pinned staging buffer like the node's other host arrays: download_arrays)
        HostXfer down[3] = {{&n, d + o_total, 4}, {nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
        int nd = 1;
        if (visual) down[nd++] = {visual, d + o_vis, cells * 36};
        if (image_bgr) down[nd++] = {image_bgr, d + o_img, cells * 3};
        if ((rc = download_arrays(h, down, nd, 0))) return rc;
    }
    if (n) {                                            // only the kept cells' points travel
        HostXfer down[2]; int nd = 0;
        if (points_xyz) down[nd++] = {points_xyz, d + o_xyz, (size_t)n * 12};
        if (points_rgb) down[nd++] = {points_rgb, d + o_rgb, (size_t)n * 3};
        if (nd && (rc = download_arrays(h, down, nd, 0))) return rc;
    }
    if (out_count) *out_count = (int)n;
    return GEM_OK;
}

// ---- input colourisation (EMg.cpp:349-381): device-resident cloud and image -> 0x00RRGGBB per point, intensity zeroed outside
static int colorize_device(gem_handle* h, const gem_camera* cam, int n, float* d_xyzi, const unsigned char* d_image, size_t stride, uint32_t* d_rgb)
{
    if (!cam || n < 0 || (n > 0 && (!d_xyzi || !d_image || !d_rgb))) return fail(h, GEM_ERR_INVALID, "gem_colorize: null argument");
    if (cam->width <= 0 || cam->height <= 0 || (long long)cam->width * cam->height > (1ll << 26))
        return fail(h, GEM_ERR_INVALID, "gem_colorize: image size out of range");
    if (stride == 0) stride = (size_t)cam->width * 3;
    if (stride < (size_t)cam->width * 3) return fail(h, GEM_ERR_INVALID, "gem_colorize: row stride below width * 3");
    if (n == 0) return GEM_OK;
    const long long pixels = (long long)cam->width * cam->height;
    // the key is the pixel: digits of about equal width, at most ten bits (see sort_geometry)
    SortArgs sa{};
    sa.id_bits = std::max(2, ceil_log2((int)std::min<long long>(pixels, 1ll << 30)));
    sa.n_passes = sa.id_bits <= 10 ? 1 : (sa.id_bits <= 20 ? 2 : 3);
    int shift = 0, bins_hi = 1;
    for (int i = 0; i < sa.n_passes; ++i) {
        const int left = sa.n_passes - i;
        const int bits = i == sa.n_passes - 1 ? sa.id_bits - shift : (sa.id_bits - shift + left - 1) / left;
        sa.dshift[i] = shift; sa.dbits[i] = bits;
        sa.dbins[i] = i == sa.n_passes - 1 ? (int)((pixels - 1) >> shift) + 1 : 1 << bits;
        if (i > 0) bins_hi = std::max(bins_hi, sa.dbins[i]);
        shift += bits;
    }
    const size_t N = (size_t)n, NC = (N + kSortChunkRecords - 1) / kSortChunkRecords;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
    const size_t o_hv1 = take(N * 8 + 64), o_hv2 = take(N * 8 + 64), o_key1 = take(N * 4 + 64), o_key2 = take(N * 4 + 64);
    const size_t o_cnt1 = take(NC * sa.dbins[0] * 4), o_cnt2 = take(NC * bins_hi * 4 + 16);
    size_t o_seg[3] = {0, 0, 0};
    for (int i = 0; i < sa.n_passes; ++i) o_seg[i] = take((size_t)sa.dbins[i] * 16);
    const size_t o_total = take(16), o_base = take(((size_t)sa.dbins[sa.n_passes - 1] + 1) * 4), o_segcnt = take(NC * kSortSegsPerChunk * 4);
    const size_t o_first = take((size_t)pixels * 4), o_pix = take(N * 4), o_link = take(N * 4);
    int rc;
    if ((rc = ensure(h, h->color, o))) return rc;
    unsigned char* d = static_cast<unsigned char*>(h->color.p);
    sa.n_sweeps = 1; sa.n = n; sa.xyzi = reinterpret_cast<const float4*>(d_xyzi);
    for (int k = 0; k < 12; ++k) sa.cam.P[k] = cam->lidar_to_image[k];
    sa.cam.width = cam->width; sa.cam.height = cam->height;
    sa.tiles_per_row = 1; sa.T = 1; sa.n_chunks1 = (int)NC; sa.chunk = kSortChunkRecords;
    for (int i = 0; i < sa.n_passes; ++i) {
        sa.cnt[i] = reinterpret_cast<uint32_t*>(d + (i == 0 ? o_cnt1 : o_cnt2));
        sa.segtot[i] = reinterpret_cast<uint32_t*>(d + o_seg[i]);
    }
    sa.total = reinterpret_cast<uint32_t*>(d + o_total); sa.bin_base = reinterpret_cast<uint32_t*>(d + o_base);
    sa.seg_cnt = reinterpret_cast<uint32_t*>(d + o_segcnt);
    sa.hv_a = reinterpret_cast<uint2*>(d + o_hv2); sa.hv_b = reinterpret_cast<uint2*>(d + o_hv1);
    sa.key_a = reinterpret_cast<uint32_t*>(d + o_key2); sa.key_b = reinterpret_cast<uint32_t*>(d + o_key1);
    const LaunchEvents ev[9] = {};
    GEM_HIP(h, hipMemsetAsync(d + o_first, 0xff, (size_t)pixels * 4, h->stream));
    GEM_HIP(h, hipMemsetAsync(d + o_pix, 0xff, N * 4, h->stream));
    GEM_HIP(h, launch_sort(h->stream, sa, 3, false, ev));
    const bool final_b = (sa.n_passes & 1) != 0;
    ColorArgs ca{};
    ca.key = final_b ? sa.key_b : sa.key_a; ca.hv = final_b ? sa.hv_b : sa.hv_a; ca.total = sa.total;
    ca.first = reinterpret_cast<uint32_t*>(d + o_first); ca.pix = reinterpret_cast<uint32_t*>(d + o_pix); ca.link = reinterpret_cast<uint32_t*>(d + o_link);
    ca.n = n; ca.width = cam->width; ca.height = cam->height; ca.image = d_image; ca.stride = stride;
    ca.xyzi = reinterpret_cast<float4*>(d_xyzi); ca.rgb = d_rgb;
    GEM_HIP(h, launch_colorize(h->stream, ca));
    return GEM_OK;
}

int gem_colorize_device(gem_handle* h, const gem_camera* cam, int n, float* d_xyzi, const unsigned char* d_image_bgr, size_t row_stride,
                        uint32_t* d_rgb)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    return colorize_device(h, cam, n, d_xyzi, d_image_bgr, row_stride, d_rgb);
}

int gem_colorize(gem_handle* h, const gem_camera* cam, int n, float* xyzi, const unsigned char* image_bgr, size_t row_stride, uint32_t* rgb)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    if (!cam || n < 0 || (n > 0 && (!xyzi || !image_bgr || !rgb))) return fail(h, GEM_ERR_INVALID, "gem_colorize: null argument");
    if (cam->width <= 0 || cam->height <= 0) return fail(h, GEM_ERR_INVALID, "gem_colorize: image size out of range");
    if (n == 0) return GEM_OK;
    if (row_stride == 0) row_stride = (size_t)cam->width * 3;
    { const int rcd = settle(h); if (rcd) return rcd; }       // the staging arena may hold a deferred frame's cloud
    const size_t N = (size_t)n, b_xyzi = (N * 16 + 255) & ~(size_t)255, b_rgb = (N * 4 + 255) & ~(size_t)255, b_img = row_stride * cam->height;
    int rc;
    if ((rc = ensure(h, h->stage, b_xyzi + b_rgb + b_img))) return rc;
    unsigned char* d = static_cast<unsigned char*>(h->stage.p);
    HostXfer up[2] = {{xyzi, d, N * 16}, {const_cast<unsigned char*>(image_bgr), d + b_xyzi + b_rgb, b_img}};
    if ((rc = upload_arrays(h, up, 2))) return rc;
    if ((rc = colorize_device(h, cam, n, reinterpret_cast<float*>(d), d + b_xyzi + b_rgb, row_stride, reinterpret_cast<uint32_t*>(d + b_xyzi)))) return rc;
    HostXfer down[2] = {{xyzi, d, N * 16}, {rgb, d + b_xyzi, N * 4}};
    return download_arrays(h, down, 2, 0);
}

int gem_set_lowest_tracking(gem_handle* h, int enabled)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    h->track_lowest = enabled != 0;
    return GEM_OK;
}

int gem_raytracing(gem_handle* h)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    // the walks read the lowest scan points of the whole map, which a row strip (multi-GPU tiling) only holds for its own cells
    if (h->row0 != 0 || h->row1 != h->L) return fail(h, GEM_ERR_INVALID, "gem_raytracing: not available on a row-strip handle");
    int rc = flush_pending(h, false);               // the queued variance increments are part of what the kernel reads
    if (rc) return rc;
    if (!h->ray.p) {                                    // the list of walking cells, its two counters (zeroed once; every call zeroes the next one's), the snapshot
        if ((rc = ensure(h, h->ray, ((size_t)h->cells * 2 + 4) * sizeof(uint32_t)))) return rc;
        GEM_HIP(h, hipMemsetAsync(static_cast<uint32_t*>(h->ray.p) + h->cells, 0, 4 * sizeof(uint32_t), h->stream));
    }
    uint32_t* list = static_cast<uint32_t*>(h->ray.p);
    GEM_HIP(h, launch_raytracing(h->stream, h->layers, h->L, h->start[0], h->start[1], h->sensor_z, h->cfg.obstacle_threshold,
                                 h->row0, h->row1, list, list + h->cells, (int)(h->ray_calls++ & 1u), reinterpret_cast<float*>(list + h->cells + 4), h->ray_depth, h->ray_lanes));
    return GEM_OK;
}

int gem_set_timing(gem_handle* h, int enabled)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    h->timing = enabled != 0;
    h->step_timed = h->timing && h->tp_x != nullptr && h->nranks > 1;
    return GEM_OK;
}

int gem_set_counting(gem_handle* h, int enabled)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    h->counting = enabled != 0;
    return GEM_OK;
}

int gem_get_stats(gem_handle* h, gem_stats* out, int reset)
{
    if (!h || !out) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    GEM_HIP(h, hipStreamSynchronize(h->stream));
    fold_events(h);
    if (h->counting) {
        unsigned long long c[2] = {0, 0};
        GEM_HIP(h, hipMemcpy(c, h->d_counters, sizeof(c), hipMemcpyDeviceToHost));
        h->stats.points_binned = (long long)c[0];
        h->stats.cells_touched = (long long)c[1];
    }
    *out = h->stats;
    if (reset) { const long long pin = h->stats.points_in; h->stats = gem_stats{}; h->stats.points_in = pin; }
    return GEM_OK;
}

// Tuning / test knobs (include/gem_hip_debug.h; not part of the drop-in surface).  They select between code paths that all
// produce the same map: the tests use them to drive every path with small inputs.
int gem_debug_set(gem_handle* h, const char* key, long long value)
{
    if (!h || !key) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    const std::string k(key);
    if (k == "fuse_variant")            { if (value < 10 || value > 12) return fail(h, GEM_ERR_INVALID, "fuse_variant: 10..12"); h->fuse_variant = (int)value; }
    else if (k == "tile_shift")         { if (value != 0 && value != 4 && value != 5) return fail(h, GEM_ERR_INVALID, "tile_shift: 0, 4 or 5"); h->ts = (int)value; }
    else if (k == "defer")              h->defer = value != 0;
    else if (k == "walk_always_wait")   h->walk_always_wait = value != 0;
    else if (k == "defer_walk")         { const int rcw = flush_walk(h); if (rcw) return rcw; h->defer_walk = value != 0; }
    else if (k == "dense_min")          { if (value < 0 || value > 0xffffffffll) return fail(h, GEM_ERR_INVALID, "dense_min: 0 .. 2^32 - 1"); h->dense_min = (unsigned)value; }
    else if (k == "dbg_sweep")          h->dbg_sweep = (int)value;
    else if (k == "dbg_frame")          h->dbg_frame = value != 0;
    else if (k == "overlap")            h->overlap = value != 0;
    else if (k == "overlap_min_points") { if (value < 0) return fail(h, GEM_ERR_INVALID, "overlap_min_points: >= 0"); h->overlap_min_points = value; h->sort_overlap_min_points = value; }
    else if (k == "sort_path")          h->sort_path = value != 0;
    else if (k == "sort_min_points")    { if (value < 0) return fail(h, GEM_ERR_INVALID, "sort_min_points: >= 0"); h->sort_min_points = value; h->sort_min_points_batch = value; }
    else if (k == "walk_permute")       h->walk_permute = value != 0;
    else if (k == "trace")              h->trace = value != 0;
    else if (k == "stream_roles") {
        // experiment: permute the roles of the handle's four streams; the decimal digits of `value` name, for own / bin / bin2 /
        // tab, which of the CURRENT (own, bin, bin2, tab) takes the role (e.g. 3210 reverses them)
        if (h->stream != h->own_stream) return fail(h, GEM_ERR_INVALID, "stream_roles: caller-provided stream in use");
        for (hipStream_t st : {h->own_stream, h->bin_stream, h->bin_stream2, h->tab_stream}) hipStreamSynchronize(st);
        const hipStream_t cur[4] = {h->own_stream, h->bin_stream, h->bin_stream2, h->tab_stream};
        const int d[4] = {(int)(value / 1000 % 10), (int)(value / 100 % 10), (int)(value / 10 % 10), (int)(value % 10)};
        bool seen[4] = {false, false, false, false};
        for (int i = 0; i < 4; ++i) { if (d[i] < 0 || d[i] > 3 || seen[d[i]]) return fail(h, GEM_ERR_INVALID, "stream_roles: not a permutation of 0123"); seen[d[i]] = true; }
        h->own_stream = cur[d[0]]; h->bin_stream = cur[d[1]]; h->bin_stream2 = cur[d[2]]; h->tab_stream = cur[d[3]];
        h->stream = h->own_stream;
    }
    else if (k == "sort_ring")          { if (value < 2 || value > 4) return fail(h, GEM_ERR_INVALID, "sort_ring: 2..4"); h->sort_ring = (int)value; }
    else if (k == "sort_streams")       { if (value != 1 && value != 2) return fail(h, GEM_ERR_INVALID, "sort_streams: 1 or 2"); h->sort_streams = (int)value; }
    else if (k == "sort_passes")        { if (value < 0 || value > 3) return fail(h, GEM_ERR_INVALID, "sort_passes: 0..3"); h->sort_passes = (int)value; }
    else if (k == "rank_by_ballot")     h->rank_by_ballot = value != 0;
    else if (k == "lane_sort")          h->lane_sort = value != 0;
    else if (k == "plain_loop")         h->plain_loop = value != 0;
    else if (k == "cache_tables")       h->cache_tables = value != 0;
    else if (k == "light_fast")         h->light_fast = value != 0;
    else if (k == "walk_lds_pad")       { if (value < 0 || value > 100 * 1024) return fail(h, GEM_ERR_INVALID, "walk_lds_pad: 0 .. 102400 bytes"); h->walk_lds_pad = (int)value; }
    else if (k == "ride_events")        { if (value < 0 || value > 1) return fail(h, GEM_ERR_INVALID, "ride_events: 0 or 1"); h->ride_events = value != 0; }
    else if (k == "download_groups")    { if (value < 1 || value > 14) return fail(h, GEM_ERR_INVALID, "download_groups: 1 .. 14"); h->download_groups = (int)value; }
    else if (k == "copy_threads")       { if (value < 0 || value > gem::CopyPool::kMaxThreads) return fail(h, GEM_ERR_INVALID, "copy_threads: 0 (the runtime's pageable path) .. 16"); h->copy_threads = (int)value; }
    else if (k == "fuse_count")         { if (value < 0 || value > 2) return fail(h, GEM_ERR_INVALID, "fuse_count: 0 (never), 1 (small passes) or 2 (always)"); h->fuse_count = (int)value; }
    else if (k == "sort_chunk")         { if (value != 0 && value != kSortChunkSmall && value != kSortChunkRecords) return fail(h, GEM_ERR_INVALID, "sort_chunk: 0 (by pass), 1024 or 4096"); h->sort_chunk = (int)value; }
    else if (k == "walk_prio")          { if (value < 0 || value > (1 << 30)) return fail(h, GEM_ERR_INVALID, "walk_prio: 0 (off) or a record count"); h->walk_prio = (int)value; }
    else if (k == "blk_batch")          { if (value != 0 && value != 512 && value != 2048) return fail(h, GEM_ERR_INVALID, "blk_batch: 0 (by pass), 512 or 2048"); h->blk_batch = (int)value; }
    else if (k == "few_bins")           { if (value < -1 || value > 64) return fail(h, GEM_ERR_INVALID, "few_bins: -1 (one ballot per digit bit), 0 (by pass), 1..64"); h->few_bins = (int)value; }
    else if (k == "ray_depth")          { if (value != 4 && value != 8) return fail(h, GEM_ERR_INVALID, "ray_depth: 4 or 8"); h->ray_depth = (int)value; }
    else if (k == "ray_lanes")          { if (value != 1 && value != 4 && value != 8 && value != 16) return fail(h, GEM_ERR_INVALID, "ray_lanes: 1, 4, 8 or 16"); h->ray_lanes = (int)value; }
    else if (k == "fast_laser")         h->fast_laser = value != 0;
    else if (k == "sort_form")          { if (value < 0 || value > 2) return fail(h, GEM_ERR_INVALID, "sort_form: 0 (by pass), 1 (cell-sorted), 2 (block-sorted)"); h->sort_form = (int)value; }
    else return fail(h, GEM_ERR_INVALID, "gem_debug_set: unknown key");
    return GEM_OK;
}

int gem_debug_get(gem_handle* h, const char* key, long long* out)
{
    if (!h || !key || !out) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    const std::string k(key);
    if (k == "arena_allocations") *out = h->arena_allocations;
    else if (k == "hstage_allocations") *out = h->hstage_allocations;
    else if (k == "copy_threads") *out = h->copy_threads;
    else if (k == "xfer_upload_memcpy_ns") *out = h->xfer_ns[0];
    else if (k == "xfer_upload_enqueue_ns") *out = h->xfer_ns[1];
    else if (k == "xfer_download_enqueue_ns") *out = h->xfer_ns[2];
    else if (k == "xfer_download_wait_ns") *out = h->xfer_ns[3];
    else if (k == "xfer_download_memcpy_ns") *out = h->xfer_ns[4];
    else if (k == "sort_fallbacks") *out = h->sort_fallbacks;
    else if (k == "walks_unwaited") *out = h->walks_unwaited;
    else if (k == "walks_left") *out = h->walks_left;
    else if (k == "step_pending") *out = h->step.valid ? 1 : 0;
    else if (k.rfind("step_", 0) == 0) {
        // time stamps of the LAST finished step of gem_add_sharded_device on W > 1 ranks (recorded while gem_set_timing is on; read
        // after gem_synchronize): nanoseconds between two of them
        int a = -1, b = -1;
        if (k == "step_exchange_ns") { a = 2; b = 3; }                 // the grouped send / recv of the sorted records
        else if (k == "step_walk_ns") { a = 4; b = 5; }                // k_fuse_block over the strip, all sources
        else if (k == "step_publish_ns") { a = 6; b = 7; }             // the copy of the own strip the all-gather's sends read (+ the hand-over to the gather stream)
        else if (k == "step_gather_ns") { a = 7; b = 8; }              // the all-gather of the layers
        else if (k == "step_exchange_to_walk_ns") { a = 3; b = 4; }    // hand-over communication stream -> handle's stream
        else return fail(h, GEM_ERR_INVALID, "gem_debug_get: unknown key");
        hipSetDevice(h->device);
        float ms = 0.f;
        // (not recorded: a value no pair of time stamps gives; two stamps taken on different hardware queues may come out a few
        //  microseconds apart the wrong way round)
        if (!h->ev_t[a] || hipEventElapsedTime(&ms, h->ev_t[a], h->ev_t[b]) != hipSuccess) { (void)hipGetLastError(); *out = -(1ll << 62); return GEM_OK; }
        *out = (long long)((double)ms * 1e6);
    }
    else return fail(h, GEM_ERR_INVALID, "gem_debug_get: unknown key");
    return GEM_OK;
}

// profiling aid (not part of the drop-in surface): per-tile cycle stamps of the last k_fuse launch
int gem_debug_fuse_stamps(gem_handle* h, int enable, unsigned long long* out, int max_tiles)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    h->dbg_on = enable != 0;
    if (out && h->dbg.p) {
        GEM_HIP(h, hipStreamSynchronize(h->stream));
        const int rows = h->dbg_rows ? h->dbg_rows : h->T;
        const int n = max_tiles < rows ? max_tiles : rows;
        GEM_HIP(h, hipMemcpy(out, h->dbg.p, (size_t)n * 16 * 8, hipMemcpyDeviceToHost));
        return n;
    }
    return 0;
}

// ---- multi-GPU: the communicators and the all-gather of the fused row strips --------------------------------------------------
int gem_comm_unique_id(void* out_128_bytes)
{
    if (!out_128_bytes) return GEM_ERR_INVALID;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return GEM_ERR_COMM;
    memcpy(out_128_bytes, &id, sizeof(id));
    return GEM_OK;
}

} // extern "C"

namespace {

// streams, events and strips of a handle that has just been given its two transports
int comm_attach(gem_handle* h, int nranks, int rank, bool tile_strips)
{
    h->nranks = nranks; h->rank = rank; h->tile_strips = tile_strips;
    if (!h->comm_stream) {
        GEM_HIP(h, acquire_comm_stream(h->device, &h->comm_stream));
        GEM_HIP(h, acquire_comm_stream(h->device, &h->gather_stream));
        for (hipEvent_t* e : {&h->ev_sorted, &h->ev_exchanged, &h->ev_bounds[0], &h->ev_bounds[1], &h->ev_walked[0], &h->ev_walked[1], &h->ev_vu[0], &h->ev_vu[1],
                              &h->ev_published[0], &h->ev_published[1], &h->ev_gathered[0], &h->ev_gathered[1]})
            GEM_HIP(h, hipEventCreateWithFlags(e, hipEventDisableTiming));
        for (hipEvent_t& e : h->ev_t) GEM_HIP(h, hipEventCreate(&e));
    }
    // row strips in STORAGE coordinates: Move never migrates data between devices
    const int tile_rows = (h->L + 31) / 32;
    for (int k = 0; k <= nranks; ++k)
        h->strip_row[k] = tile_strips ? std::min(h->L, 32 * (int)((long long)tile_rows * k / nranks)) : (int)((long long)h->L * k / nranks);
    h->row0 = h->strip_row[rank]; h->row1 = h->strip_row[rank + 1];
    // the two published copies of this rank's strip the all-gathers send from (six layers each): no allocation inside a step
    if (nranks > 1)
        for (int g = 0; g < 2; ++g) { const int rc = ensure(h, h->published[g], (size_t)(h->row1 - h->row0) * h->L * 4 * 6 + 256); if (rc) return rc; }
    return GEM_OK;
}

int comm_init_common(gem_handle* h, const void* unique_id_128_bytes, int nranks, int rank, bool tile_strips)
{
    if (!h || !unique_id_128_bytes || nranks <= 0 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }   // the strip changes below
    if (h->tp_x) return fail(h, GEM_ERR_COMM, "gem_comm_init: the handle already joined a communicator");
    ncclUniqueId id;
    memcpy(&id, unique_id_128_bytes, sizeof(id));
    std::unique_ptr<RcclTransport> x(new RcclTransport()), g(new RcclTransport());
    ncclResult_t r = ncclCommInitRank(&x->comm, nranks, id, rank);
    if (r != ncclSuccess) { x->comm = nullptr; return fail(h, GEM_ERR_COMM, ncclGetErrorString(r)); }
    // the layers' all-gather gets a communicator of its own (same ranks): together with its own stream, the 46 MB of step p
    // then travel beside step p + 1's boundary all-gather and record exchange instead of in front of them.  Every rank issues the
    // operations of the two communicators in the same order (exchange p, gather p, boundaries p + 1), as RCCL asks of
    // communicators used side by side.
    r = ncclCommSplit(x->comm, 0, rank, &g->comm, nullptr);
    if (r != ncclSuccess || !g->comm) { g->comm = x->comm; g->owns = false; }          // (no split: one communicator carries both, in order)
    x->nranks = g->nranks = nranks; x->rank = g->rank = rank;
    // (destruction order: the borrowed communicator first -- tp_g is declared after tp_x, members die in reverse order)
    h->tp_x = std::move(x); h->tp_g = std::move(g);
    return comm_attach(h, nranks, rank, tile_strips);
}

} // namespace

extern "C" {

int gem_comm_init(gem_handle* h, const void* unique_id_128_bytes, int nranks, int rank)
{
    return comm_init_common(h, unique_id_128_bytes, nranks, rank, false);
}

int gem_comm_init_tiles(gem_handle* h, const void* unique_id_128_bytes, int nranks, int rank)
{
    return comm_init_common(h, unique_id_128_bytes, nranks, rank, true);
}

// include/gem_hip_debug.h: W handles of THIS process on ONE device form a communicator whose collectives are device-to-device
// copies (gem_transport.hpp); every handle is driven by a host thread of its own, like a rank.
int gem_comm_init_loopback(gem_handle* h, long long world_id, int nranks, int rank, int tile_strips)
{
    if (!h || nranks <= 0 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcd = settle(h); if (rcd) return rcd; }
    if (h->tp_x) return fail(h, GEM_ERR_COMM, "gem_comm_init_loopback: the handle already joined a communicator");
    std::string why;
    std::shared_ptr<LoopWorld> w = loop_world(world_id, nranks, &why);
    if (!w) return fail(h, GEM_ERR_COMM, why.c_str());
    std::unique_ptr<LoopbackTransport> x(new LoopbackTransport()), g(new LoopbackTransport());
    if (!x->join(w, 0, rank)) return fail(h, GEM_ERR_COMM, x->err.c_str());
    if (!g->join(w, 1, rank)) return fail(h, GEM_ERR_COMM, g->err.c_str());
    h->tp_x = std::move(x); h->tp_g = std::move(g);
    return comm_attach(h, nranks, rank, tile_strips != 0);
}

int gem_get_strip(gem_handle* h, int* out_row0, int* out_row1)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    if (out_row0) *out_row0 = h->row0;
    if (out_row1) *out_row1 = h->row1;
    return GEM_OK;
}

} // extern "C"

namespace {

// Every rank's strip to every other rank, DIRECT: one send / receive pair per peer and layer in one group (xGMI is point-to-point:
// each peer has its own link; a ring would pass every strip through seven hops), strips of any sizes.  The sends read a PUBLISHED
// COPY of the strip, taken on the handle's stream behind everything enqueued so far (two copies rotate: the copy for gather k + 2
// waits for gather k's sends, not for gather k + 1's); the transfers run on the gather stream, through the gather communicator,
// and write the other ranks' strips only -- so the next steps' sort / exchange / walk of this rank's own strip go on beside them.
// Whatever observes the whole map (gem_get_layer, gem_synchronize, gem_move, ...) waits for them (wait_gather).
int gather_layers_locked(gem_handle* h, int with_attributes)
{
    const int W = h->nranks;
    if (W == 1) return GEM_OK;
    const int nl = with_attributes ? 6 : 2;
    void* ptrs[6] = {h->layers.elevation, h->layers.variance, h->layers.intensity, h->layers.colorR, h->layers.colorG, h->layers.colorB};
    const size_t own = (size_t)(h->strip_row[h->rank + 1] - h->strip_row[h->rank]) * h->L;       // 4-byte elements of this rank's strip
    const int g = (int)(h->gather_seq++ & 1u);
    int rc;
    if ((rc = ensure(h, h->published[g], own * 4 * 6 + 256))) return step_abort(h, rc);    // (sized by gem_comm_init*: no allocation here)
    unsigned char* pub = static_cast<unsigned char*>(h->published[g].p);
    if (h->gathered_recorded[g]) GEM_HIP_STEP(h, hipStreamWaitEvent(h->stream, h->ev_gathered[g], 0));       // the gather before last has sent this copy
    if (h->step_timed) GEM_HIP_STEP(h, hipEventRecord(h->ev_t[6], h->stream));
    if (own)
        for (int l = 0; l < nl; ++l)
            GEM_HIP_STEP(h, hipMemcpyAsync(pub + (size_t)l * own * 4, static_cast<unsigned char*>(ptrs[l]) + (size_t)h->strip_row[h->rank] * h->L * 4, own * 4,
                                      hipMemcpyDeviceToDevice, h->stream));
    GEM_HIP_STEP(h, hipEventRecord(h->ev_published[g], h->stream));
    GEM_HIP_STEP(h, hipStreamWaitEvent(h->gather_stream, h->ev_published[g], 0));
    if (h->step_timed) GEM_HIP_STEP(h, hipEventRecord(h->ev_t[7], h->gather_stream));
    Transport& tp = *h->tp_g;
    bool ok = tp.group_begin();
    for (int p = 0; p < W && ok; ++p) {
        if (p == h->rank) continue;
        const size_t theirs = (size_t)(h->strip_row[p + 1] - h->strip_row[p]) * h->L;
        for (int l = 0; l < nl && ok; ++l) {
            if (own) ok = tp.send(pub + (size_t)l * own * 4, own, p, h->gather_stream);
            if (ok && theirs) ok = tp.recv(static_cast<unsigned char*>(ptrs[l]) + (size_t)h->strip_row[p] * h->L * 4, theirs, p, h->gather_stream);
        }
    }
    ok = tp.group_end(h->gather_stream) && ok;
    if (!ok) return step_abort(h, fail(h, GEM_ERR_COMM, tp.err.c_str()));
    GEM_HIP_STEP(h, hipEventRecord(h->ev_gathered[g], h->gather_stream));
    if (h->step_timed) GEM_HIP_STEP(h, hipEventRecord(h->ev_t[8], h->gather_stream));
    h->gathered_recorded[g] = true;
    h->gather_outstanding[g] = true;
    return GEM_OK;
}

} // namespace

extern "C" {

int gem_allgather_layers(gem_handle* h, int with_attributes)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    if (!h->tp_g) return fail(h, GEM_ERR_COMM, "gem_allgather_layers: gem_comm_init not called");
    if (h->step.valid) {                              // behind a sharded step whose walk is still to come: part of that step's second half
        h->step.gather = true; h->step.gather_attrs = with_attributes;
        return GEM_OK;
    }
    { const int rc = flush_local(h); if (rc) return rc; }                                 // (a sorted pass's walk left to "the next call": this is the next call)
    if (h->n_pending) { const int rc = flush_pending(h, false); if (rc) return rc; }      // queued increments are part of what the peers get
    return gather_layers_locked(h, with_attributes);
}

} // extern "C"

namespace {

// ---- multi-GPU with the POINTS sharded (SURVEY 8e stage B) ---------------------------------------------------------------------
// Rank r holds a contiguous index range of the batch's points.  It projects, bins and sorts them for the WHOLE map
// (gem_shard_sort_device); the sorted records of every strip go to the strip's owner, which walks its cells through the
// sources in rank order -- ranks hold ascending index ranges, so rank order is input order and the result is the
// single-device one bit for bit (gem_shard_fuse_device).  gem_add_sharded_device does both with an RCCL exchange in between.
int shard_checks(gem_handle* h, int n_global_sweeps, SortGeometry* geo)
{
    if (h->track_lowest) return fail(h, GEM_ERR_INVALID, "sharded path: lowest tracking is not supported (use the replicated path)");
    *geo = sort_geometry(h, n_global_sweeps, true);            // block-sorted: a strip's records are one contiguous range, a block's too
    if (!geo->ok) return fail(h, GEM_ERR_INVALID, "sharded path: map or batch too large for the record key");
    return GEM_OK;
}

int shard_sort_locked(gem_handle* h, int n_local_sweeps, const gem_frame_params* params, const void* d_xyzi, const long long* offsets,
                             int first_global_sweep, int n_global_sweeps, int first_point_in_sweep, int nstrips, const int* strip_rows,
                             uint32_t* out_bounds, const void** out_d_hv, const void** out_d_key, const void** out_d_ranges, bool bounds_stay_on_device)
{
    if (first_point_in_sweep < 0) return fail(h, GEM_ERR_INVALID, "gem_shard_sort_device: negative first_point_in_sweep");
    if (n_local_sweeps < 0 || nstrips <= 0 || nstrips > kMaxRanks || !strip_rows || first_global_sweep < 0 ||
        first_global_sweep + n_local_sweeps > n_global_sweeps || (n_local_sweeps > 0 && (!params || !offsets || !d_xyzi)))
        return fail(h, GEM_ERR_INVALID, "gem_shard_sort_device: bad argument");
    hipSetDevice(h->device);
    for (int k = 0; k <= nstrips; ++k) {
        const bool ok = (strip_rows[k] % 32 == 0 || strip_rows[k] >= h->L) && strip_rows[k] >= 0 && (k == 0 || strip_rows[k] >= strip_rows[k - 1]);
        if (!ok) return fail(h, GEM_ERR_INVALID, "gem_shard_sort_device: strips must be whole rows of 32x32 tiles, ascending");
    }
    if (strip_rows[0] != 0 || strip_rows[nstrips] < h->L) return fail(h, GEM_ERR_INVALID, "gem_shard_sort_device: the strips must cover the map");
    SortGeometry geo;
    int rc = shard_checks(h, n_global_sweeps, &geo);
    if (rc) return rc;
    gem_handle::Shard& sd = h->shard;
    const long long n = n_local_sweeps > 0 ? offsets[n_local_sweeps] - offsets[0] : 0;
    if (n >= (1ll << 31)) return fail(h, GEM_ERR_INVALID, "gem_shard_sort_device: shard too large");
    if (n == 0) {                                        // an empty shard contributes nothing to any strip: no records, empty ranges, zero bounds
        const size_t n_blocks = (size_t)4 * geo.T;
        if ((rc = ensure(h, h->sh_ranges, n_blocks * sizeof(uint2)))) return rc;
        if ((rc = ensure(h, h->sh_dev, kShardDevBytes))) return rc;
        GEM_HIP(h, hipMemsetAsync(h->sh_ranges.p, 0, n_blocks * sizeof(uint2), h->stream));
        GEM_HIP(h, hipMemsetAsync(static_cast<uint32_t*>(h->sh_dev.p) + 16, 0, 16 * sizeof(uint32_t), h->stream));
        sd.valid = true; sd.hv = nullptr; sd.key = nullptr; sd.nstrips = nstrips; sd.slot = -1;
        sd.ranges = static_cast<const uint2*>(h->sh_ranges.p); sd.d_bounds = static_cast<const uint32_t*>(h->sh_dev.p) + 16;
        for (int k = 0; k <= nstrips; ++k) sd.bounds[k] = 0;
    } else {
        for (int s = 0; s < n_local_sweeps; ++s) if (offsets[s + 1] < offsets[s]) return fail(h, GEM_ERR_INVALID, "gem_shard_sort_device: offsets not monotone");
        PassInput in; in.src = 0; in.n_sweeps = n_local_sweeps; in.n = n; in.params = params; in.device_input = true;
        std::vector<long long> off0(n_local_sweeps + 1);
        for (int s = 0; s <= n_local_sweeps; ++s) off0[s] = offsets[s] - offsets[0];
        in.offsets = off0.data(); in.var_updates = nullptr;
        in.xyzi = static_cast<const float4*>(d_xyzi) + offsets[0];
        // a sweep split between two ranks: the camera models take the pixel row / column from the point's index INSIDE ITS SWEEP
        // (gem_device.hpp, sensor_variances), so the shard that holds a sweep's tail says where that tail begins
        std::vector<int> orig0(n_local_sweeps, 0);
        orig0[0] = first_point_in_sweep;
        in.sweep_orig0 = orig0.data();
        ShardOpts so{first_global_sweep, nstrips, strip_rows, bounds_stay_on_device};
        if ((rc = run_sort_pipeline(h, in, 0, geo, &so))) return rc;
    }
    sd.n_global_sweeps = n_global_sweeps;
    sd.points = n;
    if (out_bounds) for (int k = 0; k <= nstrips; ++k) out_bounds[k] = sd.bounds[k];
    if (out_d_hv) *out_d_hv = sd.hv;
    if (out_d_key) *out_d_key = sd.key;
    if (out_d_ranges) *out_d_ranges = sd.ranges;
    return GEM_OK;
}

} // namespace

extern "C" {

int gem_shard_sort_device(gem_handle* h, int n_local_sweeps, const gem_frame_params* params, const void* d_xyzi, const long long* offsets,
                          int first_global_sweep, int n_global_sweeps, int first_point_in_sweep, int nstrips, const int* strip_rows,
                          uint32_t* out_bounds, const void** out_d_hv, const void** out_d_key, const void** out_d_ranges)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    return shard_sort_locked(h, n_local_sweeps, params, d_xyzi, offsets, first_global_sweep, n_global_sweeps, first_point_in_sweep, nstrips, strip_rows,
                             out_bounds, out_d_hv, out_d_key, out_d_ranges, false);
}

} // extern "C"

namespace {

// d_ranges / bases (both or neither): per source the block ranges of ITS sorted records, entry 0 = the first block of this handle's
// strip, and the position d_hv[s] / d_key[s] point at in the source's own arrays; without them the walk searches every source.
// own: this device's own sorted records are the ONLY source (one rank), taken in place through their block ranges.
// slot: the pass-buffer set whose sort the walk reads (its fuse_done event lets the sort after next reuse the buffers), -1: none.
// recv_parity: the set of receive buffers the sources live in (its ev_walked tells the exchange after next that they have been read), -1: none.
int shard_fuse_locked(gem_handle* h, int n_src, const void* const* d_hv, const void* const* d_key, const uint32_t* counts,
                      const void* const* d_ranges, const uint32_t* bases, int n_global_sweeps, const float* var_updates_global,
                      const gem_handle::Shard* own = nullptr, int slot = -1, int recv_parity = -1, long long step_points = -1)
{
    SortGeometry geo;
    int rc = shard_checks(h, n_global_sweeps, &geo);
    if (rc) return rc;
    { const int rcd = flush_local(h); if (rcd) return rcd; }            // an earlier pass's walk fuses BEFORE this one (the recurrence is order dependent)
    WalkArgs wa{};
    wa.n_src = own ? 1 : std::max(n_src, 2);             // the multi-source form (a single source is followed by an empty one) unless the records are this device's own
    if (own) { wa.hv = own->hv; wa.key = own->key; wa.ranges = own->ranges; }
    for (int s = 0; s < kMaxRanks && !own; ++s) {
        const bool ranged = s < n_src && d_ranges && bases && d_ranges[s] && d_hv[s] && d_key[s];
        const bool on = ranged || (s < n_src && counts && counts[s] > 0);
        wa.src_hv[s] = on ? static_cast<const uint2*>(d_hv[s]) : nullptr;
        wa.src_key[s] = on ? static_cast<const uint32_t*>(d_key[s]) : nullptr;
        wa.src_n[s] = on && counts ? counts[s] : 0u;
        wa.src_ranges[s] = ranged ? static_cast<const uint2*>(d_ranges[s]) : nullptr;
        wa.src_base[s] = ranged ? bases[s] : 0u;
    }
    wa.blk0 = (uint32_t)((h->row0 / 32) * geo.tiles_per_row) << 2;    // first block of this handle's strip (whole tile rows)
    wa.T = geo.T; wa.tiles_per_row = geo.tiles_per_row; wa.L = h->L; wa.row0 = h->row0; wa.row1 = h->row1;
    wa.id_bits = geo.id_bits; wa.bin_shift = geo.dshift[geo.n_passes - 1]; wa.n_sweeps = n_global_sweeps;
    wa.mahal = h->cfg.mahalanobis_threshold; wa.var_floor = h->cfg.variance_floor;
    wa.dense = (h->n_pending > 0 || h->floor_dirty || var_updates_global != nullptr) ? 1 : 0;
    wa.n_pending = h->n_pending;
    for (int i = 0; i < kMaxPending; ++i) wa.pending[i] = h->pending[i];
    wa.plain_env = h->plain_loop ? walk_plain_env(wa.var_floor, wa.mahal, h->pending, h->n_pending, var_updates_global, n_global_sweeps) : 0;
    wa.prio_records = h->walk_prio; wa.lds_pad = h->walk_lds_pad; wa.light_fast = h->light_fast ? 1 : 0;
    wa.elevation = h->layers.elevation; wa.variance = h->layers.variance; wa.lowest = h->layers.lowest;
    wa.start0 = h->start[0]; wa.start1 = h->start[1];
    wa.counters = h->counting ? h->d_counters : nullptr;
    wa.count_per_pass = 0;
    wa.walk_order = (h->walk_permute && 4ll * geo.T <= 4096) ? 1 : 0;
    wa.lane_sort = h->lane_sort ? 1 : 0;
    {   // rounds of 512 records when the strip's blocks are light: about as many records arrive as this rank sorted (its share of the step)
        const long long strip_blocks = 4ll * ((std::min(h->row1, h->L) - h->row0 + 31) / 32) * geo.tiles_per_row;
        const long long pts = step_points >= 0 ? step_points : (h->shard.valid ? h->shard.points : -1);
        wa.light_blocks = h->blk_batch ? (h->blk_batch <= 512 ? 1 : 0) : (pts >= 0 && pts <= 768ll * strip_blocks ? 1 : 0);
        if (wa.light_blocks) wa.lane_sort = 0;
    }
    wa.center_tr = ((h->L / 2 + h->start[0]) % h->L) >> 5;
    if (var_updates_global) {
        // staged in pinned memory, two buffers in turn: the upload from a buffer is long done when its turn comes again (the event
        // is there for the caller who gets ahead), so no step waits for the handle's stream here
        if (n_global_sweeps > 512) return fail(h, GEM_ERR_INVALID, "sharded path: more than 512 sweeps");
        if (!h->sh_host) GEM_HIP(h, hipHostMalloc(&h->sh_host, kShardHostBytes, hipHostMallocDefault));
        if ((rc = ensure(h, h->sh_dev, kShardDevBytes))) return rc;
        const int b = (int)(h->vu_seq++ & 1u);
        if (!h->ev_vu[b]) GEM_HIP(h, hipEventCreateWithFlags(&h->ev_vu[b], hipEventDisableTiming));
        if (h->vu_recorded[b]) GEM_HIP(h, hipEventSynchronize(h->ev_vu[b]));
        float* hostf = reinterpret_cast<float*>(static_cast<unsigned char*>(h->sh_host) + 8192 + 2048 * b);
        memcpy(hostf, var_updates_global, sizeof(float) * n_global_sweeps);
        float* dv = reinterpret_cast<float*>(static_cast<unsigned char*>(h->sh_dev.p) + 8192 + 2048 * b);
        GEM_HIP(h, hipMemcpyAsync(dv, hostf, sizeof(float) * n_global_sweeps, hipMemcpyHostToDevice, h->stream));
        GEM_HIP(h, hipEventRecord(h->ev_vu[b], h->stream)); h->vu_recorded[b] = true;
        wa.var_updates = dv;
    }
    if (h->counting) GEM_HIP(h, hipMemsetAsync(h->d_counters, 0, 2 * sizeof(unsigned long long), h->stream));
    if (h->step_timed) GEM_HIP(h, hipEventRecord(h->ev_t[4], h->stream));
    { Timed t(h, 9); GEM_HIP(h, launch_block_walk(h->stream, wa, 0, t.events())); }
    if (h->step_timed) GEM_HIP(h, hipEventRecord(h->ev_t[5], h->stream));
    if (recv_parity >= 0) { GEM_HIP(h, hipEventRecord(h->ev_walked[recv_parity], h->stream)); h->walk_recorded[recv_parity] = true; }    // (the receive buffers have been read)
    if (slot >= 0) { GEM_HIP(h, hipEventRecord(h->pb[slot].fuse_done, h->stream)); h->pb[slot].fuse_recorded = true; }
    else h->main_reads_pb = true;
    h->n_pending = 0;
    h->floor_dirty = false;
    return GEM_OK;
}

// a rank that cannot go on between two collectives of a step: the peers' pending calls fail instead of hanging
int step_abort(gem_handle* h, int rc)
{
    if (h->tp_x) h->tp_x->abort();
    if (h->tp_g) h->tp_g->abort();
    return rc;
}

size_t strip_blocks_of(const gem_handle* h, int p)
{
    const int tpr = (h->L + 31) / 32;
    return (size_t)4 * tpr * ((std::min(h->strip_row[p + 1], tpr * 32) + 31) / 32 - h->strip_row[p] / 32);
}

// receive buffers of one parity for up to `records` records (and the W tables of block ranges)
int ensure_recv(gem_handle* h, int q, size_t records)
{
    int rc;
    if ((rc = ensure(h, h->sh_recv_hv[q], records * 8 + 64 + 32 * kMaxRanks))) return rc;
    if ((rc = ensure(h, h->sh_recv_key[q], records * 4 + 64 + 16 * kMaxRanks))) return rc;
    return ensure(h, h->sh_recv_rng[q], (size_t)h->nranks * strip_blocks_of(h, h->rank) * sizeof(uint2) + 64);
}

// The SECOND HALF of a gem_add_sharded_device step on W > 1 ranks: the gathered strip boundaries (on the host by now: the sort
// they waited for was enqueued a call ago) say what this rank sends and receives; one group of sends / receives moves every
// strip's records and block ranges to its owner on the communication stream; the walk follows on the handle's stream, and the
// all-gather of the layers, if gem_allgather_layers was called behind the step, on the gather stream.
int shard_finish_locked(gem_handle* h)
{
    if (!h->step.valid) return GEM_OK;
    gem_handle::Step& st = h->step;
    st.valid = false;                                                 // (whatever happens below, the step is not retried)
    hipSetDevice(h->device);
    const int W = h->nranks, q = st.parity;
    gem_handle::Shard& sd = st.sd;
    uint32_t* host = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(h->sh_host) + 4096 * q);
    GEM_HIP_STEP(h, hipEventSynchronize(h->ev_bounds[q]));
    for (int k = 0; k <= W; ++k) sd.bounds[k] = host[64 + h->rank * 16 + k];
    const int tpr = (h->L + 31) / 32;
    const size_t my_blocks = strip_blocks_of(h, h->rank);
    const uint32_t my_blk0 = (uint32_t)((h->row0 / 32) * tpr) << 2;
    uint32_t cnt[kMaxRanks], off[kMaxRanks + 1], base[kMaxRanks];
    off[0] = 0;
    long long arriving = 0;
    for (int s = 0; s < W; ++s) {
        base[s] = host[64 + s * 16 + h->rank];
        cnt[s] = host[64 + s * 16 + h->rank + 1] - base[s];
        off[s + 1] = off[s] + (s == h->rank ? 0u : ((cnt[s] + 3u) & ~3u));      // (this rank's own records stay where they are)
        arriving += cnt[s];
    }
    // sized before the step's first collective (gem_add_sharded_device); a step that brings more than was foreseen grows them here,
    // and a rank that cannot takes the communicators down with it rather than leave the others waiting in their receives
    int rc;
    if ((rc = ensure_recv(h, q, off[W]))) return step_abort(h, rc);
    uint2* rhv = static_cast<uint2*>(h->sh_recv_hv[q].p); uint32_t* rkey = static_cast<uint32_t*>(h->sh_recv_key[q].p);
    uint2* rrng = static_cast<uint2*>(h->sh_recv_rng[q].p);
    // the walk before last has read this parity's receive buffers
    if (h->walk_recorded[q]) GEM_HIP_STEP(h, hipStreamWaitEvent(h->comm_stream, h->ev_walked[q], 0));
    if (h->step_timed) GEM_HIP_STEP(h, hipEventRecord(h->ev_t[2], h->comm_stream));
    // the exchange: every strip's records and their block ranges to the strip's owner
    Transport& tp = *h->tp_x;
    bool ok = tp.group_begin();
    for (int p = 0; p < W && ok; ++p) {
        if (p == h->rank) continue;
        const uint32_t sc = sd.bounds[p + 1] - sd.bounds[p];
        if (sc > 0) {
            const uint32_t p_blk0 = (uint32_t)((h->strip_row[p] / 32) * tpr) << 2;
            ok = tp.send(sd.hv + sd.bounds[p], (size_t)sc * 2, p, h->comm_stream) &&
                 tp.send(sd.key + sd.bounds[p], sc, p, h->comm_stream) &&
                 tp.send(sd.ranges + p_blk0, strip_blocks_of(h, p) * 2, p, h->comm_stream);
        }
        if (ok && cnt[p] > 0)
            ok = tp.recv(rhv + off[p], (size_t)cnt[p] * 2, p, h->comm_stream) &&
                 tp.recv(rkey + off[p], cnt[p], p, h->comm_stream) &&
                 tp.recv(rrng + (size_t)p * my_blocks, my_blocks * 2, p, h->comm_stream);
    }
    ok = tp.group_end(h->comm_stream) && ok;
    if (!ok) return step_abort(h, fail(h, GEM_ERR_COMM, tp.err.c_str()));
    GEM_HIP_STEP(h, hipEventRecord(h->ev_exchanged, h->comm_stream));
    if (h->step_timed) GEM_HIP_STEP(h, hipEventRecord(h->ev_t[3], h->comm_stream));
    GEM_HIP_STEP(h, hipStreamWaitEvent(h->stream, h->ev_exchanged, 0));
    const void* phv[kMaxRanks]; const void* pkey[kMaxRanks]; const void* prng[kMaxRanks];
    for (int s = 0; s < W; ++s) {
        const bool mine = s == h->rank;
        phv[s] = cnt[s] ? (mine ? (const void*)(sd.hv + base[s]) : (const void*)(rhv + off[s])) : nullptr;
        pkey[s] = cnt[s] ? (mine ? (const void*)(sd.key + base[s]) : (const void*)(rkey + off[s])) : nullptr;
        prng[s] = cnt[s] ? (mine ? (const void*)(sd.ranges + my_blk0) : (const void*)(rrng + (size_t)s * my_blocks)) : nullptr;
    }
    if ((rc = shard_fuse_locked(h, W, phv, pkey, cnt, prng, base, st.n_global_sweeps, st.has_vu ? st.vu : nullptr, nullptr, sd.slot, q, arriving))) return step_abort(h, rc);
    if (st.gather) { st.gather = false; return gather_layers_locked(h, st.gather_attrs); }
    return GEM_OK;
}

// would run_sort_pipeline put a shard's sort of n points on a binning stream, in a pass-buffer set of its own?
bool shard_sort_rotates(const gem_handle* h, long long n)
{
    return n > 0 && h->overlap && n >= std::min(h->overlap_min_points, h->sort_overlap_min_points) && h->stream == h->own_stream && !h->counting;
}

} // namespace

extern "C" {

int gem_shard_fuse_device(gem_handle* h, int n_src, const void* const* d_hv, const void* const* d_key, const uint32_t* counts,
                          const void* const* d_ranges, const uint32_t* bases, int n_global_sweeps, const float* var_updates_global)
{
    if (!h || n_src <= 0 || n_src > kMaxRanks || !d_hv || !d_key || !counts || n_global_sweeps <= 0 || ((d_ranges == nullptr) != (bases == nullptr)))
        return h ? fail(h, GEM_ERR_INVALID, "gem_shard_fuse_device: bad argument") : GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    { const int rcs = shard_finish_locked(h); if (rcs) return rcs; }
    if (h->tile_strips == false && (h->row0 % 32 != 0)) return fail(h, GEM_ERR_INVALID, "gem_shard_fuse_device: the handle's strip must start at a tile row");
    return shard_fuse_locked(h, n_src, d_hv, d_key, counts, d_ranges, bases, n_global_sweeps, var_updates_global);
}

// One step of the map tiled over the ranks.  W > 1, in the order things are enqueued by call p:
//   sort p            binning streams   this rank's points, block-sorted for the whole map (a pass-buffer set of its own, three rotate)
//   [second half of step p - 1: shard_finish_locked]
//       exchange p-1  communication stream / exchange communicator
//       walk p-1      the handle's stream
//       gather p-1    gather stream / gather communicator          (when gem_allgather_layers followed the step)
//   boundaries p      communication stream: all-gather of the W + 1 strip boundaries of every rank's sorted records, copied to the host
// and the call returns.  Nothing in it waits for work the same call enqueued: the host's one wait -- for the boundaries of step
// p - 1 -- has the sort of step p queued behind it.  Per stream the steady state is sort | exchange + boundaries | walk | gather,
// each on its own queue: a step takes as long as the slowest of them, not their sum.
// Every rank issues the same sequence of collectives on each communicator: the calls, their order and the flush points
// (settle) are the same on all ranks by the API's contract.
int gem_add_sharded_device(gem_handle* h, int n_local_sweeps, const gem_frame_params* params, const void* d_xyzi, const long long* offsets,
                           int first_global_sweep, int n_global_sweeps, int first_point_in_sweep, const float* var_updates_global)
{
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    if (!h->tp_x || !h->tile_strips) return fail(h, GEM_ERR_COMM, "gem_add_sharded_device: gem_comm_init_tiles not called");
    const int W = h->nranks;
    hipSetDevice(h->device);
    // Everything that can fail on this rank alone -- arguments, geometry, allocations -- fails HERE, before the step's first
    // collective: a rank that returned early would leave the others waiting in theirs.
    const long long n_local = (n_local_sweeps > 0 && offsets) ? offsets[n_local_sweeps] - offsets[0] : 0;
    {
        SortGeometry geo;
        int rc0 = shard_checks(h, n_global_sweeps, &geo);
        if (rc0) return rc0;
        if (n_local_sweeps < 0 || first_global_sweep < 0 || first_global_sweep + n_local_sweeps > n_global_sweeps || first_point_in_sweep < 0 ||
            (n_local_sweeps > 0 && (!params || !offsets || !d_xyzi)) || n_local < 0 || n_local >= (1ll << 31) || n_global_sweeps > 512)
            return fail(h, GEM_ERR_INVALID, "gem_add_sharded_device: bad argument");
        if (!h->sh_host) GEM_HIP(h, hipHostMalloc(&h->sh_host, kShardHostBytes, hipHostMallocDefault));
        if ((rc0 = ensure(h, h->sh_dev, kShardDevBytes))) return rc0;
        if (W > 1) {
            // both sets of receive buffers, for what a step can bring at most: gem_reserve's bound when there is
            // one, else W shares like this rank's (the ranks hold N / W points each, and no strip gets more records than there are points)
            const long long bound = h->recv_bound > 0 ? h->recv_bound : (n_local + 1) * W;
            if ((rc0 = ensure_recv(h, 0, (size_t)bound + 4 * W)) || (rc0 = ensure_recv(h, 1, (size_t)bound + 4 * W))) return rc0;
        }
    }
    // A pending step's sorted records live in a pass-buffer set of their own only if its sort rotated (big shards on the handle's
    // own streams); a sort that does not -- small shards, a caller's stream -- would overwrite them: finish the pending step first.
    if (h->step.valid && !(h->step.sd.slot >= 0 && shard_sort_rotates(h, n_local))) { const int rcs = shard_finish_locked(h); if (rcs) return rcs; }
    // the sort leaves this rank's strip boundaries on the device (k_strip_bounds' output, 16 words reserved) ...
    int rc = shard_sort_locked(h, n_local_sweeps, params, d_xyzi, offsets, first_global_sweep, n_global_sweeps, first_point_in_sweep, W, h->strip_row,
                               nullptr, nullptr, nullptr, nullptr, true);
    if (rc) return W > 1 ? step_abort(h, rc) : rc;
    hipSetDevice(h->device);
    gem_handle::Shard& sd = h->shard;
    hipStream_t sorted_on = sd.slot >= 0 ? nullptr : h->stream;
    if (W == 1) {
        // one rank: its own sorted records, in place, through their block ranges -- no exchange, nothing returns to the host
        const uint32_t cnt[1] = {0u};
        const void* none[1] = {nullptr};
        if (sd.slot >= 0) GEM_HIP(h, hipStreamWaitEvent(h->stream, h->pb[sd.slot].bin_done, 0));
        if (!sd.hv) return shard_fuse_locked(h, 1, none, none, cnt, nullptr, nullptr, n_global_sweeps, var_updates_global);
        return shard_fuse_locked(h, 1, none, none, cnt, nullptr, nullptr, n_global_sweeps, var_updates_global, &sd, sd.slot);
    }
    // the second half of the step before (its boundaries are on the host, or will be as soon as its sort is through)
    if ((rc = shard_finish_locked(h))) return rc;                     // (aborts the communicators itself when it fails)
    // ... and every rank learns what it receives from whom -- and what it sends -- from ONE all-gather of them (16 words per rank),
    // copied to the host behind it on the communication stream; the next call (or settle) picks them up
    const int q = (int)(h->step_seq++ & 1u);
    uint32_t* host = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(h->sh_host) + 4096 * q);
    uint32_t* d_all = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(h->sh_dev.p) + 4096 * q) + 64;   // [W][16]
    // (the peers read this rank's sorted records over xGMI behind this edge: an event WITH the system-scope fence, recorded on the
    //  stream the sort ran on -- not the fence-less bin_done of the pass buffers, which orders this device's own streams only)
    if (!sorted_on) sorted_on = sd.stream ? sd.stream : h->stream;
    GEM_HIP_STEP(h, hipEventRecord(h->ev_sorted, sorted_on)); GEM_HIP_STEP(h, hipStreamWaitEvent(h->comm_stream, h->ev_sorted, 0));
    if (h->step_timed) { GEM_HIP_STEP(h, hipEventRecord(h->ev_t[1], h->comm_stream)); }
    if (!h->tp_x->all_gather(sd.d_bounds, d_all, 16, h->comm_stream)) return step_abort(h, fail(h, GEM_ERR_COMM, h->tp_x->err.c_str()));
    GEM_HIP_STEP(h, hipMemcpyAsync(host + 64, d_all, (size_t)W * 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, h->comm_stream));
    GEM_HIP_STEP(h, hipEventRecord(h->ev_bounds[q], h->comm_stream));
    gem_handle::Step& st = h->step;
    st.valid = true; st.parity = q; st.n_global_sweeps = n_global_sweeps; st.sd = sd; st.gather = false; st.gather_attrs = 0;
    st.has_vu = var_updates_global != nullptr;
    if (st.has_vu) memcpy(st.vu, var_updates_global, sizeof(float) * n_global_sweeps);
    return GEM_OK;
}

} // extern "C"
