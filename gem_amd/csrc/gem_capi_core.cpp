// gem_capi_core.cpp -- the handle's plumbing (see gem_capi_internal.hpp): errors, persistent arenas, the pinned staging buffer and the
// host <-> device transfers, the frame constants, the launches a call may leave to the next one, the process-lifetime stream pools.
//
// The reference keeps the map as hidden process-global __device__ state and does 15 cudaMalloc +
// 15 cudaFree + 15 cudaMemcpy per frame on this path (gpu_process.cu:1096-1141, 1165-1192).
// Here a handle owns persistent arenas that only ever grow, everything is enqueued on one HIP
// stream, and nothing returns to the host unless the caller asks for it.
#include "gem_capi_internal.hpp"

#include <dlfcn.h>

namespace gemi {

thread_local std::string g_create_error;

namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)();
std::once_flag g_roctx_once;
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;
}

bool roctx_load()
{
    std::call_once(g_roctx_once, [] {
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            void* so = dlopen(lib, RTLD_NOW | RTLD_GLOBAL);
            if (!so) continue;
            g_roctx_push = reinterpret_cast<roctx_push_fn>(dlsym(so, "roctxRangePushA"));
            g_roctx_pop = reinterpret_cast<roctx_pop_fn>(dlsym(so, "roctxRangePop"));
            if (g_roctx_push && g_roctx_pop) return;
            g_roctx_push = nullptr; g_roctx_pop = nullptr;
        }
    });
    return g_roctx_push != nullptr;
}
void roctx_push(const char* name) { if (g_roctx_push) g_roctx_push(name); }
void roctx_pop() { if (g_roctx_pop) g_roctx_pop(); }
ApiRange::ApiRange(const gem_handle* h, const char* name) : on(h && h->roctx && g_roctx_push != nullptr) { if (on) roctx_push(name); }

int fail(gem_handle* h, int code, const char* what, hipError_t e)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof(buf), "%s", what);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

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
int upload_arrays(gem_handle* h, const HostXfer* x, int n, bool defer_ok, unsigned char** zero_copy_region, int* zero_copy_half)
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

hipError_t acquire_streams(int device, StreamSet& out)
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

hipError_t acquire_comm_stream(int device, hipStream_t* out)
{
    {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        if (device >= 0 && device < 64 && !g_comm_pool[device].empty()) { *out = g_comm_pool[device].back(); g_comm_pool[device].pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void release_comm_stream(int device, hipStream_t st)
{
    if (!st) return;
    hipStreamSynchronize(st);
    if (device < 0 || device >= 64) { hipStreamDestroy(st); return; }
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    g_comm_pool[device].push_back(st);
}

void release_streams(int device, const StreamSet& set)
{
    if (!set.s[0]) return;
    for (hipStream_t st : set.s) hipStreamSynchronize(st);
    if (device < 0 || device >= 64) { for (hipStream_t st : set.s) hipStreamDestroy(st); return; }
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    g_stream_pool[device].push_back(set);
}

} // namespace gemi
