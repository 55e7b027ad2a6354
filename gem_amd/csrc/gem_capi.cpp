// gem_capi.cpp -- the C ABI of libgem_hip.so (see include/gem_hip.h): handle, persistent device
// arenas, host-side Move logic and launch orchestration for the gfx950 kernels.
//
// The reference keeps the map as hidden process-global __device__ state and does 15 cudaMalloc +
// 15 cudaFree + 15 cudaMemcpy per frame on this path (gpu_process.cu:1096-1141, 1165-1192).
// Here a handle owns persistent arenas that only ever grow, everything is enqueued on one HIP
// stream, and nothing returns to the host unless the caller asks for it.
#include "gem_capi_internal.hpp"

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
    ApiRange api_range(h, "gem_synchronize");
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
    ApiRange api_range(h, "gem_move");
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
    ApiRange api_range(h, "gem_process_points");
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
    ApiRange api_range(h, "gem_fuse");
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
    ApiRange api_range(h, "gem_add_device");
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
    ApiRange api_range(h, "gem_add");
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
    ApiRange api_range(h, "gem_add_batch");
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
    ApiRange api_range(h, "gem_add_aos");
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
    ApiRange api_range(h, "gem_add_batch_device");
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
    ApiRange api_range(h, "gem_mapvar_update");
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
    ApiRange api_range(h, "gem_get_layer");
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
    ApiRange api_range(h, "gem_map_feature");
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
    ApiRange api_range(h, "gem_show");
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
    ApiRange api_range(h, "gem_colorize_device");
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    hipSetDevice(h->device);
    return colorize_device(h, cam, n, d_xyzi, d_image_bgr, row_stride, d_rgb);
}

int gem_colorize(gem_handle* h, const gem_camera* cam, int n, float* xyzi, const unsigned char* image_bgr, size_t row_stride, uint32_t* rgb)
{
    ApiRange api_range(h, "gem_colorize");
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
    ApiRange api_range(h, "gem_raytracing");
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
    else if (k == "roctx")              { if (value != 0 && !roctx_load()) return fail(h, GEM_ERR_INVALID, "roctx: no ROCm marker library (librocprofiler-sdk-roctx / libroctx64) to load"); h->roctx = value != 0; }
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
    else if (k == "step_exchange_bytes_out") *out = h->xbytes_out;
    else if (k == "step_exchange_bytes_in") *out = h->xbytes_in;
    else if (k == "gather_bytes_out") *out = h->gbytes_out;
    else if (k == "gather_bytes_in") *out = h->gbytes_in;
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
