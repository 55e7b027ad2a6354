// gem_capi_comm.cpp -- the multi-GPU part of the C ABI (see gem_capi_internal.hpp): communicators (RCCL, or the
// loopback of include/gem_hip_debug.h), the all-gather of the fused layers, the sharded step and its two halves.
#include "gem_capi_internal.hpp"


namespace gemi {

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

} // namespace gemi

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

namespace gemi {

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
    h->gbytes_out = h->gbytes_in = 0;
    for (int p = 0; p < W && ok; ++p) {
        if (p == h->rank) continue;
        const size_t theirs = (size_t)(h->strip_row[p + 1] - h->strip_row[p]) * h->L;
        h->gbytes_out += 4ll * (long long)own * nl; h->gbytes_in += 4ll * (long long)theirs * nl;
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

} // namespace gemi

extern "C" {

int gem_allgather_layers(gem_handle* h, int with_attributes)
{
    ApiRange api_range(h, "gem_allgather_layers");
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

namespace gemi {

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

} // namespace gemi

extern "C" {

int gem_shard_sort_device(gem_handle* h, int n_local_sweeps, const gem_frame_params* params, const void* d_xyzi, const long long* offsets,
                          int first_global_sweep, int n_global_sweeps, int first_point_in_sweep, int nstrips, const int* strip_rows,
                          uint32_t* out_bounds, const void** out_d_hv, const void** out_d_key, const void** out_d_ranges)
{
    ApiRange api_range(h, "gem_shard_sort_device");
    if (!h) return GEM_ERR_INVALID;
    std::lock_guard<std::mutex> lk(h->mu);
    return shard_sort_locked(h, n_local_sweeps, params, d_xyzi, offsets, first_global_sweep, n_global_sweeps, first_point_in_sweep, nstrips, strip_rows,
                             out_bounds, out_d_hv, out_d_key, out_d_ranges, false);
}

} // extern "C"

namespace gemi {

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
    h->xbytes_out = h->xbytes_in = 0;
    for (int p = 0; p < W && ok; ++p) {
        if (p == h->rank) continue;
        const uint32_t sc = sd.bounds[p + 1] - sd.bounds[p];
        if (sc > 0) h->xbytes_out += 12ll * sc + 8ll * (long long)strip_blocks_of(h, p);
        if (cnt[p] > 0) h->xbytes_in += 12ll * cnt[p] + 8ll * (long long)my_blocks;
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

} // namespace gemi

extern "C" {

int gem_shard_fuse_device(gem_handle* h, int n_src, const void* const* d_hv, const void* const* d_key, const uint32_t* counts,
                          const void* const* d_ranges, const uint32_t* bases, int n_global_sweeps, const float* var_updates_global)
{
    ApiRange api_range(h, "gem_shard_fuse_device");
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
    ApiRange api_range(h, "gem_add_sharded_device");
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
