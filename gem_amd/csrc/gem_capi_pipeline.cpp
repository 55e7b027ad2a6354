// gem_capi_pipeline.cpp -- one pass over points that are on the device (see gem_capi_internal.hpp): the tile pipeline (k_frame / k_bin_wave +
// k_fuse_list) and the two sorted forms (gem_sort.hip), their buffers, streams and hand-overs.
#include "gem_capi_internal.hpp"

namespace gemi {

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

int run_sort_pipeline(gem_handle* h, const PassInput& in, int attr, const SortGeometry& geo, const ShardOpts* shard)
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


} // namespace gemi
