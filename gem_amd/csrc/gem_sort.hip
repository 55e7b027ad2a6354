// gem_sort.hip -- the SORTED pipelines of the GEM hot path for big passes (batches of sweeps, aggregated clouds, depth images).
//
// G_fuse (gpu_process.cu:477-537) is one thread per CELL scanning all points in input order; the recurrence is not
// associative (variance floor inside the loop, Mahalanobis branch), so what every cell needs is ITS points, in input order.
// For a stream of single LiDAR sweeps that list is short and k_frame (gem_kernels.hip) builds it per tile in LDS.  For a big
// pass -- millions of points, tens of sweeps -- the lists are long, the tile under the sensor carries a hundred times the
// records of a tile at the rim, and any per-tile batching is bound by that one tile.  Here the lists are built by the whole
// chip instead: a stable LSD counting sort of the in-map points, every pass split into equal 4096-record chunks, in two forms
// (gem_kernels.hpp; gem_capi.cpp picks one per pass, both give the same map):
//
//   k_sort_project  chunk of 4096 points, four waves of 1024: project + bin (G_pointsprocess, GPU:384-455) ONCE; {h, var} and the
//                   key {id | sweep} of every point that stays, stored in input order, compacted per wave; histogram over the
//                   first digit in LDS                                                        -> cnt[chunk][bins0]
//   k_sort_scan     column-wise exclusive prefix over the chunks (four segments), segment sums -> cnt (in place), segtot, M
//   k_sort_scatter  the same chunk again, records only: STABLE rank inside the chunk (wave w owns the w-th contiguous share,
//                   64 consecutive records per step; equal bins matched through the LDS or by ballot, a per-wave cursor per
//                   bin in LDS), the chunk staged in LDS in sorted order and written out from there
//   k_sort_count / k_sort_scan / k_sort_scatter   the same over the next digit on the records of the pass before
//
//   CELL-sorted:  digits over the whole id (tile << 10 | cell in tile), two passes (three beyond 2^20 cells);
//     k_fuse_walk   256 consecutive cells per workgroup: a 32-ary search finds their share of the sorted records, one look at
//                   its keys gives the cell boundaries, then every lane streams its own cell's run through the reference's
//                   recurrence (GPU:480-531), the sweeps' variance increments (GPU:540-547) and floors (GPU:533-534) replayed
//                   per cell in between.
//   BLOCK-sorted: digits over the block id (id >> 8: eight rows of a tile) only -- ONE pass for maps of up to 2048 blocks, two for
//     bigger ones (k_sort_project then also counts every block's records, k_block_prefix turns the counts into the blocks' ranges);
//     k_fuse_block  one workgroup per block: the block's records, in input order, a round of 2048 (512 for light blocks) at a time:
//                   ordered by cell in LDS (stable), then every thread runs its cell's records of the round from LDS; the cell
//                   state stays in registers.  Light rounds: arrival slots per cell + a sorting network instead of rank / scan /
//                   placement.
//
// Both walks run the per-cell recurrence in a PLAIN chain loop whenever the values allow it (every record and cell state within
// 2^-28 .. 2^28, checked outside the chain: per block where the records are placed, per pass by k_sort_project): one rarely taken
// branch per step instead of the guards of fuse_step<true> and of the increments' replay; anything else takes the guarded loops.
//
// Stability of every pass keeps ascending input order inside every cell; no float atomics.  A wave takes the time of its LONGEST
// cell chain.  Algorithmic bytes: 16 B per point (read) + 16 B per distinct touched cell (+ 8 L^2 per dense variance pass).  What
// the sort moves on top of that is stated in DESIGN.md section 4.
//
// Built with -ffp-contract=off (see gem_device.hpp).
#include "gem_kernels.hpp"
#include "gem_wave.hpp"

#include <hip/hip_ext.h>

#include <algorithm>
#include <mutex>
#include <type_traits>

namespace gem {

constexpr int kSortChunk = kSortChunkRecords;    // records per chunk of a big pass (its sorted copy is staged in LDS); small passes: kSortChunkSmall
static_assert(kSortSegsPerChunk == 4, "k_sort_project: four waves per chunk, one seg_cnt word each");
constexpr uint32_t kKeyInvalid = 0xffffffffu;   // key of a rejected / outside point in the input-ordered record arrays

// ------------------------------------------------------------------------------------------
// one point of the pass: projection + binning, the same decisions as bin_wave_body (gem_kernels.hip)
// ------------------------------------------------------------------------------------------
struct Binned { bool valid; uint32_t id; float h, v; bool colour_ok; };

// SRC: 0 = XYZI cloud, sensor model taken from the frame; 2 = XYZI cloud, every frame uses the laser model; 4 = the same and every
//      frame's rotation variance is zero (kModelLaserFast, gem_device.hpp); 1 = Fuse()'s arrays;
//      3 = XYZI cloud binned by camera pixel (the input colourisation, k_color_* below)
template <int SRC>
__device__ __forceinline__ Binned bin_one(const SortArgs& a, const FrameConst& fc, const float4& p, long long i, int orig_fallback)
{
    Binned b; b.valid = false; b.id = 0; b.h = 0.0f; b.v = 0.0f; b.colour_ok = false;
    if constexpr (SRC == 3) {                                          // input colourisation: bin by the sampled pixel
        const int pixel = camera_pixel(a.cam, p.x, p.y, p.z);
        b.valid = pixel >= 0; b.id = (uint32_t)pixel; b.h = __uint_as_float((uint32_t)i);
        return b;
    }
    int row, col; float h, v; bool colour_ok = false;
    if (SRC != 1) {
        const Projected r = SRC == 4 ? project_point<kModelLaserFast>(fc, p.x, p.y, p.z, 0)
                          : SRC == 2 ? project_point<0>(fc, p.x, p.y, p.z, 0)
                                     : project_point<-1>(fc, p.x, p.y, p.z, a.orig ? a.orig[i] : orig_fallback);
        row = r.row; col = r.col; h = r.h; v = r.var;
        if (a.rgb) {
            const uint32_t c = a.rgb[i];
            colour_ok = ((c >> 16) & 0xff) != 0 && ((c >> 8) & 0xff) != 0 && (c & 0xff) != 0 && p.w != 0.0f;
        }
    } else {
        const int cell = a.f_index[i]; h = a.f_height[i]; v = a.f_var[i];
        row = -1; col = -1;
        if (cell >= 0 && cell < fc.L * fc.L) { row = cell / fc.L; col = cell - row * fc.L; }
        if (a.f_R) colour_ok = a.f_R[i] != 0 && a.f_G[i] != 0 && a.f_B[i] != 0 && a.f_I[i] != 0.0f;
    }
    // GPU:482: "point_index[i] != map_index || points_h[i] == -1" -> the point is skipped (kept when the lowest scan
    // points are tracked: GPU:430-439 sees the point, the LOWEST walk skips its fusion)
    if (row >= fc.row0 && row < fc.row1 && (h != -1.0f || a.keep_sentinel)) {
        b.valid = true;
        const uint32_t tile = (uint32_t)((row >> 5) * a.tiles_per_row + (col >> 5));
        b.id = (tile << 10) | (uint32_t)(((row & 31) << 5) | (col & 31));
        b.h = h; b.v = v; b.colour_ok = colour_ok;
    }
    return b;
}

// SRC 4 (every frame the laser model with zero rotation variance): straight-line projection + binning (gem_device.hpp).  The
// branchy form spent per point 17 s_and_saveexec, 25 s_or and 23 v_mov on carrying values round its branches: 3.75 M SALU next to
// 9.8 M VALU instructions per C4 batch; this one takes C4 from 104.0 to 98.2 us per batch on the same box.
__device__ __forceinline__ Binned bin_one_laser_fast(const SortArgs& a, const FrameConst& fc, const float4& p, bool in_range)
{
    Binned b;
    int row, col;
    b.valid = project_bin_laser_fast(fc, p.x, p.y, p.z, in_range, a.keep_sentinel != 0, row, col, b.h, b.v);
    const uint32_t tile = (uint32_t)((row >> 5) * a.tiles_per_row + (col >> 5));
    b.id = (tile << 10) | (uint32_t)(((row & 31) << 5) | (col & 31));
    b.colour_ok = false;                                                           // (colours: the kernel looks the point's up)
    return b;
}

// chunk -> (sweep, first item of the chunk, end of the sweep): pass-1 chunks never span sweeps (the frame constants differ)
struct ChunkRange { int sweep; long long first, end; int orig0; };

template <int CH>
__device__ __forceinline__ ChunkRange chunk_range(const int* __restrict__ sweep_chunk0, const long long* __restrict__ sweep_first,
                                                  const int* __restrict__ sweep_orig0, int n_sweeps, long long n, int chunk)
{
    ChunkRange r; r.sweep = 0; r.first = (long long)chunk * CH; r.end = n; r.orig0 = 0;
    if (sweep_chunk0) {                                                // batched call; block-uniform: scalar loads
        int lo = 0, hi = n_sweeps;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (sweep_chunk0[mid] <= chunk) lo = mid; else hi = mid; }
        r.sweep = lo;
        const long long sb = sweep_first[lo];
        r.first = sb + (long long)(chunk - sweep_chunk0[lo]) * CH;
        r.end = sweep_first[lo + 1];
        r.orig0 = (sweep_orig0 ? sweep_orig0[lo] : 0) - (int)sb;       // original index of point i = i + orig0
    }
    return r;
}

// ------------------------------------------------------------------------------------------
// pass 1, project + count: the only kernel that touches the cloud
// ------------------------------------------------------------------------------------------
// (256-thread workgroups whatever the chunk size: the projection needs ~100 VGPRs, and five light workgroups per CU hide its
//  load latency better than one of 1024 threads)
// waves per SIMD the laser-only projection's register budget is set for / point loads a thread keeps in flight.  Round 6, same-box A/B
// (tools/ab_run.sh, bench_configs c4,c5): 4 waves x 8 loads (128 VGPRs) C5 299.1-299.3 us, 5 x 4: 293.7-297.0, 6 x 2 (80 VGPRs): 287.0-288.6;
// C4 98.2 / 98.5-98.8 / 97.5-98.3 -- the kernel shares its CUs with the scatter and the walk of the neighbouring passes, and what it
// gives up in loads in flight per wave it gets back in waves.
#ifndef GEM_PROJECT_WAVES
#define GEM_PROJECT_WAVES 6
#endif
template <int SRC, int CH>
__global__ __launch_bounds__(256, (SRC == 4 ? GEM_PROJECT_WAVES : (SRC == 2 ? 3 : 1))) void k_sort_project(SortArgs a)
{
#ifndef GEM_PROJECT_KB
#define GEM_PROJECT_KB 2
#endif
    constexpr int KBW = SRC == 4 ? GEM_PROJECT_KB : 8;                // (the other sources keep eight: their budgets are not the laser form's)
    constexpr int NT = 256, K = CH / NT, KB = K < KBW ? K : KBW;       // KB: point loads of a thread in flight together
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_sort[];
    uint32_t* hist = lds_sort;                                         // [bins0]
    __shared__ uint32_t btag[NT], bcnt[NT];                            // the chunk's records per block, direct-mapped by the block id's low bits (a.blk_cnt)
    const int tid = (int)threadIdx.x, chunk = (int)blockIdx.x;
    for (int i = tid; i < a.dbins[0]; i += NT) hist[i] = 0u;
    btag[tid] = 0xffffffffu; bcnt[tid] = 0u;
    if (chunk == 0 && tid == 0) *a.total = 0u;                         // k_sort_scan of this pass adds the column totals up
    // fuse_count: pass 1's scatter adds pass 2's counts up with atomics; the rows pass 2 can have (at most n / CH) start from zero
    if (a.fuse_count && (long long)chunk * CH < a.n) { uint32_t* row = a.cnt[1] + (size_t)chunk * a.dbins[1]; for (int i = tid; i < a.dbins[1]; i += NT) row[i] = 0u; }
    ChunkRange cr = chunk_range<CH>(a.sweep_chunk0, a.sweep_first, a.sweep_orig0, a.n_sweeps, a.n, chunk);
    if (!a.sweep_chunk0) cr.orig0 = a.orig0_single;                    // (a single sweep whose head another device holds)
    // BY VALUE: the stores below may alias the frame table as far as the compiler knows, and a reference would make it reload
    // every constant after every store (measured: 105 us instead of 25 for the 32 sweeps of C4)
    const FrameConst fc = a.sweep_chunk0 ? a.frames[cr.sweep] : a.frame0;
    // The records of a wave's 1024 points are stored COMPACTED at the head of the wave's segment of the chunk, in input order: a
    // point outside the map (or rejected) leaves nothing behind -- on a LiDAR batch that is three points in ten, which pass 1's
    // scatter then neither reads nor ranks.  seg_cnt[chunk][wave] says how many there are.
    // The chunk's arrays through UNIFORM bases (the chunk's first point: scalar registers) and 32-bit byte offsets per lane -- the
    // loads' and stores' scalar-base form; 64-bit indices cost six VALU instructions per point on address arithmetic alone.
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(1))) char* gbytes_t;
    typedef const __attribute__((address_space(1))) char* gcbytes_t;
    typedef const __attribute__((address_space(1))) float4* gf4_t;
    typedef __attribute__((address_space(1))) uint32_t* gu32_t;
    typedef __attribute__((address_space(1))) uint2* gu2_t;
#else
    typedef char* gbytes_t; typedef const char* gcbytes_t; typedef const float4* gf4_t; typedef uint32_t* gu32_t; typedef uint2* gu2_t;
#endif
    const gcbytes_t xyzi_c = (gcbytes_t)(a.xyzi + cr.first);
    const gbytes_t key_c = (gbytes_t)(a.key_a + cr.first), hv_c = (gbytes_t)(a.hv_a + cr.first), src_c = (gbytes_t)(a.src_a ? a.src_a + cr.first : nullptr);
    const uint32_t n_here = (uint32_t)(cr.end - cr.first < (long long)CH ? (cr.end > cr.first ? cr.end - cr.first : 0) : CH);   // points of this chunk
    const uint32_t j_lane = (uint32_t)(tid >> 6) * (uint32_t)(K * 64) + (uint32_t)(tid & 63), j_seg = (uint32_t)(tid >> 6) * (uint32_t)(K * 64);
    const uint64_t lt = lanemask_lt();
    uint32_t kept = 0;                                                 // wave-uniform
    bool odd = false;                                                  // a record outside the plain range of the walks' chain loops (k_fuse_walk)
    const uint32_t sweep_bits = (uint32_t)(cr.sweep + a.sweep_id0) << a.id_bits;   // (a shard of a multi-GPU batch numbers its sweeps globally)
    const uint32_t d0mask = (1u << a.dbits[0]) - 1u, d0shift = (uint32_t)a.dshift[0];
    __syncthreads();
    // blocks of eight points: the loads of a block are in flight together, then each point is projected, stored and counted
    for (int k0 = 0; k0 < K; k0 += KB) {
        float4 p[KB];
        if (SRC != 1) {
#pragma unroll
            for (int k = 0; k < KB; ++k) { const uint32_t j = j_lane + (uint32_t)((k0 + k) * 64); p[k] = *(gf4_t)(xyzi_c + (uint32_t)((j < n_here ? j : 0u) * 16u)); }
        }
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const uint32_t j = j_lane + (uint32_t)((k0 + k) * 64);
            const bool in_chunk = j < n_here;
            const long long i = cr.first + (long long)j;               // (only the colour / camera-model paths look at it)
            Binned b; b.valid = false;
            if constexpr (SRC == 4) {
                b = bin_one_laser_fast(a, fc, p[k], in_chunk);
                if (a.rgb) {                                           // (frame-uniform)
                    const uint32_t c = a.rgb[in_chunk ? i : cr.first];
                    b.colour_ok = ((c >> 16) & 0xff) != 0 && ((c >> 8) & 0xff) != 0 && (c & 0xff) != 0 && p[k].w != 0.0f;
                }
            } else if (in_chunk) b = bin_one<SRC>(a, fc, SRC != 1 ? p[k] : make_float4(0.f, 0.f, 0.f, 0.f), i, (int)i + cr.orig0);
            const uint64_t m = __ballot(b.valid);
            const uint32_t bin = (b.id >> d0shift) & d0mask;
            if (b.valid) {
                if constexpr (SRC != 3) odd = odd | !(fabsf(b.h) <= 268435456.0f) | !(b.v >= 3.7252902984619140625e-9f) | !(b.v <= 268435456.0f);
                const uint32_t at = j_seg + kept + (uint32_t)__popcll(m & lt);       // slot inside the chunk
                *(gu32_t)(key_c + (uint32_t)(at * 4u)) = b.id | sweep_bits;
                *(gu2_t)(hv_c + (uint32_t)(at * 8u)) = make_uint2(__float_as_uint(b.h), __float_as_uint(b.v));
                if (a.src_a) *(gu32_t)(src_c + (uint32_t)(at * 4u)) = (uint32_t)i | (b.colour_ok ? 0x80000000u : 0u);  // source point; bit 31: R, G, B, intensity all non-zero
            }
            // Histogram.  With a coarse digit (the blocks of the block-sorted form) consecutive points of a scan share their bin, and
            // 64 lanes adding 1 to one LDS word take 64 turns: a lane adds for its whole RUN of equal neighbours instead (the run's
            // first lane, found with one DPP shift and one ballot; a bin that comes back later in the wave simply adds twice).
            // The runs are runs of one BLOCK (d0shift >= 8: the digit is part of the block id): the same lane also adds its run to the
            // block's record count in HBM (a.blk_cnt, when the last pass's bins are not the blocks: launch_block_prefix turns the
            // counts into every block's range in the sorted records, instead of a pass over the sorted keys).
            if (d0shift != 0u) {                                       // block-uniform
                const uint32_t kb = b.valid ? (b.id >> 8) : 0xffffffffu;
                const uint32_t pv = wave_prev(kb);
                const bool head = (tid & 63) == 0 || kb != pv;
                const uint64_t heads = __ballot(head);
                if (head && b.valid) {
                    const uint64_t later = (tid & 63) == 63 ? 0ull : heads >> ((tid & 63) + 1);
                    const uint32_t run = later ? (uint32_t)__ffsll((unsigned long long)later) : 64u - (uint32_t)(tid & 63);
                    atomicAdd(&hist[bin], run);
                    if (a.blk_cnt) {
                        // (through the LDS first: a chunk's 4096 points make hundreds of runs in a few dozen blocks -- C5: 2.5 M atomics
                        //  to HBM cost the kernel 11 us of its 55)
                        const uint32_t slot = kb & (uint32_t)(NT - 1);
                        const uint32_t old = atomicCAS(&btag[slot], 0xffffffffu, kb);
                        if (old == 0xffffffffu || old == kb) atomicAdd(&bcnt[slot], run);
                        else __hip_atomic_fetch_add(a.blk_cnt + kb, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else if (b.valid) atomicAdd(&hist[bin], 1u);
            kept += (uint32_t)__popcll(m);
        }
    }
    if ((tid & 63) == 0) a.seg_cnt[(size_t)chunk * kSortSegsPerChunk + (tid >> 6)] = kept;
    if (odd && a.odd_flag) *a.odd_flag = a.epoch;                      // (every writer stores the same value)
    __syncthreads();
    for (int i = tid; i < a.dbins[0]; i += NT) a.cnt[0][(size_t)chunk * a.dbins[0] + i] = hist[i];
    if (a.blk_cnt && btag[tid] != 0xffffffffu) __hip_atomic_fetch_add(a.blk_cnt + btag[tid], bcnt[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------
// column scan over the chunks (both passes), in kScanSegs SEGMENTS of the chunk range so that enough workgroups take part:
// workgroup (x, seg) owns 64 bins x the seg-th quarter of the chunks, wave w the w-th contiguous share of those (lane = bin:
// 256-byte coalesced rows).  cnt[c][b] -> records of bin b in the earlier chunks OF THE SAME SEGMENT, in place;
// segtot[seg][b] = the segment's column sum; *total_out += everything.  The consumers add the (at most three) earlier
// segments' sums themselves.
// (Tried and dropped: one workgroup per 64 bins over all chunks -- 8 workgroups for 512 bins, 15 us once the per-wave share
//  spilled registers; a "last workgroup scans the totals" epilogue -- its device-scope fences write the L2s back on this
//  multi-XCD part and tripled the kernel's time.)
// ------------------------------------------------------------------------------------------
constexpr int kScanSegs = 4;

__device__ __forceinline__ int scan_seg_chunks(int nc) { return (nc + kScanSegs - 1) / kScanSegs; }     // chunks per segment

__global__ __launch_bounds__(1024) void k_sort_scan(uint32_t* __restrict__ cnt, uint32_t* __restrict__ segtot, int bins, int n_chunks,
                                                    const uint32_t* __restrict__ records, uint32_t* __restrict__ total_out, int chunk_records)
{
    __shared__ uint32_t part[16][64];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6), seg = (int)blockIdx.y;
    const int b = (int)blockIdx.x * 64 + lane;
    // pass 2: the number of chunks depends on how many records pass 1 kept (known on the device only)
    const int nc = records ? (int)(((unsigned long long)*records + (unsigned)chunk_records - 1u) / (unsigned)chunk_records) : n_chunks;
    const int Q = scan_seg_chunks(nc);
    const int s_lo = min(nc, seg * Q), s_hi = min(nc, s_lo + Q);
    const int S = (Q + 15) >> 4;
    const int c_lo = min(s_hi, s_lo + w * S), c_hi = min(s_hi, c_lo + S);
    const bool on = b < bins;
    uint32_t* col = cnt + (on ? b : 0);
    const int c_last = c_hi > c_lo ? c_hi - 1 : 0;
    uint32_t sum = 0;
    uint32_t keep[16];
    if (S <= 16) {                                                     // block-uniform: the share stays in registers
#pragma unroll
        for (int j = 0; j < 16; ++j) keep[j] = col[(size_t)min(c_lo + j, c_last) * bins];
#pragma unroll
        for (int j = 0; j < 16; ++j) { keep[j] = c_lo + j < c_hi ? keep[j] : 0u; sum += keep[j]; }
    } else {
        for (int c0 = c_lo; c0 < c_hi; c0 += 16) {                     // sixteen independent loads in flight (clamped rows)
            uint32_t v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = col[(size_t)min(c0 + j, c_last) * bins];
#pragma unroll
            for (int j = 0; j < 16; ++j) sum += c0 + j < c_hi ? v[j] : 0u;
        }
    }
    part[w][lane] = on ? sum : 0u;
    __syncthreads();
    uint32_t run = 0, all = 0;
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) { const uint32_t v = part[ww][lane]; run += ww < w ? v : 0u; all += v; }
    if (S <= 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (on && c_lo + j < c_hi) { col[(size_t)(c_lo + j) * bins] = run; run += keep[j]; }
        }
    } else {
        for (int c0 = c_lo; c0 < c_hi; c0 += 16) {
            uint32_t v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = col[(size_t)min(c0 + j, c_last) * bins];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (on && c0 + j < c_hi) { col[(size_t)(c0 + j) * bins] = run; run += v[j]; }
            }
        }
    }
    if (on && w == 0) segtot[(size_t)seg * bins + b] = all;
    if (total_out && w == 0) {
        const uint32_t s = wave_inclusive_scan(all);
        if (lane == 63 && s) atomicAdd(total_out, s);
    }
}

// Stable rank inside a wave's share of a chunk (k_sort_scatter, step 1).  The wave's cursor of a bin (LDS, private to the wave: a
// wave's LDS operations execute in order) carries the count from step to step; inside a step the lanes of a bin have to find
// each other.  Two ways:
//   * THROUGH THE LDS (digits of scattered values: the low digits of the cell-sorted form): every lane stores its number into
//     the wave's slot of its bin (whichever store lands last names the group), reads the name back, ORs its bit into the wave's
//     64-bit mask of that name and reads the mask.  Five LDS instructions and a handful of VALU ones for any digit width --
//     matching the digits by one ballot per bit (wave_peers) costs six VALU instructions per bit, and this kernel is bound by
//     VALU issue.
//   * BY BALLOT (COHERENT: the coarse digits of the block-sorted form, where the 64 consecutive records of a wave instruction
//     fall into a handful of bins and 64 lanes hitting one LDS word would take 64 turns): one ballot per distinct bin, taken
//     from the first lane still unmatched; a step with more than kFewBins distinct bins falls back to the LDS way.
constexpr int kFewBins = 4;       // (C5's first pass: 8 ballots 68 us, 4 62, 16 84 -- past a few keys the LDS way is the cheaper one)

// words of LDS the ranking phase of k_sort_scatter needs: per wave a cursor per bin, a name byte per bin (+ 64), 64 masks
__host__ __device__ constexpr size_t slot_words(int bins) { return (size_t)((bins + 3) / 4) + 16; }   // name bytes of the bins + 64 spare ones (lanes without a record)
__host__ __device__ constexpr size_t rank_words(int nw, int bins) { return (size_t)nw * bins + (size_t)nw * slot_words(bins) + (size_t)nw * 128; }

// ------------------------------------------------------------------------------------------
// count + scatter of one digit (both passes), chunks of kSortChunk records.  Pass 1 reads the input-ordered records of
// k_sort_project (sweep-aligned chunks, rejected points carry an invalid key); pass 2 reads the output of pass 1.
// ------------------------------------------------------------------------------------------
template <int NT, int CH>
__global__ __launch_bounds__(NT) void k_sort_count(PassArgs a)
{
    constexpr int K = CH / NT;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_sort[];
    uint32_t* hist = lds_sort;                                         // [bins]
    const int tid = (int)threadIdx.x, chunk = (int)blockIdx.x;
    const uint32_t M = *a.n_dev;
    if ((unsigned long long)chunk * CH >= M) return;                   // k_sort_scan only reads the rows of live chunks
    for (int i = tid; i < a.bins; i += NT) hist[i] = 0u;
    const uint32_t base = (uint32_t)chunk * CH + (uint32_t)(tid >> 6) * (K * 64) + (uint32_t)(tid & 63);
    uint32_t key[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { const uint32_t i = base + k * 64; key[k] = a.key_in[i < M ? i : M - 1u]; }
    __syncthreads();
    // (a lane adds for its whole run of equal neighbours, as in k_sort_project: the higher digit of a block id is the same for
    //  hundreds of consecutive records of the pass before, and 64 lanes adding 1 to one LDS word take 64 turns -- C5: 23 -> 8 us)
    if (a.coherent) {                                                  // block-uniform
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const bool on = base + k * 64 < M;
            const uint32_t bin = (key[k] >> a.shift) & a.mask;
            const uint32_t kb = on ? bin : 0xffffffffu;
            const uint32_t pv = wave_prev(kb);                         // (every lane takes part: a DPP read of a lane that sits out is undefined)
            const bool head = (tid & 63) == 0 || kb != pv;
            const uint64_t heads = __ballot(head);
            if (head && on) {
                const uint64_t later = (tid & 63) == 63 ? 0ull : heads >> ((tid & 63) + 1);
                atomicAdd(&hist[bin], later ? (uint32_t)__ffsll((unsigned long long)later) : 64u - (uint32_t)(tid & 63));
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) if (base + k * 64 < M) atomicAdd(&hist[(key[k] >> a.shift) & a.mask], 1u);
    }
    __syncthreads();
    for (int i = tid; i < a.bins; i += NT) a.cnt[(size_t)chunk * a.bins + i] = hist[i];
}

// The chunk's records are ranked (stable), put into LDS in their sorted order and written out from there: consecutive threads
// then write consecutive records of a bin -- runs of several records, 64-byte segments -- instead of 64 lanes writing 64
// scattered 8-byte pieces (which cost 55 us for the 2.9 M records of C4 against 24 from LDS).
template <int NT, bool ATTR, bool COHERENT, int CH>
__global__ __launch_bounds__(NT, (NT == 512 ? 4 : 2)) void k_sort_scatter(PassArgs a)
{
    constexpr int NW = NT / 64, K = CH / NT, SEG = CH / kSortSegsPerChunk;     // SEG: slots of a k_sort_project wave's segment
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_sort[];
    const int bins = a.bins;
    uint32_t* lbase = lds_sort;                                        // [bins] first LOCAL (chunk-sorted) position of every bin
    uint32_t* delta = lbase + bins;                                    // [bins] global position - local position
    uint32_t* scratch = delta + bins;                                  // [16]
    uint32_t* region = scratch + 16;                                   // per-wave cursors, then (aliased) the staged records
    uint32_t* wcnt = region;                                           // [NW][bins]
    uint8_t* wslots = reinterpret_cast<uint8_t*>(region + NW * bins);   // [NW][4 * slot_words(bins)] bytes
    unsigned long long* wpms = reinterpret_cast<unsigned long long*>(region + NW * bins + NW * slot_words(bins));   // [NW][64]
    uint2* st_hv = reinterpret_cast<uint2*>(region);                   // [CH]
    uint32_t* st_key = region + 2 * CH;                                // [CH]
    uint32_t* st_src = region + 3 * CH;                                // [CH] (ATTR)
    const int tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6, chunk = (int)blockIdx.x;
    const uint64_t lt = lanemask_lt();
    long long first, end;
    if (a.n_dev) { first = (long long)chunk * CH; end = (long long)*a.n_dev; }
    else { const ChunkRange cr = chunk_range<CH>(a.sweep_chunk0, a.sweep_first, nullptr, a.n_sweeps, a.n_host, chunk); first = cr.first; end = cr.end; }
    if (first >= end && !(a.bin_base && chunk == 0)) return;           // workgroup 0 of the last pass always publishes the bin bases
    for (int i = tid; i < NW * bins; i += NT) wcnt[i] = 0u;
    for (int i = tid; i < NW * 64; i += NT) wpms[i] = 0ull;
    // pass 1: the wave's 512 or 1024 positions lie in ONE 1024-slot segment of the chunk, whose first seg_cnt records are there
    // (k_sort_project); the later passes read dense arrays
    uint32_t seg_off = 0, seg_n = 0xffffffffu;
    if (a.seg_cnt) { const uint32_t pos = (uint32_t)(w * (K * 64)); seg_off = pos % (uint32_t)SEG; seg_n = a.seg_cnt[(size_t)chunk * kSortSegsPerChunk + pos / (uint32_t)SEG]; }
    uint2 hv[K]; uint32_t key[K], src[K], rk[K];
    {
        // the chunk's records through uniform bases (its first record: scalar registers) and 32-bit byte offsets per lane, see k_sort_project
#if defined(__HIP_DEVICE_COMPILE__)
        typedef const __attribute__((address_space(1))) char* gcbytes_t;
        typedef const __attribute__((address_space(1))) uint32_t* gcu32_t;
        typedef const __attribute__((address_space(1))) uint2* gcu2_t;
#else
        typedef const char* gcbytes_t; typedef const uint32_t* gcu32_t; typedef const uint2* gcu2_t;
#endif
        const gcbytes_t key_c = (gcbytes_t)(a.key_in + first), hv_c = (gcbytes_t)(a.hv_in + first), src_c = (gcbytes_t)(ATTR ? a.src_in + first : nullptr);
        const uint32_t n_here = (uint32_t)(end - first < (long long)CH ? (end > first ? end - first : 0) : CH);
        const uint32_t j0 = (uint32_t)(w * (K * 64) + lane);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t j = j0 + (uint32_t)(k * 64);
            const bool there = j < n_here && seg_off + (uint32_t)(k * 64 + lane) < seg_n;
            const uint32_t jc = there ? j : 0u;
            key[k] = *(gcu32_t)(key_c + (uint32_t)(jc * 4u)); hv[k] = *(gcu2_t)(hv_c + (uint32_t)(jc * 8u));
            if (ATTR) src[k] = *(gcu32_t)(src_c + (uint32_t)(jc * 4u));
            if (!there) key[k] = kKeyInvalid;
        }
    }
    __syncthreads();
    // ---- 1. stable rank inside the wave's share, per-wave counts
    {
        uint32_t* wcur = wcnt + w * bins;
        uint8_t* wslot = wslots + (size_t)w * 4 * slot_words(bins);
        const uint32_t spare = 4u * (uint32_t)((bins + 3) / 4) + (uint32_t)lane;     // the slot of a lane without a record: its own
        unsigned long long* wpm = wpms + w * 64;
        const unsigned long long mybit = 1ull << lane;
        // The K steps of wave_rank_step (see there), phase by phase instead of step by step: a wave's LDS operations execute in
        // order, so step k + 1 may store its names behind step k's read-back without waiting for it -- three LDS round trips per
        // chunk instead of three per step.
        uint32_t bin[K], name[K]; uint64_t peers[K]; bool valid[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { valid[k] = key[k] != kKeyInvalid; bin[k] = (key[k] >> a.shift) & a.mask; }
        if constexpr (COHERENT) {
            // (tried: the steps the ballots do not settle going through the LDS together afterwards, phase by phase like the other way
            //  below -- the flags and names of eight steps cost more registers than the overlapped round trips save: C5's first pass
            //  62.9 -> 67.4 us)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                uint64_t rem = __ballot(valid[k]);
                peers[k] = 0ull;
                // (few_bins < 0: one ballot per digit BIT instead -- no loop over the distinct bins, no LDS; the first of several passes
                //  sees the records in input order, a dozen blocks in 64 lanes at long range: C5 62.8 -> 55.6 us; later passes see one
                //  or two bins per instruction and keep the loop: 43.9 against 46.1; C4's single 11-bit pass is even)
                if (a.few_bins < 0) { peers[k] = wave_peers(valid[k], bin[k], a.digit_bits); rem = 0ull; }
#pragma unroll 1
                for (int it = 0; it < a.few_bins && rem != 0; ++it) {  // wave-uniform
                    const uint32_t kb = (uint32_t)__builtin_amdgcn_readlane((int)bin[k], __ffsll((unsigned long long)rem) - 1);
                    const bool mine = valid[k] && bin[k] == kb;
                    const uint64_t same = __ballot(mine);
                    peers[k] = mine ? same : peers[k];
                    rem &= ~same;
                }
                if (rem != 0) {                                        // wave-uniform: many bins in this step, through the LDS
                    uint8_t* slot = wslot + (valid[k] ? bin[k] : spare);
                    __hip_atomic_store(slot, (uint8_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t nm = (uint32_t)__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (valid[k]) atomicOr(&wpm[nm], mybit);
                    peers[k] = valid[k] ? (uint64_t)__hip_atomic_load(&wpm[nm], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
                    if (valid[k] && nm == (uint32_t)lane) __hip_atomic_store(&wpm[nm], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                uint8_t* slot = wslot + (valid[k] ? bin[k] : spare);   // no branch: the eight round trips overlap
                __hip_atomic_store(slot, (uint8_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                name[k] = (uint32_t)__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (valid[k]) atomicOr(&wpm[name[k]], mybit);
                peers[k] = valid[k] ? (uint64_t)__hip_atomic_load(&wpm[name[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
                if (valid[k] && name[k] == (uint32_t)lane) __hip_atomic_store(&wpm[name[k]], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t rank = (uint32_t)__popcll(peers[k] & lt);
            uint32_t old = 0;
            if (valid[k] && rank == 0) old = atomicAdd(&wcur[bin[k]], (uint32_t)__popcll(peers[k]));
            old = (uint32_t)__shfl((int)old, valid[k] ? __ffsll((unsigned long long)peers[k]) - 1 : lane, 64);
            rk[k] = old + rank;
        }
    }
    __syncthreads();
    // ---- 2. per bin: the waves in order (exclusive prefix), the chunk's local base, the way from local to global positions
    {
        const int per = (bins + NT - 1) / NT, b0 = tid * per;
        uint32_t* tbv = reinterpret_cast<uint32_t*>(wslots);          // [bins] records of the bin in the whole pass (the name slots are free now)
        // the bin's records in the whole pass / in the chunk segments before this chunk's (k_sort_scan)
        const int nc = a.n_dev ? (int)((end + CH - 1) / CH) : a.n_chunks;
        const int seg = chunk / max(1, scan_seg_chunks(nc));                 // (nc = 0: pass 1 kept nothing, workgroup 0 only publishes the bases)
        uint32_t sum = 0, gsum = 0;
        for (int j = 0; j < per; ++j) {
            const int b = b0 + j;
            if (b < bins) {
                uint32_t sv[kScanSegs];
#pragma unroll
                for (int sg = 0; sg < kScanSegs; ++sg) sv[sg] = a.segtot[(size_t)sg * bins + b];
                const uint32_t earlier = first < end ? a.cnt[(size_t)chunk * bins + b] : 0u;   // (an empty pass has no count rows)
                uint32_t run = 0;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) { const uint32_t c = wcnt[ww * bins + b]; wcnt[ww * bins + b] = run; run += c; }
                lbase[b] = run;                                        // the bin's records in this chunk, for now
                sum += run;
                uint32_t tb = 0, before = 0;
#pragma unroll
                for (int sg = 0; sg < kScanSegs; ++sg) { before += sg < seg ? sv[sg] : 0u; tb += sv[sg]; }
                tbv[b] = tb; delta[b] = before + earlier;              // earlier segments + earlier chunks of this segment
                gsum += tb;
            }
        }
        uint32_t chunk_records, all;
        uint32_t ex = block_exclusive_scan<NT>(sum, scratch, &chunk_records);
        uint32_t gex = block_exclusive_scan<NT>(gsum, scratch, &all);  // first record of the bin in the pass's output
        for (int j = 0; j < per; ++j) {
            const int b = b0 + j;
            if (b < bins) {
                const uint32_t c = lbase[b];
                lbase[b] = ex;
                delta[b] = gex + delta[b] - ex;                        // bin base + earlier records of the bin - local position
                if (a.bin_base && chunk == 0) a.bin_base[b] = gex;
                ex += c; gex += tbv[b];
            }
        }
        if (tid == 0) {
            scratch[15] = chunk_records;
            if (a.bin_base && chunk == 0) { a.bin_base[bins] = all; if (a.counters) atomicAdd(&a.counters[0], (unsigned long long)all); }
        }
    }
    __syncthreads();
    // ---- 3. local position of every record (the cursors are read for the last time)
    {
        const uint32_t* wcur = wcnt + w * bins;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t bin = (key[k] >> a.shift) & a.mask;
            rk[k] = key[k] != kKeyInvalid ? lbase[bin] + wcur[bin] + rk[k] : 0xffffffffu;
        }
    }
    const uint32_t chunk_records = scratch[15];
    __syncthreads();
    // ---- 4. the records in chunk-sorted order in LDS (over the cursors)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (rk[k] != 0xffffffffu) {
            st_hv[rk[k]] = hv[k]; st_key[rk[k]] = key[k];
            if (ATTR) st_src[rk[k]] = src[k];
        }
    }
    __syncthreads();
    // ---- 5. out, consecutive threads = consecutive records of a bin
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t j = (uint32_t)(tid + k * NT);
        const bool on = j < chunk_records;
        uint32_t kk = 0u, pos = 0u;
        if (on) {
            kk = st_key[j];
            pos = j + delta[(kk >> a.shift) & a.mask];
            a.hv_out[pos] = st_hv[j];
            a.key_out[pos] = kk;
            if (ATTR) a.src_out[pos] = st_src[j];
        }
        // The NEXT pass's count on the way out (small passes: one launch and one pass over the keys less): the record's place in
        // the output is its chunk there.  A lane adds for its whole run of neighbours with the same (chunk, digit) -- the records
        // of a dense tile follow each other here, and a thousand atomics on one word would take a thousand turns at the L2.
        if (a.next_cnt) {                                              // block-uniform
            const uint32_t tag = on ? (pos / (uint32_t)CH) * (uint32_t)a.next_bins + ((kk >> a.next_shift) & a.next_mask) : 0xffffffffu;
            const uint32_t pv = wave_prev(tag);                        // (every lane takes part, see k_sort_count)
            const bool head = lane == 0 || tag != pv;
            const uint64_t heads = __ballot(head);
            if (head && on) {
                const uint64_t later = lane == 63 ? 0ull : heads >> (lane + 1);
                const uint32_t run = later ? (uint32_t)__ffsll((unsigned long long)later) : 64u - (uint32_t)lane;
                __hip_atomic_fetch_add(a.next_cnt + tag, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The sweeps' variance increments between the records of a cell (both walks).  Between two records of sweeps s < t the cell lives
// through t - s times {the floor that ends a Fuse (GPU:533-534); the next sweep's Mapvar_update (GPU:540-547)} -- a serial chain
// of rounded additions, so it is replayed step by step.  The lanes of a wave stand at unrelated sweeps and the increments come
// from LDS: a loop that fetches one increment per round exposes an LDS round trip per sweep to every lane of the wave (that
// loop was 60 % of the walks' time on a batch of 32 sweeps).  Here every lane keeps the increments of its NEXT FOUR sweeps in
// registers, refilled behind the record's fusion step: gaps of up to four sweeps -- nearly all of them -- cost four VALU
// instructions per sweep and no wait.  vu[] is padded with 8 words.
// ------------------------------------------------------------------------------------------
struct SweepReplay {
    float w0, w1, w2, w3;          // increments of sweeps cur + 1 .. cur + 4
    uint32_t cur;                  // "inside sweep cur, its increment applied"

    __device__ __forceinline__ void refill(const float* vu) { w0 = vu[cur + 1u]; w1 = vu[cur + 2u]; w2 = vu[cur + 3u]; w3 = vu[cur + 4u]; }
    __device__ __forceinline__ void one(float& cs, float u, float var_floor)
    {
        if (cs < var_floor) cs = var_floor;
        ++cur;
        if (cs != kInitVariance) cs += u;
    }
    // this lane: to sweep `to` (>= cur; lanes that stay pass cur)
    __device__ __forceinline__ void advance(float& cs, uint32_t to, const float* vu, float var_floor)
    {
        if (__ballot(cur < to) == 0) return;                           // wave-uniform
        if (cur < to) one(cs, w0, var_floor);
        if (__ballot(cur < to) != 0) {
            if (cur < to) one(cs, w1, var_floor);
            if (__ballot(cur < to) != 0) {
                if (cur < to) one(cs, w2, var_floor);
                if (__ballot(cur < to) != 0) {
                    if (cur < to) one(cs, w3, var_floor);
                    while (__ballot(cur < to) != 0) { if (cur < to) one(cs, vu[cur + 1u], var_floor); }   // a gap of five or more sweeps
                }
            }
        }
        refill(vu);
    }
    // every lane to the last sweep: the sweep index runs wave-uniform from the wave's earliest lane, four increments per round trip
    __device__ __forceinline__ void finish(float& cs, uint32_t last_sw, const float* vu, float var_floor)
    {
        const uint32_t behind = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(0xffffu - cur), 63);
        for (uint32_t s = 0xffffu - behind + 1u; s <= last_sw; s += 4u) {   // wave-uniform
            const float u0 = vu[s], u1 = vu[s + 1u], u2 = vu[s + 2u], u3 = vu[s + 3u];
            if (cur < s && s <= last_sw) one(cs, u0, var_floor);
            if (cur < s + 1u && s + 1u <= last_sw) one(cs, u1, var_floor);
            if (cur < s + 2u && s + 2u <= last_sw) one(cs, u2, var_floor);
            if (cur < s + 3u && s + 3u <= last_sw) one(cs, u3, var_floor);
        }
    }
    // ... when the pass's floor is positive and its increments are not negative (WalkArgs::plain_env): after a lane's first floor the
    // later ones change nothing and the variance is never the -10 of an empty cell, so a lane's way to the last sweep is ONE floor and
    // the rounded additions of its sweeps' increments, in order -- three instructions per sweep of the wave's earliest lane instead
    // of the guarded steps' dozen (every cell of the map lives through every sweep: a batch of 32 sweeps spent an eighth of
    // k_fuse_block's VALU instructions here)
    __device__ __forceinline__ void finish_plain(float& cs, uint32_t last_sw, const float* vu, float var_floor)
    {
        const uint32_t behind = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(0xffffu - cur), 63);
        const uint32_t s0 = 0xffffu - behind + 1u;                      // the earliest lane's next sweep (wave-uniform)
        if (s0 > last_sw) return;
        const float fl = cs < var_floor ? var_floor : cs;
        cs = cur < last_sw ? fl : cs;
        for (uint32_t s = s0; s <= last_sw; ++s) { const float c1 = cs + vu[s]; cs = cur < s ? c1 : cs; }
        cur = cur < last_sw ? last_sw : cur;
    }
};

// ------------------------------------------------------------------------------------------
// k_fuse_walk : one wave per 64 consecutive cells (two rows of a 32x32 tile), one lane per cell
// ------------------------------------------------------------------------------------------
// FLAGS: bits 0-1 = ATTR (0 none, 1 colours from the cloud, 2 colours from gem_fuse's arrays), bit 2 = LOWEST (also maintain
// map_lowest, GPU:432-439).  MODE: bit 0 = variance increments between the sweeps (batched call with var_updates), bit 1 =
// count the touched cells per sweep (statistics).
constexpr int kWalkMaxSweeps = 512;

constexpr int kWalkDepth = 3;                   // groups of four records a lane has in flight
constexpr int kWalkNT = 256;                    // threads of a k_fuse_walk workgroup: 256 consecutive cells (eight rows of a tile)

template <int FLAGS, int MODE>
__global__ __launch_bounds__(kWalkNT) void k_fuse_walk(WalkArgs a)
{
    constexpr int ATTR = FLAGS & 3;
    constexpr bool LOWEST = (FLAGS & 4) != 0;
    constexpr bool HAS_VU = (MODE & 1) != 0, COUNT_SWEEPS = (MODE & 2) != 0, KEYED = HAS_VU || COUNT_SWEEPS;
    constexpr int NT = kWalkNT, NW = NT / 64;
    __shared__ uint32_t cstart[NT], cend[NT], hist[128];
    __shared__ float sh_e[NT], sh_s[NT], sh_l[LOWEST ? NT : 1];
    __shared__ uint16_t perm[NT];
    __shared__ float vu[HAS_VU ? kWalkMaxSweeps + 8 : 1];
    const int tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6;
    // Block -> 256 cells (a quarter of a tile), CENTRE ROWS FIRST: the map is robot-centric, the cells under the sensor carry
    // chains a hundred times longer than the rim's, and a wave takes as long as its longest chain -- so the tile rows start in
    // the order c, c-1, c+1, c-2, ... from the row holding the map centre in storage coordinates (wrapped: a bijection); inside a
    // row the tiles keep their memory order (C4's walk alone 62 -> 52 us; ordering the columns centre-first too cost an
    // aggregated cloud, whose records then are not read front to back, 115 -> 120 us).
    // The four waves of a workgroup land on the CU's four SIMDs: a long chain still has a SIMD to itself, and the search and the
    // boundary scan below are shared by four times as many cells as with one wave per workgroup.
    int tile, q4 = (int)(blockIdx.x & 3);
    {
        const int tpr = a.tiles_per_row, rnk = (int)(blockIdx.x >> 2);
        if (a.walk_order) {
            const int bi = rnk / tpr, bj = rnk - bi * tpr;
            const int oi = (bi & 1) ? -((bi + 1) >> 1) : (bi >> 1);
            int r = a.center_tr + oi; r = r < 0 ? r + tpr : (r >= tpr ? r - tpr : r);
            tile = r * tpr + bj;
        } else tile = rnk;
    }
    const int tr = tile / a.tiles_per_row, tc = tile - tr * a.tiles_per_row;
    if ((tr << 5) >= a.row1 || (tr << 5) + 32 <= a.row0) return;       // a tile row outside this device's strip
    const uint32_t idmask = (1u << a.id_bits) - 1u;
    const uint32_t id0 = ((uint32_t)tile << 10) | ((uint32_t)q4 << 8); // the cell ids of this workgroup: id0 .. id0 + 255
    // the records of these cells lie inside the run of the last pass's bin that holds id0 (256 divides the bin width)
    const uint32_t bin0 = id0 >> a.bin_shift;
    const uint32_t run_lo = a.bin_base[bin0], run_hi = a.bin_base[bin0 + 1];
    if (run_lo == run_hi && !a.dense) return;

    // ---- the thread's OWN cell (cell tid of the workgroup): its map values are fetched now, coalesced, in flight behind the search
    const int L = a.L;
    const int row_t = (tr << 5) + (q4 << 3) + (tid >> 5), col_t = (tc << 5) + (tid & 31);
    const bool owned_t = row_t >= a.row0 && row_t < a.row1 && col_t < L;
    const size_t g_t = owned_t ? (size_t)row_t * L + col_t : 0;
    const float e_t = a.elevation[g_t], s_t = a.variance[g_t];
    float l_t = 0.0f;
    if constexpr (LOWEST) {                                            // map_lowest is indexed by the GEOGRAPHIC cell (GPU:430)
        int gr = row_t - a.start0, gc = col_t - a.start1;
        gr += gr < 0 ? L : 0; gc += gc < 0 ? L : 0;
        l_t = a.lowest[owned_t ? (size_t)gr * L + gc : 0];
    }
    if constexpr (HAS_VU) for (int i = tid; i < a.n_sweeps + 8; i += NT) vu[i] = i < a.n_sweeps ? a.var_updates[i] : 0.0f;   // (nothing reads past sweep n_sweeps + 3)

    // ---- where the cells of this workgroup begin and end in one SOURCE of records sorted by cell id (keys[lo0, hi0) may hold
    //      them): cstart / cend in LDS; false if the source has nothing for these cells.  Block-uniform.
    uint32_t rb = 0;
    auto find_bounds = [&](const uint32_t* __restrict__ keys, uint32_t lo0, uint32_t hi0) -> bool {
        // 32-ary search, both ends at once: lanes 0-31 look for the first record with id >= id0, lanes 32-63 for id >= id0 + 256
        // (every wave of the workgroup runs the same search: the probes of the other three hit the L1)
        uint32_t lo = lo0, hi = hi0;                                   // ids below `lo` are < target, ids from `hi` on are >= target
        {
            const uint32_t target = id0 + (uint32_t)(lane >> 5) * (uint32_t)NT, l5 = (uint32_t)lane & 31u;
            while (__ballot(lo < hi) != 0) {                           // wave-uniform
                const uint32_t n = hi - lo, s = (n + 32u) / 33u;       // probes lo + j s + s - 1, j = 0..31
                const uint32_t pos = lo + l5 * s + s - 1u;
                const bool probe = lo < hi && pos < hi;
                const uint32_t id = probe ? (keys[pos] & idmask) : 0xffffffffu;
                const uint64_t bl = __ballot(probe && id < target);
                const uint32_t k = (uint32_t)__popc((uint32_t)(bl >> (lane & 32)));   // a prefix of the probes: the ids are sorted
                // probes 0 .. k-1 are below the target, probe k (if there is one: k < 32 and inside the range) is not
                if (lo < hi) { const uint32_t nl = lo + k * s; if (k < 32u) hi = min(hi, nl + s - 1u); lo = nl; }
            }
        }
        rb = (uint32_t)__shfl((int)lo, 0, 64);
        const uint32_t re = (uint32_t)__shfl((int)lo, 32, 64);
        __syncthreads();                                               // (the previous source's boundaries have been read)
        cstart[tid] = 0u; cend[tid] = 0u;
        __syncthreads();
        if (rb == re) return false;
        if (re - rb > 4096u) {
            // a long run (the cells under a depth camera hold hundreds of points each: 75 k records in one workgroup's run): every
            // thread bisects for the start of ITS cell -- 17 dependent loads, all 256 cells at once -- instead of the four waves
            // scanning the run's keys round by round (10 us of the depth image's 74 us walk)
            uint32_t lo = rb, hi = re;                                     // first record whose cell is >= tid
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if ((keys[mid] & (uint32_t)(NT - 1)) < (uint32_t)tid) lo = mid + 1u; else hi = mid;
            }
            cstart[tid] = lo;
            __syncthreads();
            cend[tid] = tid + 1 < NT ? cstart[tid + 1] : re;
            __syncthreads();
            return true;
        }
        // cell boundaries of the workgroup's run: the records are sorted by cell, so a cell starts -- and the one before it
        // ends -- where the key's cell changes.  The waves take the rounds in turn; rounds of sixteen loads for long runs: the run
        // of the cells under the sensor is thousands of records long, and one load per round made this loop a chain of memory
        // latencies (half of the kernel's time).
        {
            auto scan_rounds = [&](auto Uc, uint32_t p_begin, uint32_t p_end) {
                constexpr int U = decltype(Uc)::value;
                for (uint32_t p0 = p_begin + (uint32_t)w * 64u * U; p0 < p_end; p0 += (uint32_t)NW * 64u * U) {   // wave-uniform
                    uint32_t c[U];
                    // the cell of the record before the round's first (another wave's round): one more load
                    uint32_t carry = p0 > rb ? (keys[p0 - 1u] & (uint32_t)(NT - 1)) : 0xffffffffu;
#pragma unroll
                    for (int u = 0; u < U; ++u) { const uint32_t p = p0 + 64u * u + (uint32_t)lane; c[u] = keys[min(p, re - 1u)] & (uint32_t)(NT - 1); }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t p = p0 + 64u * u + (uint32_t)lane;
                        const bool live = p < re;
                        const uint32_t cell = live ? c[u] : 0xfffffffeu;
                        uint32_t prev = (uint32_t)__shfl_up((int)cell, 1, 64);
                        if (lane == 0) prev = carry;
                        carry = (uint32_t)__builtin_amdgcn_readlane((int)cell, 63);
                        if (live && cell != prev) { cstart[cell] = p; if (prev < (uint32_t)NT) cend[prev] = p; }
                        if (p + 1u == re) cend[cell] = re;
                    }
                }
            };
            const uint32_t long_part = re - rb >= 8192u ? ((re - rb) / 4096u) * 4096u : 0u;
            if (long_part) scan_rounds(std::integral_constant<int, 16>{}, rb, rb + long_part);
            scan_rounds(std::integral_constant<int, 2>{}, rb + long_part, re);
        }
        __syncthreads();
        return true;
    };

    // ---- which cell this thread walks.  A wave lasts as long as its longest run, and the runs of 64 neighbouring cells differ
    //      widely (a LiDAR ring crosses some cells of a row and misses the next: on C4 the lanes of a wave were busy half of the
    //      time, 0.51 = sum of the runs / 64 x the longest; a depth image 0.57).  So the workgroup's cells are handed out in
    //      descending order of their run length -- wave 0 takes the 64 longest, the last wave the empty ones: 0.80 on C4 -- by a
    //      counting sort over the run lengths (exact below 64, eight steps per octave above).
    uint32_t c = (uint32_t)tid;
    bool have = false;
    {
        have = find_bounds(a.key, run_lo, run_hi);
        sh_e[tid] = e_t; sh_s[tid] = s_t;
        if constexpr (LOWEST) sh_l[tid] = l_t;
        if (tid < 128) hist[tid] = 0u;
        __syncthreads();
        const uint32_t n_t = cend[tid] - cstart[tid];
        uint32_t bin = n_t;
        if (n_t >= 64u) { const uint32_t lg = 31u - (uint32_t)__clz((int)n_t); bin = 64u + min(63u, (lg - 6u) * 8u + ((n_t >> (lg - 3u)) & 7u)); }
        const uint32_t rank = atomicAdd(&hist[bin], 1u);
        __syncthreads();
        if (w == 0) {                                                  // hist[b] -> cells with a longer run than bin b's
            const uint32_t v1 = hist[127 - 2 * lane], v0 = hist[126 - 2 * lane];
            const uint32_t incl = wave_inclusive_scan(v1 + v0), excl = incl - (v1 + v0);
            hist[127 - 2 * lane] = excl; hist[126 - 2 * lane] = excl + v1;
        }
        __syncthreads();
        perm[hist[bin] + rank] = (uint16_t)tid;
        __syncthreads();
        c = perm[tid];
    }
    const int row = (tr << 5) + (q4 << 3) + (int)(c >> 5), col = (tc << 5) + (int)(c & 31u);
    const bool owned = row >= a.row0 && row < a.row1 && col < L;
    const size_t g = owned ? (size_t)row * L + col : 0;
    const float e0 = sh_e[c], s0 = sh_s[c];
    size_t lgeo = 0; float lw = 0.0f, lw0 = 0.0f;
    if constexpr (LOWEST) {
        int gr = row - a.start0, gc = col - a.start1;
        gr += gr < 0 ? L : 0; gc += gc < 0 ? L : 0;
        lgeo = owned ? (size_t)gr * L + gc : 0;
        lw0 = lw = sh_l[c];
    }

    float ce = e0, cs = s0;
    // From sweep `cur` to sweep `to`: the floor that ends every Fuse (GPU:533-534), then the next sweep's increment.  The lanes of
    // a wave stand at unrelated sweeps, so this loop runs as long as the lane with the widest gap needs; the increment of the
    // lane's NEXT sweep is kept in a register (fetched behind the previous use), so that the common one-sweep gap costs no LDS
    // round trip.  (Tried: the first two sweeps of a gap as straight-line predicated code -- every record then pays for them,
    // 69 -> 144 us on C4.)
    const uint32_t last_sw = (uint32_t)(a.n_sweeps > 0 ? a.n_sweeps - 1 : 0);
    SweepReplay rp; rp.cur = 0u; rp.w0 = rp.w1 = rp.w2 = rp.w3 = 0.0f;
    uint32_t wlast = 0xffffffffu, sweeps_seen = 0, last_sweep = 0xffffffffu, n_total = 0;

    // ---- the run of cell c in one source (its boundaries are in LDS)
    auto walk_run = [&](const uint32_t* __restrict__ keys, const uint2* __restrict__ hvs, const uint32_t* __restrict__ srcs) {
        const uint32_t first = cstart[c], n = cend[c] - first;
        n_total += n;
        // The cell's own run.  The 64 lanes read 64 different streams, and a wave load whose lanes fall into 64 different cache
        // lines occupies the CU's vector L1 for 64 cycles whatever its width: with one 8-byte and one 4-byte load per step the
        // waves of a CU queue up at the L1 (125 us for the 32 sweeps of C4).  So the records come in GROUPS of four steps, as
        // 16-byte loads -- two {h, var} pairs, four keys per request: 0.75 requests per step instead of 2 -- and three groups
        // are in flight (the address is clamped to the lane's last group: never a branch round a load; the arrays are padded).
        struct __attribute__((packed, aligned(4))) Quad { uint32_t x, y, z, w; };
        struct Group { Quad h01, h23, k4, s4; };
        const uint32_t nmax = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(n), 63);    // the wave's longest run: no loads beyond it
        const uint32_t glast = n ? (n - 1u) >> 2 : 0u;
        const uint2* hp = hvs + (n ? first : rb);
        const uint32_t* kp = keys + (n ? first : rb);
        const uint32_t* sp = srcs + (n ? first : rb);
        auto load_group = [&](uint32_t gi, Group& G) {
            if (4u * gi < nmax) {                                      // wave-uniform
                const uint32_t r = 4u * min(gi, glast);
                G.h01 = *reinterpret_cast<const Quad*>(hp + r);
                G.h23 = *reinterpret_cast<const Quad*>(hp + r + 2u);
                if (KEYED) G.k4 = *reinterpret_cast<const Quad*>(kp + r);
                if (ATTR) G.s4 = *reinterpret_cast<const Quad*>(sp + r);
            }
        };
        auto step = [&](uint32_t idx, uint32_t hb, uint32_t vb, uint32_t cur_k, uint32_t cur_s) {
            const bool live = idx < n;
            const float h = __uint_as_float(hb), v = __uint_as_float(vb);
            if constexpr (KEYED) {
                const uint32_t sw = cur_k >> a.id_bits;
                if constexpr (HAS_VU) rp.advance(cs, live ? sw : rp.cur, vu, a.var_floor);
                if constexpr (COUNT_SWEEPS) { if (live && sw != last_sweep) { ++sweeps_seen; last_sweep = sw; } }
            }
            float e2 = ce, s2 = cs;
            const bool taken = fuse_step(e2, s2, h, v, a.mahal, a.var_floor);
            const bool fl = live && (!LOWEST || h != -1.0f);           // GPU:482 (only LOWEST passes carry such records)
            ce = fl ? e2 : ce; cs = fl ? s2 : cs;
            if constexpr (LOWEST) { const float l2 = lowest_step(lw, h, v); lw = live ? l2 : lw; }
            if constexpr (ATTR != 0) { if (fl && taken && (cur_s & 0x80000000u)) wlast = cur_s & 0x7fffffffu; }
        };
        auto run_group = [&](uint32_t i0, const Group& G) {
            step(i0, G.h01.x, G.h01.y, G.k4.x, G.s4.x);
            step(i0 + 1u, G.h01.z, G.h01.w, G.k4.y, G.s4.y);
            step(i0 + 2u, G.h23.x, G.h23.y, G.k4.z, G.s4.z);
            step(i0 + 3u, G.h23.z, G.h23.w, G.k4.w, G.s4.w);
        };
        // kWalkDepth groups in flight.  (Six instead of three: no gain on a depth image, whose 300-point cells are bound by the
        // recurrence itself -- about 70 dependent-issue instructions per record on one SIMD -- and C5 slower by 10 %: the
        // registers cost a wave per SIMD.)
        Group G[kWalkDepth];
#pragma unroll
        for (int d = 0; d < kWalkDepth; ++d) { G[d] = Group{}; load_group((uint32_t)d, G[d]); }
        for (uint32_t gi = 0; 4u * gi < nmax; gi += (uint32_t)kWalkDepth) {   // wave-uniform
#pragma unroll
            for (int d = 0; d < kWalkDepth; ++d) {
                if (4u * (gi + (uint32_t)d) >= nmax) break;
                run_group(4u * (gi + (uint32_t)d), G[d]);
                load_group(gi + (uint32_t)(d + kWalkDepth), G[d]);
            }
        }
    };

    // ---- the same run through the PLAIN chain loop (k_fuse_block has the reasoning): every record of the pass has |h| <= 2^28 and
    //      2^-28 <= v <= 2^28 (k_sort_project says so), every cell of the wave starts in range, the pass's constants are in range --
    //      then a step looks at |N1| >= 2^-60 and at the threshold band only, both quotients share one refined reciprocal, and the
    //      records come through a pipeline of three groups of four whose waits the compiler can count: the guarded loop's rotating
    //      groups sit behind conditional loads, and it waits for EVERY load in flight at the head of each group (`s_waitcnt vmcnt(0)`:
    //      a memory round trip per four steps, half of a depth image's 248 ns per step).  Long runs only: the pipeline always runs
    //      whole rounds of twelve steps.
    auto walk_run_plain = [&](const uint2* __restrict__ hvs) {
        typedef float v2f __attribute__((ext_vector_type(2)));
        struct __attribute__((packed, aligned(4))) Quad { uint32_t x, y, z, w; };
        const uint32_t first = cstart[c], n = cend[c] - first;
        n_total += n;
        const uint32_t nmax = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(n), 63);
        const uint32_t glast = n ? (n - 1u) >> 2 : 0u;
        const uint2* hp = hvs + (n ? first : rb);
        const float fl_ = a.var_floor, thr = a.mahal, band = 1e-5f * fabsf(a.mahal);
        auto load2 = [&](uint32_t gi, Quad& q01, Quad& q23) {
            const uint32_t r = 4u * min(gi, glast);
            q01 = *reinterpret_cast<const Quad*>(hp + r);
            q23 = *reinterpret_cast<const Quad*>(hp + r + 2u);
        };
        auto step = [&](uint32_t idx, uint32_t hb, uint32_t vb) {
            const bool live = idx < n;
            const float h = __uint_as_float(hb), v = __uint_as_float(vb);
            // GPU:500-501.  ONE v_max_f32 on the chain's critical path instead of a compare and a select: the plain loop never sees a
            // NaN state (state_ok above; in-range arithmetic makes none), which is the only value the two forms treat differently
            float sf;
#if defined(__HIP_DEVICE_COMPILE__)
            asm("v_max_f32 %0, %2, %1" : "=v"(sf) : "v"(cs), "s"(fl_));
#else
            sf = cs < fl_ ? fl_ : cs;
#endif
            const float m = fabsf(h - ce) * __builtin_amdgcn_rsqf(sf);         // GPU:502, see fuse_step
            const float D = sf + v;                                            // GPU:518, 519
            v2f N; N.x = sf * h + v * ce; N.y = v * sf;
            const bool rare = ((fabsf(m - thr) <= band) | !(fabsf(N.x) >= 8.673617379884035e-19f)) & live;   // 2^-60
            const float r0 = __builtin_amdgcn_rcpf(D);
            const float rr = __builtin_fmaf(__builtin_fmaf(-D, r0, 1.0f), r0, r0);
            const v2f rr2 = {rr, rr}, nD2 = {-D, -D};
            v2f q = N * rr2;
            v2f t = __builtin_elementwise_fma(nD2, q, N);
            q = __builtin_elementwise_fma(t, rr2, q);
            t = __builtin_elementwise_fma(nD2, q, N);
            q = __builtin_elementwise_fma(t, rr2, q);
            const bool outlier = m > thr;
            const bool replace = (ce == kEmptyElevation) | (outlier & (ce < h));   // GPU:484-486, 505-507
            float e2 = replace ? h : (outlier ? ce : q.x);
            float s2 = replace ? v : (outlier ? sf : q.y);
            if (__builtin_expect(__ballot(rare) != 0, 0)) { e2 = ce; s2 = cs; fuse_step<true>(e2, s2, h, v, thr, fl_); }
            ce = live ? e2 : ce; cs = live ? s2 : cs;
        };
        auto run4 = [&](uint32_t i0, const Quad& q01, const Quad& q23) {
            step(i0, q01.x, q01.y); step(i0 + 1u, q01.z, q01.w); step(i0 + 2u, q23.x, q23.y); step(i0 + 3u, q23.z, q23.w);
        };
        Quad a01, a23, b01, b23, c01, c23;
        load2(0u, a01, a23); load2(1u, b01, b23); load2(2u, c01, c23);
        for (uint32_t gi = 0; 4u * gi < nmax; gi += 3u) {              // wave-uniform; every load unconditional (clamped), every wait countable
            run4(4u * gi, a01, a23);        load2(gi + 3u, a01, a23);
            run4(4u * gi + 4u, b01, b23);   load2(gi + 4u, b01, b23);
            run4(4u * gi + 8u, c01, c23);   load2(gi + 5u, c01, c23);
        }
    };

    __syncthreads();                                                   // vu is in LDS
    // Mapvar_update increments queued before this pass, then the one of sweep 0 (GPU:540-547)
    for (int k = 0; k < a.n_pending; ++k) if (cs != kInitVariance) cs += a.pending[k];
    if constexpr (HAS_VU) { if (cs != kInitVariance) cs += vu[0]; rp.refill(vu); }
    bool plain = false;
    if constexpr (FLAGS == 0 && MODE == 0) {
        const bool state_ok = (fabsf(ce) <= 268435456.0f) & (cs <= 268435456.0f);
        const uint32_t longest = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(have ? cend[c] - cstart[c] : 0u), 63);
        plain = a.plain_env != 0 && a.odd_flag != nullptr && *a.odd_flag != a.epoch && __ballot(!state_ok) == 0 && longest >= 24u;   // wave-uniform
    }
    if (have) { if (plain) walk_run_plain(a.hv); else walk_run(a.key, a.hv, a.src); }
    if (__ballot(n_total != 0) == 0 && !a.dense) return;               // nothing reached this wave's cells and nothing is pending
    if constexpr (HAS_VU) { if (a.plain_env) rp.finish_plain(cs, last_sw, vu, a.var_floor); else rp.finish(cs, last_sw, vu, a.var_floor); }
    if (cs < a.var_floor) cs = a.var_floor;                            // GPU:533-534, on every cell

    if (owned) {
        // only what changed goes back (a pass touches a fraction of the cells; whole-tile write-backs were most of the write
        // traffic of the tile kernels)
        if (__float_as_uint(ce) != __float_as_uint(e0)) a.elevation[g] = ce;
        if (__float_as_uint(cs) != __float_as_uint(s0)) a.variance[g] = cs;
        if constexpr (LOWEST) { if (__float_as_uint(lw) != __float_as_uint(lw0)) a.lowest[lgeo] = lw; }
        if constexpr (ATTR != 0) {
            if (wlast != 0xffffffffu) {                                // colour / intensity of the last taken point with all four non-zero (GPU:487-494)
                if (ATTR == 1) {
                    const uint32_t cc = a.rgb[wlast];
                    a.intensity[g] = a.xyzi[wlast].w;
                    a.colorR[g] = (int)((cc >> 16) & 0xff); a.colorG[g] = (int)((cc >> 8) & 0xff); a.colorB[g] = (int)(cc & 0xff);
                } else {
                    a.intensity[g] = a.f_I[wlast];
                    a.colorR[g] = a.f_R[wlast]; a.colorG[g] = a.f_G[wlast]; a.colorB[g] = a.f_B[wlast];
                }
            }
        }
    }
    if (a.counters) {                                                  // distinct touched cells: per pass, or summed over the sweeps
        const uint32_t mine = COUNT_SWEEPS ? sweeps_seen : (n_total ? 1u : 0u);
        const uint32_t s = wave_inclusive_scan(mine);
        if (lane == 63 && s) atomicAdd(&a.counters[1], (unsigned long long)s);
    }
}

// ------------------------------------------------------------------------------------------
// k_fuse_block : BLOCK-sorted records -- one workgroup per block of 256 cells (eight rows of a tile), one thread per cell
// ------------------------------------------------------------------------------------------
// The sort has brought the records of a block together and left them in INPUT ORDER.  The workgroup takes them a batch of B at a
// time (coalesced loads), orders the batch by cell in LDS with a stable counting sort over the 256 cells -- wave w ranks the w-th
// contiguous share; the lanes of a wave instruction that hold the same cell find each other through a 64-bit OR mask per (wave,
// cell), a per-(wave, cell) cursor carries the count from step to step (a wave's LDS operations execute in order) -- and then
// every thread runs ITS cell's records of the batch, in input order, through the reference's recurrence (fuse_step,
// GPU:480-531), the sweeps' variance increments (GPU:540-547) and floors (GPU:533-534) replayed in between; the cell state stays
// in registers from batch to batch.  The chains are read from LDS: 64 lanes reading 64 unrelated runs cost a few bank conflicts
// instead of 64 cache lines per load (k_fuse_walk), and no kernel of the sort ever orders records by cell.
// A batch takes as long as its longest chain, so a block of R records takes sum over its batches of (longest chain in the batch):
// close to the longest chain of the block when every batch spreads over the block's cells (LiDAR sweeps), far above it when a
// batch holds a few cells' records only (the rows of a depth image) -- such passes take the CELL-sorted form (k_fuse_walk).
// Several sources (multi-GPU strip owner): the block's records of every source, in rank order = input order, form one sequence.
constexpr int kBlkNT = 256;

template <int B, bool KEYED, bool ATTR>
__host__ __device__ constexpr size_t block_walk_lds()
{
    return (size_t)B * 8 + (KEYED ? (size_t)B * 2 : 0) + (ATTR ? (size_t)B * 4 : 0)                     // staged batch
           + (size_t)4 * 320 * 4 + (B > 512 ? (size_t)4 * 320 * 8 : 0) + (size_t)4 * 256 * 4;           // cursors, masks (rounds of more than 512 records), bases
}

// MULTI: several sources (a multi-GPU strip owner).  A kernel of its own: with both forms in one, the values the next batch's loads
// bring were copied between the two forms' registers where the paths join -- behind an `s_waitcnt vmcnt(0)`, right after the loads.
template <int FLAGS, int MODE, int B, bool MULTI>
__global__ __launch_bounds__(kBlkNT) void k_fuse_block(WalkArgs a)
{
    constexpr int ATTR = FLAGS & 3;
    constexpr bool LOWEST = (FLAGS & 4) != 0;
    constexpr bool HAS_VU = (MODE & 1) != 0, COUNT_SWEEPS = (MODE & 2) != 0, KEYED = HAS_VU || COUNT_SWEEPS;
    constexpr int NT = kBlkNT, NW = NT / 64, K = B / NT;
    static_assert(NW == 4 && K >= 1 && K <= 32 && B <= 65536, "batch geometry");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_sort[];
    uint2* st_hv = reinterpret_cast<uint2*>(lds_sort);                                  // [B]   the batch, ordered by cell
    // Light rounds (B <= 512: blocks of a few hundred records, one or two wave instructions per wave and round) match the lanes of
    // equal cells by BALLOT, one per bit of the cell number -- 24 scalar / vector instructions per step instead of three LDS round
    // trips -- and keep no mask table: 18 KB of LDS per workgroup instead of 28, eight workgroups per CU instead of five (C5's walk alone
    // 114 -> 111 us, 153 -> 133 beside the sort chains; the call's period did not move: it is the sum of the kernels' work).
#ifndef GEM_BALLOT_MAX
#define GEM_BALLOT_MAX 512
#endif
    constexpr bool BALLOT = B <= GEM_BALLOT_MAX;
    unsigned long long* wpm = reinterpret_cast<unsigned long long*>(st_hv + B);         // [NW][320] who shares my cell in this wave instruction (!BALLOT)
    uint32_t* wcur = reinterpret_cast<uint32_t*>(wpm + (BALLOT ? 0 : NW * 320));        // [NW][320] records of (wave, cell) so far
    uint32_t* cbase = wcur + NW * 320;                                                  // [NW][256] where the run of (wave, cell) starts in the batch
    uint32_t* st_src = cbase + NW * 256;                                                // [B] (ATTR)
    uint16_t* st_sw = reinterpret_cast<uint16_t*>(st_src + (ATTR ? B : 0));             // [B] (KEYED) the record's sweep
    __shared__ uint32_t scratch[16], ccnt[NT], phist[128];
    __shared__ uint16_t perm[NT];
    __shared__ float sh_e[NT], sh_s[NT], sh_l[LOWEST ? NT : 1];
    __shared__ float vu[HAS_VU ? kWalkMaxSweeps + 8 : 1];
    __shared__ uint32_t seg_first[kMaxRanks], seg_off[kMaxRanks + 1];                  // per source: first record of the block; prefix of the counts
    __shared__ unsigned long long seg_key[kMaxRanks], seg_hv[kMaxRanks];
    __shared__ uint32_t blk_odd;                                                       // some record or cell state of this block is outside the plain range (below)
    // LIGHT rounds the fast way (below): arrival slots per cell + a flag "some cell got more than kLightSlots records this round"
    constexpr bool FAST_LIGHT = B <= 512 && FLAGS == 0 && !COUNT_SWEEPS;
    constexpr int kLightSlots = 6;
    __shared__ uint16_t cslot[FAST_LIGHT ? NT * kLightSlots : 1];
    __shared__ uint32_t blk_over;
    const int tid = (int)threadIdx.x, lane = lane_id(), w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The PLAIN chain loop (the blocks of every LiDAR pass): one step of the recurrence is a single wave issuing ~90 instructions in
    // order, a third of them guards -- the exponent ranges that let both Kalman quotients share one reciprocal (fuse_step<true>), the
    // subnormal test of the Mahalanobis shortcut, the increments' replay with a ballot and a branch per sweep.  All of that is
    // decidable OUTSIDE the chain: if every record of the block has |h| <= 2^28 and 2^-28 <= v <= 2^28 (checked where the records
    // are placed, in parallel), every cell starts with |e| <= 2^28 and s <= 2^28 (checked here), and the pass's floor / threshold /
    // increments are in range (a.plain_env, checked by the host), then at every step sf = max(s, floor) lies in [2^-28, 2^30],
    // D = sf + v in [2^-28, 2^31], N2 = v sf in [2^-56, 2^58], |N1| <= 2^59, |e| stays <= 2^28 -- and a step only has to look at
    // |N1| >= 2^-60 (a cancellation) and at the threshold band.  Anything else takes the loop as it was.
    constexpr bool PLAIN_OK = FLAGS == 0 && !COUNT_SWEEPS;
    constexpr float kPlainHi = 268435456.0f, kPlainLo = 3.7252902984619140625e-9f;      // 2^28, 2^-28
    // (k_sort_project has looked at every record of THIS device's pass already and says so in one word: then only the cells' states
    //  are checked here; records that came from other ranks carry no such word)
    const bool records_plain = !MULTI && a.odd_flag != nullptr && *a.odd_flag != a.epoch;      // block-uniform
    if (tid == 0) { blk_odd = 0u; blk_over = 0u; }
    ccnt[tid] = 0u;

    // workgroup -> block of 256 cells: tile rows centre-first when all workgroups are resident at once (see k_fuse_walk), else memory
    // order.  (Tried: the blocks sorted by record count on the device and dealt out heavy / middle / light to consecutive
    // workgroups, so that no CU holds three blocks of the sensor's neighbourhood at once -- a kernel more, no gain: the walk ends
    // with the chains of its heaviest block whatever runs beside it.)
    int tile, q4 = (int)(blockIdx.x & 3);
    {
        const int tpr = a.tiles_per_row, rnk = (int)(blockIdx.x >> 2);
        if (a.walk_order) {
            const int bi = rnk / tpr, bj = rnk - bi * tpr;
            const int oi = (bi & 1) ? -((bi + 1) >> 1) : (bi >> 1);
            int r = a.center_tr + oi; r = r < 0 ? r + tpr : (r >= tpr ? r - tpr : r);
            tile = r * tpr + bj;
        } else tile = rnk;
    }
    const int tr = tile / a.tiles_per_row, tc = tile - tr * a.tiles_per_row;
    if ((tr << 5) >= a.row1 || (tr << 5) + 32 <= a.row0) return;       // a tile row outside this device's strip
    const uint32_t idmask = (1u << a.id_bits) - 1u;
    const uint32_t id0 = ((uint32_t)tile << 10) | ((uint32_t)q4 << 8); // the cell ids of this block: id0 .. id0 + 255
    const int n_src = MULTI ? (a.n_src <= 1 ? 1 : a.n_src) : 1;
    if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 16] = __builtin_readcyclecounter();

    // ---- cell tid of the block: its map values are fetched now, coalesced, in flight behind the search; which cell the thread
    //      WALKS is decided after the first batch (below)
    const int L = a.L;
    const int row_t = (tr << 5) + (q4 << 3) + (tid >> 5), col_t = (tc << 5) + (tid & 31);
    const bool owned_t = row_t >= a.row0 && row_t < a.row1 && col_t < L;
    const size_t g_t = owned_t ? (size_t)row_t * L + col_t : 0;
    // (only REQUESTED here: the block's range, its first batch of records and these are three memory round trips -- for the light
    //  blocks of a big map, a few hundred records each, most of the block's time -- so they are all put in flight before anything
    //  waits: the values go to LDS behind the first batch's loads, below)
    const float e_in = a.elevation[g_t], s_in = a.variance[g_t];
    float l_in = 0.0f;
    if constexpr (LOWEST) {                                            // map_lowest is indexed by the GEOGRAPHIC cell (GPU:430)
        int gr = row_t - a.start0, gc = col_t - a.start1;
        gr += gr < 0 ? L : 0; gc += gc < 0 ? L : 0;
        l_in = a.lowest[owned_t ? (size_t)gr * L + gc : 0];
    }
    bool odd = false;                                                  // this thread has seen a value outside the plain range
    if (tid < 128) phist[tid] = 0u;
    perm[tid] = (uint16_t)tid;
    if constexpr (HAS_VU) for (int i = tid; i < a.n_sweeps + 8; i += NT) vu[i] = i < a.n_sweeps ? a.var_updates[i] : 0.0f;   // (nothing reads past sweep n_sweeps + 3)

    // ---- where the block's records are in every source.  One-pass sort: the last pass's bins are the blocks.  Otherwise a
    //      32-ary search, both ends at once (lanes 0-31 look for the first record of the block, lanes 32-63 for the first one
    //      behind it), inside the run of the last pass's bin (own sort) or over the whole source (records received from a rank);
    //      the waves share the sources out.
    auto search = [&](const uint32_t* __restrict__ keys, uint32_t lo, uint32_t hi, uint32_t& first, uint32_t& end) {
        const uint32_t target = id0 + (uint32_t)(lane >> 5) * (uint32_t)NT, l5 = (uint32_t)lane & 31u;
        while (__ballot(lo < hi) != 0) {                               // wave-uniform
            const uint32_t n = hi - lo, s = (n + 32u) / 33u;           // probes lo + j s + s - 1, j = 0..31
            const uint32_t pos = lo + l5 * s + s - 1u;
            const bool probe = lo < hi && pos < hi;
            const uint32_t id = probe ? (keys[pos] & idmask) : 0xffffffffu;
            const uint64_t bl = __ballot(probe && id < target);
            const uint32_t k = (uint32_t)__popc((uint32_t)(bl >> (lane & 32)));   // a prefix of the probes: the block ids are sorted
            if (lo < hi) { const uint32_t nl = lo + k * s; if (k < 32u) hi = min(hi, nl + s - 1u); lo = nl; }
        }
        first = (uint32_t)__shfl((int)lo, 0, 64);
        end = (uint32_t)__shfl((int)lo, 32, 64);
    };
    uint32_t first0 = 0, r_single = 0;                                 // single source: the block's first record, its records (block-uniform)
    if constexpr (!MULTI) {
        uint32_t end;
        if (a.ranges) { const uint2 r = a.ranges[id0 >> 8]; first0 = r.x; end = r.y; }
        else { const uint32_t bin = id0 >> a.bin_shift; first0 = a.bin_base[bin]; end = a.bin_base[bin + 1]; }
        if (first0 == end && !a.dense) return;
        if (!a.ranges && !a.exact_bins && first0 != end) search(a.key, first0, end, first0, end);
        r_single = end - first0;
    } else {
        for (int s = w; s < n_src; s += NW) {                          // wave-uniform
            uint32_t first = 0, end = 0;
            if (a.src_ranges[s]) {                                     // the source's own block ranges came with its records
                const uint2 r = a.src_ranges[s][(id0 >> 8) - a.blk0];
                first = r.x - (r.y != r.x ? a.src_base[s] : r.x); end = first + (r.y - r.x);
            } else if (a.src_n[s]) search(a.src_key[s], 0u, a.src_n[s], first, end);
            if (lane == 0) {
                seg_first[s] = first; cbase[s] = end - first;         // (cbase: free until the first batch)
                seg_key[s] = (unsigned long long)(uintptr_t)a.src_key[s]; seg_hv[s] = (unsigned long long)(uintptr_t)a.src_hv[s];
            }
        }
        __syncthreads();
        if (tid == 0) { uint32_t o = 0; for (int s = 0; s < n_src; ++s) { seg_off[s] = o; o += cbase[s]; } seg_off[n_src] = o; }
    }
    auto zero_tables = [&]() {
        for (int i = tid; i < NW * 320; i += NT) { wcur[i] = 0u; if constexpr (!BALLOT) wpm[i] = 0ull; }
        __syncthreads();
    };
    if constexpr (MULTI) zero_tables();                                // (and seg_off is complete)
    const uint32_t R = MULTI ? seg_off[n_src] : r_single;              // records of this block, all sources
    if (R == 0 && !a.dense) return;                                    // block-uniform
    // The kernel ends with its heaviest block, and that block's time is its chains: a wave issuing one instruction after the
    // other.  VALU issue on a SIMD is arbitrated by priority, then age (MI355X_MICROARCH.md): the waves of a heavy block take
    // precedence over whatever shares their SIMDs -- lighter blocks of this kernel, the next pass's sort kernels.
    if (a.prio_records && R >= (uint32_t)a.prio_records) __builtin_amdgcn_s_setprio(3);
    else if (a.prio_records && R >= (uint32_t)a.prio_records / 4u) __builtin_amdgcn_s_setprio(1);

    // record q of the block's sequence -> its source arrays (global memory: the pointers kept in LDS are cast back to that address
    // space, a generic pointer would make every load a FLAT one that also counts as an LDS operation) and its place in them
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(1))) uint32_t* gkey_t;
    typedef const __attribute__((address_space(1))) uint2* ghv_t;
#else
    typedef const uint32_t* gkey_t;                                    // (the host pass only parses the kernel)
    typedef const uint2* ghv_t;
#endif
    auto locate = [&](uint32_t q, gkey_t& kp, ghv_t& hp) -> uint32_t {
        if constexpr (!MULTI) { kp = (gkey_t)a.key; hp = (ghv_t)a.hv; return first0 + q; }
        int s = 0;
        while (s + 1 < n_src && q >= seg_off[s + 1]) ++s;
        kp = (gkey_t)(uintptr_t)seg_key[s]; hp = (ghv_t)(uintptr_t)seg_hv[s];
        return seg_first[s] + (q - seg_off[s]);
    };

    // ---- the cell this thread walks, and its state.  A wave lasts as long as its longest chain and the chains of 64 neighbouring
    //      cells differ widely (a LiDAR ring crosses some cells of a row and misses the next), so after the first batch the block's
    //      cells are handed out in descending order of their record count in that batch -- wave 0 takes the 64 busiest cells, the
    //      last wave the empty ones -- by a counting sort over the counts (exact below 64, eight steps per octave above).  The
    //      assignment then stays: the state lives in registers from batch to batch.
    uint32_t c = (uint32_t)tid;
    int row = 0, col = 0; bool owned = false; size_t g = 0, lgeo = 0;
    float e0 = 0.0f, s0 = 0.0f, lw = 0.0f, lw0 = 0.0f, ce = 0.0f, cs = 0.0f;
    const uint32_t last_sw = (uint32_t)(a.n_sweeps > 0 ? a.n_sweeps - 1 : 0);
    SweepReplay rp; rp.cur = 0u; rp.w0 = rp.w1 = rp.w2 = rp.w3 = 0.0f;
    uint32_t wlast = 0xffffffffu, sweeps_seen = 0, last_sweep = 0xffffffffu, n_total = 0;
    auto take_cell = [&]() {                                           // (sh_e / sh_s / perm are final: behind a barrier)
        c = perm[tid];
        row = (tr << 5) + (q4 << 3) + (int)(c >> 5); col = (tc << 5) + (int)(c & 31u);
        owned = row >= a.row0 && row < a.row1 && col < L;
        g = owned ? (size_t)row * L + col : 0;
        e0 = sh_e[c]; s0 = sh_s[c];
        if constexpr (LOWEST) {
            int gr = row - a.start0, gc = col - a.start1;
            gr += gr < 0 ? L : 0; gc += gc < 0 ? L : 0;
            lgeo = owned ? (size_t)gr * L + gc : 0;
            lw0 = lw = sh_l[c];
        }
        ce = e0; cs = s0;
        // Mapvar_update increments queued before this pass, then the one of sweep 0 (GPU:540-547)
        for (int k = 0; k < a.n_pending; ++k) if (cs != kInitVariance) cs += a.pending[k];
        if constexpr (HAS_VU) { if (cs != kInitVariance) cs += vu[0]; rp.refill(vu); }
    };

    const uint64_t lt = lanemask_lt();
    const unsigned long long mybit = 1ull << lane;
    unsigned long long* wpm_w = wpm + w * 320;
    uint32_t* wcur_w = wcur + w * 320;
    uint2 hv[K]; uint32_t key[K], src[K];
    uint2 nhv[K]; uint32_t nkey[K], nsrc[K];                           // the NEXT batch's records, in flight behind this batch's chains (see load_batch)
#pragma unroll
    for (int k = 0; k < K; ++k) { nhv[k] = make_uint2(0u, 0u); nkey[k] = 0u; nsrc[k] = 0u; }
    // The wave's share of a batch of nb records: `steps` wave instructions of 64 consecutive records.  load_batch only ISSUES the
    // loads (clamped addresses, nothing looks at the values): the next batch's records are in flight behind the chains of this one.
    auto in_batch = [&](int k, uint32_t steps, uint32_t nb) -> bool {
        return (uint32_t)k < steps && ((uint32_t)w * steps + (uint32_t)k) * 64u + (uint32_t)lane < nb;
    };
    // (Into registers of their OWN, copied at the top of the next round: when the loads went straight into the registers the ranking
    //  reads, those were dead in between, the compiler used them as temporaries of the address arithmetic, and -- a register with a
    //  load possibly still pending from the round before -- put an `s_waitcnt vmcnt` in front of every pair: eight serialised memory
    //  round trips per batch, 40-47 k of the heaviest C4 block's 175 k cycles.  All addresses first, then all loads back to back.)
    auto load_batch = [&](uint32_t P, uint32_t nb) {
        const uint32_t steps = (nb + (uint32_t)NT - 1u) / (uint32_t)NT;
        if constexpr (!MULTI) {
            // one source: a uniform base and a 32-bit byte offset per lane (the loads' scalar-base form: no 64-bit address arithmetic in VGPRs)
#if defined(__HIP_DEVICE_COMPILE__)
            typedef const __attribute__((address_space(1))) char* gbytes_t;
#else
            typedef const char* gbytes_t;
#endif
            const gbytes_t kb = (gbytes_t)a.key, hb = (gbytes_t)a.hv, sb = (gbytes_t)a.src;
            const uint32_t at0 = first0 + P, last = nb - 1u, j0 = (uint32_t)w * steps * 64u + (uint32_t)lane;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if ((uint32_t)k < steps) {                             // block-uniform
                    const uint32_t at = at0 + min(j0 + (uint32_t)k * 64u, last);
                    nkey[k] = *(gkey_t)(kb + (uint32_t)(at * 4u));
                    nhv[k] = *(ghv_t)(hb + (uint32_t)(at * 8u));
                    if (ATTR) nsrc[k] = *(gkey_t)(sb + (uint32_t)(at * 4u));
                }
            }
            return;
        }
        uint32_t at[K]; gkey_t kps[K]; ghv_t hps[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            at[k] = 0u; kps[k] = (gkey_t)a.key; hps[k] = (ghv_t)a.hv;
            if ((uint32_t)k < steps) {                                 // block-uniform
                const uint32_t j = ((uint32_t)w * steps + (uint32_t)k) * 64u + (uint32_t)lane;
                at[k] = locate(P + min(j, nb - 1u), kps[k], hps[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if ((uint32_t)k < steps) { nkey[k] = kps[k][at[k]]; nhv[k] = hps[k][at[k]]; }
        }
    };

    // profiling aid (gem_debug_fuse_stamps): cycle stamps of thread 0 -- start, set-up done, then the sums over the batches of
    // {ranking + waiting for the slowest wave's chains, bases, placement, own chains}, end; records, batches, sum of wave 0's longest chains
    unsigned long long* dbg = a.dbg ? a.dbg + (size_t)blockIdx.x * 16 : nullptr;
    unsigned long long t_prev = 0, acc_rank = 0, acc_base = 0, acc_place = 0, acc_walk = 0, acc_nmax = 0, n_batches = 0, acc_rare = 0, acc_pre = 0;
    if (dbg && tid == 0) { t_prev = __builtin_readcyclecounter(); dbg[1] = t_prev; dbg[7] = R; }
    auto lap = [&](unsigned long long& acc) {
        if (dbg && tid == 0) { const unsigned long long t = __builtin_readcyclecounter(); acc += t - t_prev; t_prev = t; }
    };

    // ---- one step of the PLAIN chain loop (see the head of the kernel for what makes a block plain)
    typedef float v2f __attribute__((ext_vector_type(2)));
    const float fl_ = a.var_floor, thr_ = a.mahal, band_ = 1e-5f * fabsf(a.mahal);
    float un1 = 0.0f, un2 = 0.0f, un3 = 0.0f;                          // increments of sweeps cur + 1 .. cur + 3, fetched a step ahead
    auto plain_begin = [&]() { if constexpr (HAS_VU) { un1 = vu[rp.cur + 1u]; un2 = vu[rp.cur + 2u]; un3 = vu[rp.cur + 3u]; } };
    auto plain_step = [&](const uint2 r, const uint32_t swr, const bool live) {
        const float h = __uint_as_float(r.x), v = __uint_as_float(r.y);
        bool rare = false;
        uint32_t sw = 0;
        float sf_vu = 0.0f;
        if constexpr (HAS_VU) {
            // Between two records of sweeps s < t the cell lives through (t - s) x {floor (GPU:533-534); next sweep's increment
            // (GPU:540-547)}.  The floor is positive, so a floored variance is never the -10 of an empty cell and the increment
            // always applies; the increments are not negative (plain_env), so after the first floor the later ones change
            // nothing: a gap of g sweeps is one floor and g rounded additions.  Gaps of up to three sweeps -- 99.98 % of them
            // on a LiDAR batch, and with 64 lanes per step the rest still matters -- are predicated straight-line code on
            // increments fetched a step ahead; wider ones take the rare branch.  (The guarded loop's way -- a ballot and a
            // branch per sweep of the widest gap in the wave -- was a third of its step.)
            sw = live ? swr : rp.cur;
            const uint32_t gap = sw - rp.cur;
            // The chain's critical path: the three candidate sums from the FLOORED variance without a select in between (the same
            // additions in the same order: bit-identical), one select tree at the end, and the step's floored variance sf straight
            // from it -- max(cs', floor) is cs' once an increment has been added to a floored value, and the floored value itself
            // otherwise.  Six dependent operations from cs to sf instead of nine.
            float cf;
#if defined(__HIP_DEVICE_COMPILE__)
            asm("v_max_f32 %0, %1, %2" : "=v"(cf) : "v"(cs), "v"(fl_));    // (one v_max_f32: no NaN state in the plain loop, see k_fuse_walk)
#else
            cf = cs < fl_ ? fl_ : cs;
#endif
            const float a1 = cf + un1, a2 = a1 + un2, a3 = a2 + un3;
            const float tg = gap >= 3u ? a3 : (gap >= 2u ? a2 : a1);
            cs = gap >= 1u ? tg : cs;                                      // (what the cell's variance is now: off the critical path)
            sf_vu = gap >= 1u ? tg : cf;
            rp.cur += min(gap, 3u);
            rare = gap > 3u;
            un1 = vu[rp.cur + 1u]; un2 = vu[rp.cur + 2u]; un3 = vu[rp.cur + 3u];
        }
        float sf;                                                          // GPU:500-501
        if constexpr (HAS_VU) sf = sf_vu;
        else {
#if defined(__HIP_DEVICE_COMPILE__)
            asm("v_max_f32 %0, %1, %2" : "=v"(sf) : "v"(cs), "v"(fl_));    // (see above)
#else
            sf = cs < fl_ ? fl_ : cs;
#endif
        }
        const float m = fabsf(h - ce) * __builtin_amdgcn_rsqf(sf);         // GPU:502, see fuse_step
        const float D = sf + v;                                            // GPU:518, 519
        v2f N; N.x = sf * h + v * ce; N.y = v * sf;
        rare = (rare | (fabsf(m - thr_) <= band_) | !(fabsf(N.x) >= 8.673617379884035e-19f)) & live;   // 2^-60
        // both quotients from one refined reciprocal (fuse_step<true>), as a pair
        const float r0 = __builtin_amdgcn_rcpf(D);
        const float rr = __builtin_fmaf(__builtin_fmaf(-D, r0, 1.0f), r0, r0);
        const v2f rr2 = {rr, rr}, nD2 = {-D, -D};
        v2f q = N * rr2;
        v2f t = __builtin_elementwise_fma(nD2, q, N);
        q = __builtin_elementwise_fma(t, rr2, q);
        t = __builtin_elementwise_fma(nD2, q, N);
        q = __builtin_elementwise_fma(t, rr2, q);
        const bool outlier = m > thr_;
        const bool replace = (ce == kEmptyElevation) | (outlier & (ce < h));   // GPU:484-486, 505-507
        // (tried and dropped, round 6, both bit-identical: ONE select behind the quotients with the alternatives picked beforehand --
        //  the compiler's schedule got worse, C3's walk 37.8 -> 38.6 us, C4 95.2-96.9 -> 97.1-97.5; lanes without a record sent down
        //  the outlier's way instead of a select on `live` -- nothing, 37.2-37.8 / 95.9-96.7)
        float e2 = replace ? h : (outlier ? ce : q.x);
        float s2 = replace ? v : (outlier ? sf : q.y);
        if (__builtin_expect(__ballot(rare) != 0, 0)) {                    // wave-uniform: the step as the guarded loop takes it
            if (dbg && tid == 0) ++acc_rare;
            if constexpr (HAS_VU) {
                while (__ballot(rp.cur < sw) != 0) { if (rp.cur < sw) rp.one(cs, vu[rp.cur + 1u], fl_); }
                un1 = vu[rp.cur + 1u]; un2 = vu[rp.cur + 2u]; un3 = vu[rp.cur + 3u];
            }
            e2 = ce; s2 = cs;
            fuse_step<true>(e2, s2, h, v, thr_, fl_);
        }
        ce = live ? e2 : ce; cs = live ? s2 : cs;
    };
    // ... and of the guarded one (no colours, no lowest scan points: the light rounds below)
    auto guarded_step = [&](const uint2 r, const uint32_t sw, const bool live) {
        const float h = __uint_as_float(r.x), v = __uint_as_float(r.y);
        if constexpr (HAS_VU) rp.advance(cs, live ? sw : rp.cur, vu, a.var_floor);
        float e2 = ce, s2 = cs;
        fuse_step<true>(e2, s2, h, v, a.mahal, a.var_floor);
        ce = live ? e2 : ce; cs = live ? s2 : cs;
    };

    uint32_t parity = 0;
    if (R) load_batch(0u, min(R, (uint32_t)B));
    // (light kernels clear the general way's tables only when a round needs them: `tables_clear`)
    bool tables_clear = false;
    if constexpr (!MULTI) { if (!(FAST_LIGHT && a.light_fast)) { zero_tables(); tables_clear = true; } else __syncthreads(); }
    else tables_clear = true;
    sh_e[tid] = e_in; sh_s[tid] = s_in;                                // (the first wait for memory: the map values, with the first batch in flight behind them)
    if constexpr (LOWEST) sh_l[tid] = l_in;
    if constexpr (PLAIN_OK) odd = !(fabsf(e_in) <= kPlainHi) || !(s_in <= kPlainHi);
    if (R == 0) take_cell();                                           // (a dense pass over a block without records: every thread keeps cell tid)
    for (uint32_t P = 0; P < R; P += (uint32_t)B) {                    // block-uniform
        const uint32_t nb = min(R - P, (uint32_t)B), steps = (nb + (uint32_t)NT - 1u) / (uint32_t)NT;
        // (every prefetch register is USED here, also those of steps this round does not have: a register whose load of the round before
        //  "may still be pending" as far as the compiler can tell costs an s_waitcnt vmcnt(0) wherever it is written next -- in the
        //  middle of the next prefetch, see load_batch)
#pragma unroll
        for (int k = 0; k < K; ++k) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(nkey[k]), "+v"(nhv[k].x), "+v"(nhv[k].y));
            if (ATTR) asm volatile("" : "+v"(nsrc[k]));
#endif
            key[k] = nkey[k]; hv[k] = nhv[k]; if (ATTR) src[k] = nsrc[k];
        }
        // ---- LIGHT rounds the fast way.  A big map's blocks hold a few hundred records each -- one or two per cell and round -- and
        //      the general way below (a stable rank per (wave, cell), a block scan, the placement: five barriers) is most of such a
        //      block's time.  Here every record takes an ARRIVAL slot of its cell (one LDS atomic) and leaves its position in the
        //      round there; the cell's thread sorts its at most six positions (a 12-comparator network: ascending position = input
        //      order) and runs the records straight from the stage: two barriers.  A round in which some cell gets more than six
        //      records takes the general way (the counts are cleared first).
        if constexpr (FAST_LIGHT) {
            if (a.light_fast) {                                        // (debug knob; block-uniform)
                bool over = false;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (in_batch(k, steps, nb)) {
                        const uint32_t j = ((uint32_t)w * steps + (uint32_t)k) * 64u + (uint32_t)lane, cell = key[k] & 255u;
                        st_hv[j] = hv[k];
                        if constexpr (KEYED) st_sw[j] = (uint16_t)(key[k] >> a.id_bits);
                        const uint32_t slot = atomicAdd(&ccnt[cell], 1u);
                        if (slot < (uint32_t)kLightSlots) cslot[cell * kLightSlots + slot] = (uint16_t)j; else over = true;
                        if constexpr (PLAIN_OK) {
                            if (!records_plain) {
                                const float hh = __uint_as_float(hv[k].x), vv = __uint_as_float(hv[k].y);
                                odd = odd || !(fabsf(hh) <= kPlainHi) || !(vv >= kPlainLo) || !(vv <= kPlainHi);
                            }
                        }
                    }
                }
                if constexpr (PLAIN_OK) { if (odd) blk_odd = 1u; }
                if (over) blk_over = 1u;
                __syncthreads();
                lap(acc_rank);
                if (blk_over == 0u) {                                  // block-uniform
                    if (P + (uint32_t)B < R) load_batch(P + (uint32_t)B, min(R - P - (uint32_t)B, (uint32_t)B));
                    if (P == 0u) take_cell();                          // (light passes hand the cells out in place: c == tid)
                    const uint32_t cn = ccnt[tid];
                    ccnt[tid] = 0u;
                    uint32_t sp[kLightSlots];
#pragma unroll
                    for (int i = 0; i < kLightSlots; ++i) sp[i] = (uint32_t)i < cn ? (uint32_t)cslot[tid * kLightSlots + i] : 0xffffu;
                    auto cx = [&](int i, int j) { const uint32_t lo = min(sp[i], sp[j]), hi = max(sp[i], sp[j]); sp[i] = lo; sp[j] = hi; };
                    n_total += cn;
                    const uint32_t nmax = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(cn), 63);
                    // (unused slots hold 0xffff and sort to the end; the network is as wide as the wave's busiest cell needs)
                    if (nmax > 4u) { cx(0, 1); cx(2, 3); cx(4, 5); cx(0, 2); cx(3, 5); cx(1, 4); cx(0, 1); cx(2, 3); cx(4, 5); cx(1, 2); cx(3, 4); cx(2, 3); }
                    else if (nmax > 2u) { cx(0, 1); cx(2, 3); cx(0, 2); cx(1, 3); cx(1, 2); }
                    else if (nmax > 1u) cx(0, 1);
                    lap(acc_pre);
                    bool plain = false;
                    if constexpr (PLAIN_OK) plain = a.plain_env != 0 && blk_odd == 0u;
                    if (plain) { if constexpr (PLAIN_OK) plain_begin(); } else { if constexpr (HAS_VU) rp.refill(vu); }
#pragma unroll
                    for (int i = 0; i < kLightSlots; ++i) {
                        if ((uint32_t)i < nmax) {                      // wave-uniform
                            const uint32_t at = min(sp[i], (uint32_t)B - 1u);
                            const uint2 r = st_hv[at];
                            uint32_t sw = 0;
                            if constexpr (KEYED) sw = st_sw[at];
                            if (plain) { if constexpr (PLAIN_OK) plain_step(r, sw, (uint32_t)i < cn); }
                            else guarded_step(r, sw, (uint32_t)i < cn);
                        }
                    }
                    lap(acc_walk);
                    acc_nmax += nmax; ++n_batches;
                    __syncthreads();                                   // the stage, the slots and the counts are free for the next round
                    continue;
                }
                ccnt[tid] = 0u;                                        // the general way for this round
                if (!tables_clear) { zero_tables(); tables_clear = true; } else __syncthreads();
                if (tid == 0) blk_over = 0u;
            }
        }
        // ---- 1. stable rank of every record among the records of its cell in the wave's share.  Phase by phase: a wave's LDS
        //         operations execute in order, so the K steps' round trips overlap.
        uint32_t rk[K]; uint64_t peers[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            peers[k] = 0ull;
            if ((uint32_t)k < steps) {
                const bool valid = in_batch(k, steps, nb);
                if constexpr (BALLOT) {
                    const uint64_t p = wave_peers(valid, key[k] & 255u, 8);
                    peers[k] = valid ? p : mybit;                      // (a lane without a record: a group of its own, as below)
                } else {
                    unsigned long long* m = wpm_w + (valid ? (key[k] & 255u) : 256u + (uint32_t)lane);   // no branch: a lane without a record has a slot of its own
                    __hip_atomic_fetch_or(m, mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    peers[k] = (uint64_t)__hip_atomic_load(m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(m, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // for the next step (every lane of the group: the same value)
                }
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            rk[k] = 0u;
            if ((uint32_t)k < steps) {
                const bool valid = in_batch(k, steps, nb);
                uint32_t* cp = wcur_w + (valid ? (key[k] & 255u) : 256u + (uint32_t)lane);
                const uint32_t rank = (uint32_t)__popcll(peers[k] & lt);
                const uint32_t old = __hip_atomic_load(cp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // before the group's first lane adds the group
                if (valid && rank == 0u) __hip_atomic_fetch_add(cp, (uint32_t)__popcll(peers[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                rk[k] = old + rank;
            }
        }
        __syncthreads();
        lap(acc_rank);
        // ---- 2. per cell: the waves in order, the cell's place in the batch
        const uint32_t n0 = wcur[0 * 320 + tid], n1 = wcur[1 * 320 + tid], n2 = wcur[2 * 320 + tid], n3 = wcur[3 * 320 + tid];
        wcur[0 * 320 + tid] = 0u; wcur[1 * 320 + tid] = 0u; wcur[2 * 320 + tid] = 0u; wcur[3 * 320 + tid] = 0u;
        const uint32_t n = n0 + n1 + n2 + n3;
        uint32_t all;
        const uint32_t first = block_exclusive_scan_alt<NT>(n, scratch, parity++, &all);
        cbase[0 * 256 + tid] = first; cbase[1 * 256 + tid] = first + n0; cbase[2 * 256 + tid] = first + n0 + n1; cbase[3 * 256 + tid] = first + n0 + n1 + n2;
        ccnt[tid] = n;
        if (P == 0u && a.lane_sort) {                                  // block-uniform: who walks which cell (see above)
            uint32_t bin = n;
            if (n >= 64u) { const uint32_t lg = 31u - (uint32_t)__clz((int)n); bin = 64u + min(63u, (lg - 6u) * 8u + ((n >> (lg - 3u)) & 7u)); }
            const uint32_t rank = atomicAdd(&phist[bin], 1u);
            __syncthreads();
            if (w == 0) {                                              // phist[b] -> cells with more records than bin b's
                const uint32_t v1 = phist[127 - 2 * lane], v0 = phist[126 - 2 * lane];
                const uint32_t incl = wave_inclusive_scan(v1 + v0), excl = incl - (v1 + v0);
                phist[127 - 2 * lane] = excl; phist[126 - 2 * lane] = excl + v1;
            }
            __syncthreads();
            perm[phist[bin] + rank] = (uint16_t)tid;
        }
        __syncthreads();
        lap(acc_base);
        // ---- 3. the batch in LDS, ordered by cell
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (in_batch(k, steps, nb)) {
                const uint32_t at = cbase[w * 256 + (key[k] & 255u)] + rk[k];
                st_hv[at] = hv[k];
                if constexpr (KEYED) st_sw[at] = (uint16_t)(key[k] >> a.id_bits);
                if constexpr (ATTR != 0) st_src[at] = src[k];
                if constexpr (PLAIN_OK) {
                    if (!records_plain) {
                        const float hh = __uint_as_float(hv[k].x), vv = __uint_as_float(hv[k].y);
                        odd = odd || !(fabsf(hh) <= kPlainHi) || !(vv >= kPlainLo) || !(vv <= kPlainHi);
                    }
                }
            }
        }
        if constexpr (PLAIN_OK) { if (odd) blk_odd = 1u; }            // (sticky: once a block has seen such a value it stays on the guarded loop)
        __syncthreads();
        lap(acc_place);
        // the next batch's records: in flight behind the chains
        if (P + (uint32_t)B < R) load_batch(P + (uint32_t)B, min(R - P - (uint32_t)B, (uint32_t)B));
        // ---- 4. the thread's cell: its records of this batch, in input order
        if (P == 0u) take_cell();
        const uint32_t cn = ccnt[c], cf = cbase[c];
        n_total += cn;
        const uint32_t nmax = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(cn), 63);
        const uint32_t lastp = cn ? cf + cn - 1u : 0u;
        uint2 nx = st_hv[min(cf, lastp)];
        uint32_t nx_sw = 0, nx_src = 0;
        if constexpr (KEYED) nx_sw = st_sw[min(cf, lastp)];
        if constexpr (ATTR != 0) nx_src = st_src[min(cf, lastp)];
        lap(acc_pre);
        bool plain = false;
        if constexpr (PLAIN_OK) plain = a.plain_env != 0 && blk_odd == 0u;   // block-uniform
        if (plain) {
            // ---- the PLAIN loop: straight-line, one rarely-taken branch per step
            if constexpr (PLAIN_OK) {
                plain_begin();
                // (the reads run past a lane's last record -- into the next cell's, or the tables behind the stage: LDS reads do not
                //  fault and a lane that is not `live` discards what it computes -- so the addresses are plain increments)
                const uint2* ph = st_hv + cf + 1u;
                const uint16_t* ps = st_sw + cf + 1u;
                // (tried, round 6: two steps per trip -- no register rotation, one pointer update per two steps: C4 95.2-96.7 -> 96.6-97.2 us)
                for (uint32_t i = 0; i < nmax; ++i) {                  // wave-uniform
                    const uint2 r = nx; const uint32_t swr = nx_sw;
                    nx = *ph++;
                    if constexpr (KEYED) nx_sw = *ps++;
                    plain_step(r, swr, i < cn);
                }
            }
        } else {
        if constexpr (HAS_VU) rp.refill(vu);                           // (the plain loop of an earlier batch keeps only rp.cur)
        for (uint32_t i = 0; i < nmax; ++i) {                          // wave-uniform
            const uint2 r = nx; const uint32_t sw = nx_sw, sr = nx_src;
            const uint32_t pn = min(cf + i + 1u, lastp);
            nx = st_hv[pn];
            if constexpr (KEYED) nx_sw = st_sw[pn];
            if constexpr (ATTR != 0) nx_src = st_src[pn];
            const bool live = i < cn;
            const float h = __uint_as_float(r.x), v = __uint_as_float(r.y);
            if constexpr (HAS_VU) rp.advance(cs, live ? sw : rp.cur, vu, a.var_floor);
            if constexpr (COUNT_SWEEPS) { if (live && sw != last_sweep) { ++sweeps_seen; last_sweep = sw; } }
            float e2 = ce, s2 = cs;
            const bool taken = fuse_step<true>(e2, s2, h, v, a.mahal, a.var_floor);
            const bool fl = live && (!LOWEST || h != -1.0f);           // GPU:482 (only LOWEST passes carry such records)
            ce = fl ? e2 : ce; cs = fl ? s2 : cs;
            if constexpr (LOWEST) { const float l2 = lowest_step(lw, h, v); lw = live ? l2 : lw; }
            if constexpr (ATTR != 0) { if (fl && taken && (sr & 0x80000000u)) wlast = sr & 0x7fffffffu; }
        }
        }
        lap(acc_walk);
        acc_nmax += nmax; ++n_batches;
        // (the next round's ranking touches the cursors and masks only; its stores into the stage come after two barriers -- unless
        //  it is a light round, which writes the stage and the counts at once)
        if constexpr (FAST_LIGHT) { if (a.light_fast) { ccnt[tid] = 0u; __syncthreads(); } }
    }
    if (dbg && tid == 0) {
        dbg[2] = acc_rank; dbg[3] = acc_base; dbg[4] = acc_place; dbg[5] = acc_walk; dbg[6] = __builtin_readcyclecounter();
        dbg[8] = n_batches; dbg[9] = acc_nmax; dbg[10] = acc_rare; dbg[11] = acc_pre;
    }
    if (dbg) {                                                         // per wave: chain steps it ran << 32 | records its lanes fused (lane use of the chains)
        const uint32_t recs = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(n_total), 63);
        if (lane == 0) dbg[12 + w] = ((unsigned long long)acc_nmax << 32) | recs;
    }
    if (__ballot(n_total != 0) == 0 && !a.dense) return;               // nothing reached this wave's cells and nothing is pending
    if constexpr (HAS_VU) { if (a.plain_env) rp.finish_plain(cs, last_sw, vu, a.var_floor); else rp.finish(cs, last_sw, vu, a.var_floor); }
    if (cs < a.var_floor) cs = a.var_floor;                            // GPU:533-534, on every cell

    if (owned) {                                                       // only what changed goes back
        if (__float_as_uint(ce) != __float_as_uint(e0)) a.elevation[g] = ce;
        if (__float_as_uint(cs) != __float_as_uint(s0)) a.variance[g] = cs;
        if constexpr (LOWEST) { if (__float_as_uint(lw) != __float_as_uint(lw0)) a.lowest[lgeo] = lw; }
        if constexpr (ATTR != 0) {
            if (wlast != 0xffffffffu) {                                // colour / intensity of the last taken point with all four non-zero (GPU:487-494)
                if (ATTR == 1) {
                    const uint32_t cc = a.rgb[wlast];
                    a.intensity[g] = a.xyzi[wlast].w;
                    a.colorR[g] = (int)((cc >> 16) & 0xff); a.colorG[g] = (int)((cc >> 8) & 0xff); a.colorB[g] = (int)(cc & 0xff);
                } else {
                    a.intensity[g] = a.f_I[wlast];
                    a.colorR[g] = a.f_R[wlast]; a.colorG[g] = a.f_G[wlast]; a.colorB[g] = a.f_B[wlast];
                }
            }
        }
    }
    if (a.counters) {                                                  // distinct touched cells: per pass, or summed over the sweeps
        const uint32_t mine = COUNT_SWEEPS ? sweeps_seen : (n_total ? 1u : 0u);
        const uint32_t s = wave_inclusive_scan(mine);
        if (lane == 63 && s) atomicAdd(&a.counters[1], (unsigned long long)s);
    }
}

// Where every block's records are in a device's BLOCK-sorted records: ranges[b] = {first record, one past the last}, from the per-block
// record counts k_sort_project left in blk_cnt: an exclusive prefix (the sorted order is the order of the block ids).  Used when the
// last pass's bins are not the blocks (maps of more than kOnePassMaxBins blocks); the multi-GPU strip owners get the ranges of every
// source with its records and never search.  (Before: a pass over the sorted keys looking for the places where the block id changes,
// 12 us + a 4 us memset for C5.)  One workgroup; the counts are zeroed on the way out -- the buffer is zero between passes, like the tile pipeline's tables.
__global__ __launch_bounds__(1024) void k_block_prefix(uint32_t* __restrict__ blk_cnt, int n_blocks, int seg, uint2* __restrict__ ranges)
{
    // through the LDS, a segment of `seg` blocks at a time: coalesced loads with many in flight, the prefix inside the LDS, coalesced
    // stores (one thread reading its 22 counts of the 2400^2 map one after the other from HBM took 38 us)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_sort[];
    __shared__ uint32_t scratch[16];
    uint32_t* lc = lds_sort;                                           // [seg + 1]
    const int tid = (int)threadIdx.x;
    uint32_t carry = 0;                                                // records in the segments before (block-uniform)
    for (int s0 = 0; s0 < n_blocks; s0 += seg) {                       // block-uniform
        const int n = min(seg, n_blocks - s0);
        for (int i = tid; i < n; i += 1024) { lc[i] = blk_cnt[s0 + i]; blk_cnt[s0 + i] = 0u; }
        __syncthreads();
        const int per = (n + 1023) / 1024, b0 = min(tid * per, n), b1 = min(b0 + per, n);
        uint32_t sum = 0;
        for (int b = b0; b < b1; ++b) sum += lc[b];
        uint32_t all;
        uint32_t at = carry + block_exclusive_scan<1024>(sum, scratch, &all);
        for (int b = b0; b < b1; ++b) { const uint32_t c = lc[b]; lc[b] = at; at += c; }
        if (tid == 0) lc[n] = carry + all;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) ranges[s0 + i] = make_uint2(lc[i], lc[i + 1]);
        carry += all;
        __syncthreads();
    }
}

static hipError_t lds_opt_in(const void* fn, size_t lds);

hipError_t launch_block_prefix(hipStream_t st, uint32_t* blk_cnt, int n_blocks, uint2* ranges)
{
    if (n_blocks <= 0) return hipSuccess;
    const int seg = n_blocks < 32768 ? n_blocks : 32768;               // 128 KB of LDS at most
    const size_t lds = ((size_t)seg + 1) * sizeof(uint32_t);
    const hipError_t e = lds_opt_in((const void*)k_block_prefix, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_block_prefix, dim3(1), dim3(1024), lds, st, blk_cnt, n_blocks, seg, ranges);
    return hipGetLastError();
}

// The boundaries of the tile-row strips in a device's sorted records (multi-GPU tiling): out[k] = first record whose cell id is
// >= ids[k].  One wave per boundary, the 32-ary search of k_fuse_walk over the whole array.
__global__ __launch_bounds__(64) void k_strip_bounds(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ n_records, int id_bits,
                                                     const uint32_t* __restrict__ ids, uint32_t* __restrict__ out)
{
    const int lane = (int)threadIdx.x;
    const uint32_t idmask = (1u << id_bits) - 1u, target = ids[blockIdx.x], l5 = (uint32_t)lane & 31u;
    uint32_t lo = 0, hi = *n_records;
    while (__ballot(lo < hi) != 0) {
        const uint32_t n = hi - lo, s = (n + 32u) / 33u;
        const uint32_t pos = lo + l5 * s + s - 1u;
        const bool probe = lane < 32 && lo < hi && pos < hi;
        const uint32_t id = probe ? (keys[pos] & idmask) : 0xffffffffu;
        const uint32_t k = (uint32_t)__popc((uint32_t)__ballot(probe && id < target));
        if (lo < hi) { const uint32_t nl = lo + k * s; if (k < 32u) hi = min(hi, nl + s - 1u); lo = nl; }
    }
    if (lane == 0) out[blockIdx.x] = lo;
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
#define GEM_LAUNCH(k, grid, block, lds, st, ev, ...)                                              \
    do {                                                                                           \
        if ((ev).start || (ev).stop) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)(lds), st, (ev).start, (ev).stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(k, grid, block, lds, st, __VA_ARGS__);                             \
    } while (0)

// Threads of the count / scatter workgroups of a pass with `bins` bins.  LDS of k_sort_scatter: 2 bins + 16 words + the larger of
// the ranking tables (rank_words) and the staged chunk (3 or 4 words per record).
SortShape sort_shape(int bins, bool attr, int chunk)
{
    SortShape s;
    const size_t stage = (size_t)(attr ? 4 : 3) * chunk;
    // eight waves while two workgroups of them fit a CU's LDS (the ranking tables of 1444 bins x 8 waves exceed the stage: 74 KB)
    s.nt = ((size_t)2 * bins + 16 + std::max(rank_words(8, bins), stage)) * 4 <= 80 * 1024 ? 512 : 256;
    s.chunk = chunk;
    s.lds = ((size_t)2 * bins + 16 + std::max(rank_words(s.nt / 64, bins), stage)) * 4;
    return s;
}

// more than 64 KiB of dynamic LDS needs an explicit opt-in, once per kernel and device
static hipError_t lds_opt_in(const void* fn, size_t lds)
{
    if (lds <= 64 * 1024) return hipSuccess;
    constexpr int kMaxDev = 64, kSlots = 32;
    static std::mutex mu;
    static const void* fns[kSlots] = {};
    static size_t configured[kMaxDev][kSlots] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    int slot = -1;
    for (int i = 0; i < kSlots; ++i) { if (fns[i] == fn) { slot = i; break; } if (!fns[i]) { fns[i] = fn; slot = i; break; } }
    if (dev < 0 || dev >= kMaxDev || slot < 0 || lds > configured[dev][slot]) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < kMaxDev && slot >= 0) configured[dev][slot] = lds;
    }
    return hipSuccess;
}

static hipError_t launch_project(hipStream_t st, const SortArgs& a, int src, LaunchEvents ev)
{
    const size_t lds = (size_t)a.dbins[0] * 4;
    if (a.chunk == kSortChunkSmall) {
        constexpr int C = kSortChunkSmall;
        if (src == 0)      GEM_LAUNCH((k_sort_project<0, C>), dim3(a.n_chunks1), dim3(256), lds, st, ev, a);
        else if (src == 2) GEM_LAUNCH((k_sort_project<2, C>), dim3(a.n_chunks1), dim3(256), lds, st, ev, a);
        else if (src == 4) GEM_LAUNCH((k_sort_project<4, C>), dim3(a.n_chunks1), dim3(256), lds, st, ev, a);
        else if (src == 3) GEM_LAUNCH((k_sort_project<3, C>), dim3(a.n_chunks1), dim3(256), lds, st, ev, a);
        else               GEM_LAUNCH((k_sort_project<1, C>), dim3(a.n_chunks1), dim3(256), lds, st, ev, a);
        return hipGetLastError();
    }
    constexpr int C = kSortChunk;
    if (src == 0)      GEM_LAUNCH((k_sort_project<0, C>), dim3(a.n_chunks1), dim3(256), lds, st, ev, a);
    else if (src == 2) GEM_LAUNCH((k_sort_project<2, C>), dim3(a.n_chunks1), dim3(256), lds, st, ev, a);
    else if (src == 4) GEM_LAUNCH((k_sort_project<4, C>), dim3(a.n_chunks1), dim3(256), lds, st, ev, a);
    else if (src == 3) GEM_LAUNCH((k_sort_project<3, C>), dim3(a.n_chunks1), dim3(256), lds, st, ev, a);
    else               GEM_LAUNCH((k_sort_project<1, C>), dim3(a.n_chunks1), dim3(256), lds, st, ev, a);
    return hipGetLastError();
}

template <int NT, bool ATTR, bool COHERENT, int CH>
static hipError_t launch_scatter(hipStream_t st, const PassArgs& p, int grid, size_t lds, LaunchEvents ev)
{
    const hipError_t e = lds_opt_in((const void*)k_sort_scatter<NT, ATTR, COHERENT, CH>, lds);
    if (e != hipSuccess) return e;
    GEM_LAUNCH((k_sort_scatter<NT, ATTR, COHERENT, CH>), dim3(grid), dim3(NT), lds, st, ev, p);
    return hipGetLastError();
}

template <int NT, int CH>
static hipError_t launch_pass_nt(hipStream_t st, const PassArgs& p, bool attr, bool coherent, int grid, bool count, size_t lds, LaunchEvents ev)
{
    if (count) {
        GEM_LAUNCH((k_sort_count<NT, CH>), dim3(grid), dim3(NT), (size_t)p.bins * 4, st, ev, p);
        return hipGetLastError();
    }
    if (attr) return coherent ? launch_scatter<NT, true, true, CH>(st, p, grid, lds, ev) : launch_scatter<NT, true, false, CH>(st, p, grid, lds, ev);
    return coherent ? launch_scatter<NT, false, true, CH>(st, p, grid, lds, ev) : launch_scatter<NT, false, false, CH>(st, p, grid, lds, ev);
}

static hipError_t launch_pass(hipStream_t st, const SortShape& sh, const PassArgs& p, bool attr, bool coherent, int grid, bool count, LaunchEvents ev)
{
    if (sh.chunk == kSortChunkSmall) {
        if (sh.nt == 512) return launch_pass_nt<512, kSortChunkSmall>(st, p, attr, coherent, grid, count, sh.lds, ev);
        return launch_pass_nt<256, kSortChunkSmall>(st, p, attr, coherent, grid, count, sh.lds, ev);
    }
    if (sh.nt == 512) return launch_pass_nt<512, kSortChunk>(st, p, attr, coherent, grid, count, sh.lds, ev);
    return launch_pass_nt<256, kSortChunk>(st, p, attr, coherent, grid, count, sh.lds, ev);
}

hipError_t launch_sort(hipStream_t st, const SortArgs& a, int src, bool attr, const LaunchEvents ev[9])
{
    if (a.n <= 0 || a.n_chunks1 <= 0 || a.n_passes < 1 || a.n_passes > 3 || (a.chunk != kSortChunk && a.chunk != kSortChunkSmall)) return hipErrorInvalidValue;
    SortShape sh[3];
    for (int i = 0; i < a.n_passes; ++i) { sh[i] = sort_shape(a.dbins[i], attr, a.chunk); if (sh[i].lds > 160 * 1024) return hipErrorInvalidValue; }
    hipError_t e;
    // ---- pass 1: project + count, scan, scatter by the lowest digit (input order in arrays a -> arrays b)
    if ((e = launch_project(st, a, src, ev[0])) != hipSuccess) return e;
    GEM_LAUNCH(k_sort_scan, dim3((a.dbins[0] + 63) / 64, kScanSegs), dim3(1024), 0, st, ev[1], a.cnt[0], a.segtot[0], a.dbins[0], a.n_chunks1,
               (const uint32_t*)nullptr, a.total, a.chunk);
    PassArgs p{};
    p.chunk = a.chunk;
    p.hv_in = a.hv_a; p.key_in = a.key_a; p.src_in = a.src_a; p.hv_out = a.hv_b; p.key_out = a.key_b; p.src_out = a.src_b;
    p.cnt = a.cnt[0]; p.segtot = a.segtot[0]; p.n_chunks = a.n_chunks1; p.bins = a.dbins[0]; p.shift = a.dshift[0]; p.digit_bits = a.dbits[0];
    p.mask = (1u << a.dbits[0]) - 1u;
    p.n_dev = nullptr; p.n_host = a.n; p.sweep_chunk0 = a.sweep_chunk0; p.sweep_first = a.sweep_first; p.n_sweeps = a.n_sweeps;
    p.seg_cnt = a.seg_cnt;
    const bool last0 = a.n_passes == 1;
    p.bin_base = last0 ? a.bin_base : nullptr; p.counters = last0 ? a.counters : nullptr;
    const bool coherent = a.dshift[0] >= 8 || a.rank_by_ballot;                          // block-sorted form: coarse digits, few bins per wave instruction
    p.coherent = coherent ? 1 : 0;
    p.few_bins = a.few_bins != 0 ? a.few_bins : (a.n_passes > 1 && a.dbits[0] <= 8 ? -1 : kFewBins);
    const bool fused = a.fuse_count != 0 && a.n_passes >= 2;
    if (fused) { p.next_cnt = a.cnt[1]; p.next_bins = a.dbins[1]; p.next_shift = a.dshift[1]; p.next_mask = (1u << a.dbits[1]) - 1u; }
    if ((e = launch_pass(st, sh[0], p, attr, coherent, a.n_chunks1, false, ev[2])) != hipSuccess) return e;
    p.next_cnt = nullptr;
    // ---- the higher digits: count, scan, scatter on the records of the pass before (ping-pong between the arrays); the
    //      live chunks are known on the device only
    const int grid = std::max(1, (int)((a.n + a.chunk - 1) / a.chunk));
    for (int i = 1; i < a.n_passes; ++i) {
        const bool from_b = (i & 1) != 0, last = i == a.n_passes - 1;
        p.hv_in = from_b ? a.hv_b : a.hv_a; p.key_in = from_b ? a.key_b : a.key_a; p.src_in = from_b ? a.src_b : a.src_a;
        p.hv_out = from_b ? a.hv_a : a.hv_b; p.key_out = from_b ? a.key_a : a.key_b; p.src_out = from_b ? a.src_a : a.src_b;
        p.cnt = a.cnt[i]; p.segtot = a.segtot[i]; p.n_chunks = 0; p.bins = a.dbins[i]; p.shift = a.dshift[i]; p.digit_bits = a.dbits[i];
        p.mask = (1u << a.dbits[i]) - 1u;
        p.n_dev = a.total; p.n_host = 0; p.sweep_chunk0 = nullptr; p.sweep_first = nullptr; p.n_sweeps = 1; p.seg_cnt = nullptr;
        p.bin_base = last ? a.bin_base : nullptr; p.counters = last ? a.counters : nullptr;
        p.few_bins = a.few_bins != 0 ? a.few_bins : kFewBins;
        if (!(fused && i == 1) && (e = launch_pass(st, sh[i], p, attr, coherent, grid, true, ev[3 * i])) != hipSuccess) return e;
        GEM_LAUNCH(k_sort_scan, dim3((a.dbins[i] + 63) / 64, kScanSegs), dim3(1024), 0, st, ev[3 * i + 1], a.cnt[i], a.segtot[i], a.dbins[i], 0,
                   (const uint32_t*)a.total, (uint32_t*)nullptr, a.chunk);
        if ((e = launch_pass(st, sh[i], p, attr, coherent, grid, false, ev[3 * i + 2])) != hipSuccess) return e;
    }
    return hipSuccess;
}

template <int FLAGS>
static hipError_t launch_walk_f(hipStream_t st, const WalkArgs& a, int mode, LaunchEvents ev)
{
    const dim3 grid(a.T * 4), block(kWalkNT);
    switch (mode) {
    case 0:  GEM_LAUNCH((k_fuse_walk<FLAGS, 0>), grid, block, 0, st, ev, a); break;
    case 1:  GEM_LAUNCH((k_fuse_walk<FLAGS, 1>), grid, block, 0, st, ev, a); break;
    case 2:  GEM_LAUNCH((k_fuse_walk<FLAGS, 2>), grid, block, 0, st, ev, a); break;
    default: GEM_LAUNCH((k_fuse_walk<FLAGS, 3>), grid, block, 0, st, ev, a); break;
    }
    return hipGetLastError();
}

template <int FLAGS, int MODE, int B, bool MULTI>
static hipError_t launch_block_walk_fmbm(hipStream_t st, const WalkArgs& a, LaunchEvents ev)
{
    constexpr bool KEYED = (MODE & 3) != 0, ATTR = (FLAGS & 3) != 0;
    const size_t lds = block_walk_lds<B, KEYED, ATTR>() + (size_t)std::max(0, a.lds_pad);      // (lds_pad: an experiment's way to cap the workgroups per CU)
    const hipError_t e = lds_opt_in((const void*)k_fuse_block<FLAGS, MODE, B, MULTI>, lds);
    if (e != hipSuccess) return e;
    GEM_LAUNCH((k_fuse_block<FLAGS, MODE, B, MULTI>), dim3(a.T * 4), dim3(kBlkNT), lds, st, ev, a);
    return hipGetLastError();
}

template <int FLAGS, int MODE, int B>
static hipError_t launch_block_walk_fmb(hipStream_t st, const WalkArgs& a, LaunchEvents ev)
{
    if constexpr (FLAGS == 0) { if (a.n_src > 1) return launch_block_walk_fmbm<FLAGS, MODE, B, true>(st, a, ev); }
    return launch_block_walk_fmbm<FLAGS, MODE, B, false>(st, a, ev);
}

#ifndef GEM_BLK_BATCH
#define GEM_BLK_BATCH 2048
#endif
constexpr int kBlkBatch = GEM_BLK_BATCH;   // records of a block staged in LDS per round: three workgroups per CU (4096: two per CU, half the rounds
                                  // for the blocks under the sensor -- whose chains stay as long; C4 104 -> 116 us per batch)

constexpr int kBlkBatchLight = 512;   // ... for passes whose blocks hold a few hundred records each (C5: 10 M points over 22 500 blocks): a quarter of the
                                      // ranking / staging steps per round and five workgroups per CU

template <int FLAGS>
static hipError_t launch_block_walk_f(hipStream_t st, const WalkArgs& a, int mode, LaunchEvents ev)
{
    if constexpr (FLAGS == 0) {
        if (a.light_blocks) {
            switch (mode) {
            case 0:  return launch_block_walk_fmb<FLAGS, 0, kBlkBatchLight>(st, a, ev);
            case 1:  return launch_block_walk_fmb<FLAGS, 1, kBlkBatchLight>(st, a, ev);
            case 2:  return launch_block_walk_fmb<FLAGS, 2, kBlkBatchLight>(st, a, ev);
            default: return launch_block_walk_fmb<FLAGS, 3, kBlkBatchLight>(st, a, ev);
            }
        }
    }
    switch (mode) {
    case 0:  return launch_block_walk_fmb<FLAGS, 0, kBlkBatch>(st, a, ev);
    case 1:  return launch_block_walk_fmb<FLAGS, 1, kBlkBatch>(st, a, ev);
    case 2:  return launch_block_walk_fmb<FLAGS, 2, kBlkBatch>(st, a, ev);
    default: return launch_block_walk_fmb<FLAGS, 3, kBlkBatch>(st, a, ev);
    }
}

hipError_t launch_block_walk(hipStream_t st, const WalkArgs& a, int flags, LaunchEvents ev)
{
    if (a.T <= 0) return hipSuccess;
    if (a.n_sweeps > kWalkMaxSweeps || a.n_src > kMaxRanks) return hipErrorInvalidValue;
    if (a.n_src > 1 && flags != 0) return hipErrorInvalidValue;       // records received from other ranks carry no colours / lowest scan points
    const int mode = (a.var_updates ? 1 : 0) | ((a.counters && !a.count_per_pass) ? 2 : 0);
    switch (flags) {
    case 0: return launch_block_walk_f<0>(st, a, mode, ev);
    case 1: return launch_block_walk_f<1>(st, a, mode, ev);
    case 2: return launch_block_walk_f<2>(st, a, mode, ev);
    case 4: return launch_block_walk_f<4>(st, a, mode, ev);
    case 5: return launch_block_walk_f<5>(st, a, mode, ev);
    case 6: return launch_block_walk_f<6>(st, a, mode, ev);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_strip_bounds(hipStream_t st, const uint32_t* keys, const uint32_t* n_records, int id_bits, const uint32_t* ids, uint32_t* out, int n)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_strip_bounds, dim3(n), dim3(64), 0, st, keys, n_records, id_bits, ids, out);
    return hipGetLastError();
}

hipError_t launch_walk(hipStream_t st, const WalkArgs& a, int flags, LaunchEvents ev)
{
    if (a.T <= 0) return hipSuccess;
    if (a.n_sweeps > kWalkMaxSweeps) return hipErrorInvalidValue;
    const int mode = (a.var_updates ? 1 : 0) | ((a.counters && !a.count_per_pass) ? 2 : 0);
    switch (flags) {
    case 0: return launch_walk_f<0>(st, a, mode, ev);
    case 1: return launch_walk_f<1>(st, a, mode, ev);
    case 2: return launch_walk_f<2>(st, a, mode, ev);
    case 4: return launch_walk_f<4>(st, a, mode, ev);
    case 5: return launch_walk_f<5>(st, a, mode, ev);
    case 6: return launch_walk_f<6>(st, a, mode, ev);
    default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------
// Input colourisation (EMg.cpp:349-381), the step in front of the path.  The reference loops over the points ON THE HOST: point
// i samples the BGR image at its pixel and then draws cv::circle(img, pixel, 1, sampled colour) INTO the image it samples from,
// so a later point whose pixel lies on an earlier point's circle gets that point's colour instead of the image's.  A radius-1,
// thickness-1 circle of OpenCV's Bresenham rasteriser (imgproc/src/drawing.cpp, Circle(): dx = 1, dy = 0 is its only round) is
// the four edge neighbours of the centre, clipped to the image; the centre itself is not drawn.
// Hence: colour(i) = colour(j) for the LATEST j < i that sampled a 4-neighbour of i's pixel, else the image at i's pixel -- a
// forest over the points whose roots read the untouched image.  With the records {pixel, point} sorted by pixel (the counting
// sort above, stable: ascending point index inside a pixel) the parent is four short searches, and the root comes from
// pointer jumping.
//   k_color_first   : per sorted record: pix[point] = pixel; first[pixel] = position of the pixel's first record
//   k_color_parent  : per point: the latest earlier point on a neighbouring pixel -> link[i] (itself for a root)
//   k_color_resolve : per point: root by path halving (every value ever stored in link[i] is an ancestor of i, so the racing
//                     updates of other threads, and stale reads of them, only shorten the way), colour from the image at the
//                     root's pixel; points outside the image get colour 0 and intensity 0 (EMg.cpp:372-377)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_color_first(ColorArgs a)
{
    const uint32_t M = *a.total;
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= M) return;
    const uint32_t key = a.key[s];
    a.pix[a.hv[s].x] = key;
    if (s == 0u || a.key[s - 1u] != key) a.first[key] = s;
}

__global__ __launch_bounds__(256) void k_color_parent(ColorArgs a)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint32_t)a.n) return;
    const uint32_t p = a.pix[i];
    uint32_t parent = i;
    if (p != 0xffffffffu) {
        const uint32_t M = *a.total;
        const int x = (int)(p % (uint32_t)a.width), y = (int)(p / (uint32_t)a.width);
        bool have = false;
        auto look = [&](int qx, int qy) {
            if (qx < 0 || qx >= a.width || qy < 0 || qy >= a.height) return;
            const uint32_t q = (uint32_t)(qy * a.width + qx);
            const uint32_t s0 = a.first[q];
            if (s0 == 0xffffffffu) return;
            // records of pixel q that come from points before i: a prefix of the pixel's run; galloping, then bisection
            auto before = [&](uint32_t s) { return s < M && a.key[s] == q && a.hv[s].x < i; };
            if (!before(s0)) return;
            uint32_t lo = s0, step = 1u, hi;                           // before(lo) holds, before(hi) does not
            for (;;) { hi = lo + step; if (!before(hi)) break; lo = hi; step <<= 1; }
            while (hi - lo > 1u) { const uint32_t mid = lo + ((hi - lo) >> 1); if (before(mid)) lo = mid; else hi = mid; }
            const uint32_t j = a.hv[lo].x;
            if (!have || j > parent) { parent = j; have = true; }
        };
        look(x - 1, y); look(x + 1, y); look(x, y - 1); look(x, y + 1);
    }
    a.link[i] = parent;
}

__global__ __launch_bounds__(256) void k_color_resolve(ColorArgs a)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (uint32_t)a.n) return;
    if (a.pix[i] == 0xffffffffu) {
        a.rgb[i] = 0u;
        reinterpret_cast<float*>(a.xyzi + i)[3] = 0.0f;
        return;
    }
    volatile uint32_t* link = a.link;
    uint32_t r = i;
    for (;;) {
        const uint32_t up = link[r];
        if (up == r) break;
        const uint32_t up2 = link[up];
        if (up2 != up) link[r] = up2;
        r = up2;
    }
    const uint32_t p = a.pix[r];
    const unsigned char* px = a.image + (size_t)(p / (uint32_t)a.width) * a.stride + (size_t)(p % (uint32_t)a.width) * 3u;
    a.rgb[i] = ((uint32_t)px[2] << 16) | ((uint32_t)px[1] << 8) | (uint32_t)px[0];
}

hipError_t launch_colorize(hipStream_t st, const ColorArgs& a)
{
    if (a.n <= 0) return hipSuccess;
    const dim3 grid((a.n + 255) / 256), block(256);
    hipLaunchKernelGGL(k_color_first, grid, block, 0, st, a);
    hipLaunchKernelGGL(k_color_parent, grid, block, 0, st, a);
    hipLaunchKernelGGL(k_color_resolve, grid, block, 0, st, a);
    return hipGetLastError();
}

} // namespace gem
