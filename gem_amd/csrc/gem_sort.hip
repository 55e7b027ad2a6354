// gem_sort.hip -- the SORTED pipeline of the GEM hot path for big passes (batches of sweeps, aggregated clouds, depth images).
//
// G_fuse (gpu_process.cu:477-537) is one thread per CELL scanning all points in input order; the recurrence is not
// associative (variance floor inside the loop, Mahalanobis branch), so what every cell needs is ITS points, in input order.
// For a stream of single LiDAR sweeps that list is short and k_frame (gem_kernels.hip) builds it per tile in LDS.  For a big
// pass -- millions of points, tens of sweeps -- the lists are long, the tile under the sensor carries a hundred times the
// records of a tile at the rim, and any per-tile batching is bound by that one tile.  Here the lists are built by the whole
// chip instead: a stable two-digit LSD counting sort of the in-map points by (tile, cell-in-tile), every pass split into
// equal chunks, and then every cell walks its own contiguous run:
//
//   k_sort_count1    chunk of 8192 points: project + bin (G_pointsprocess, GPU:384-455), histogram over the LOW digit
//                    (cell inside its 32x32 tile, 1024 bins) in LDS                      -> cnt1[chunk][1024]
//   k_sort_scan      column-wise exclusive prefix over the chunks, column totals          -> cnt1 (in place), tot1[1024], M
//   k_sort_scatter1  the same chunk again: project, STABLE rank inside the chunk (wave w owns the w-th contiguous share,
//                    64 consecutive points per step, equal bins matched by ballots, a per-wave cursor per bin in LDS),
//                    record {h, var} + key {cell | tile | sweep} written to its final place of pass 1
//   k_sort_count2 / k_sort_scan / k_sort_scatter2   the same over the HIGH digit (tile) on the records of pass 1
//   k_fuse_walk      one workgroup per tile, one thread per cell: cell boundaries of the tile's run from one look at the keys,
//                    then every thread streams its own run through the reference's recurrence (GPU:480-531), the variance
//                    increments of the sweeps in between (GPU:540-547) and the floors (GPU:533-534) replayed per cell.
//
// Stability of both passes keeps ascending input order inside every cell; no float atomics, no LDS batches, no fast / generic /
// dense cases: the tile takes the time of its LONGEST cell chain, which is the floor of any exact implementation.
// Algorithmic bytes: 16 B per point (read) + 16 B per distinct touched cell (+ 8 L^2 per dense variance pass).  What the
// sort moves on top of that is stated in DESIGN.md section 4.
//
// Built with -ffp-contract=off (see gem_device.hpp).
#include "gem_kernels.hpp"
#include "gem_wave.hpp"

#include <hip/hip_ext.h>

#include <mutex>

namespace gem {

constexpr int kSortK = 8;                       // items per thread and chunk

// ------------------------------------------------------------------------------------------
// one point of the pass: projection + binning, the same decisions as bin_wave_body (gem_kernels.hip)
// ------------------------------------------------------------------------------------------
struct Binned { bool valid; uint32_t cell, tile; float h, v; bool colour_ok; };

template <int SRC, int TS>
__device__ __forceinline__ Binned bin_one(const SortArgs& a, const FrameConst& fc, bool live, const float4& p, long long i, int orig_fallback)
{
    constexpr int TE = 1 << TS;
    Binned b; b.valid = false; b.cell = 0; b.tile = 0; b.h = 0.0f; b.v = 0.0f; b.colour_ok = false;
    if (!live) return b;
    int row, col; float h, v; bool colour_ok = false;
    if (SRC == 0) {
        const Projected r = project_point(fc, p.x, p.y, p.z, a.orig ? a.orig[i] : orig_fallback);
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
        b.tile = (uint32_t)((row >> TS) * a.tiles_per_row + (col >> TS));
        b.cell = (uint32_t)(((row & (TE - 1)) << TS) | (col & (TE - 1)));
        b.h = h; b.v = v; b.colour_ok = colour_ok;
    }
    return b;
}

// chunk -> (sweep, first point of the chunk, end of the sweep): chunks never span sweeps (the frame constants differ)
struct ChunkRange { int sweep; long long first, end; int orig0; };

template <bool BATCH, int CH>
__device__ __forceinline__ ChunkRange chunk_range(const SortArgs& a, int chunk)
{
    ChunkRange r; r.sweep = 0; r.first = (long long)chunk * CH; r.end = a.n; r.orig0 = 0;
    if (BATCH) {
        int lo = 0, hi = a.n_sweeps;                                   // block-uniform: scalar loads
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.sweep_chunk0[mid] <= chunk) lo = mid; else hi = mid; }
        r.sweep = lo;
        const long long sb = a.sweep_first[lo];
        r.first = sb + (long long)(chunk - a.sweep_chunk0[lo]) * CH;
        r.end = a.sweep_first[lo + 1];
        r.orig0 = (a.sweep_orig0 ? a.sweep_orig0[lo] : 0) - (int)sb;   // original index of point i = i + orig0
    }
    return r;
}

// ------------------------------------------------------------------------------------------
// pass 1, count
// ------------------------------------------------------------------------------------------
template <int SRC, int TS, bool BATCH>
__global__ __launch_bounds__(1024) void k_sort_count1(SortArgs a)
{
    constexpr int NT = 1024, K = kSortK, CH = NT * K, D0 = 1 << (2 * TS);
    __shared__ uint32_t hist[D0];
    const int tid = (int)threadIdx.x, chunk = (int)blockIdx.x;
    for (int i = tid; i < D0; i += NT) hist[i] = 0u;
    if (chunk == 0 && tid == 0) *a.total = 0u;                         // k_sort_scan of this pass adds the column totals up
    const ChunkRange cr = chunk_range<BATCH, CH>(a, chunk);
    const FrameConst& fc = BATCH ? a.frames[cr.sweep] : a.frame0;
    const long long base = cr.first + (long long)(tid >> 6) * (K * 64) + (tid & 63);
    float4 p[K]; bool live[K];
    if (SRC == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {                                  // all loads in flight before the first projection
            const long long i = base + k * 64;
            live[k] = i < cr.end;
            p[k] = a.xyzi[live[k] ? i : cr.first];
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) { live[k] = base + k * 64 < cr.end; p[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const long long i = base + k * 64;
        const Binned b = bin_one<SRC, TS>(a, fc, live[k], p[k], i, (int)i + cr.orig0);
        if (b.valid) atomicAdd(&hist[b.cell], 1u);
    }
    __syncthreads();
    for (int i = tid; i < D0; i += NT) a.cnt1[(size_t)chunk * D0 + i] = hist[i];
}

// ------------------------------------------------------------------------------------------
// column scan over the chunks (both passes): cnt[c][b] -> sum over c' < c of cnt[c'][b], in place; tot[b] = column sum.
// One workgroup per 64 bins; wave w owns the w-th contiguous share of the chunks (lane = bin: 256-byte coalesced rows).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_sort_scan(uint32_t* __restrict__ cnt, uint32_t* __restrict__ tot, int bins, int n_chunks,
                                                    const uint32_t* __restrict__ records, int chunk_records, uint32_t* __restrict__ total_out)
{
    __shared__ uint32_t part[16][64];
    const int lane = lane_id(), w = (int)(threadIdx.x >> 6);
    const int b = (int)blockIdx.x * 64 + lane;
    // pass 2: the number of chunks depends on how many records pass 1 kept (known on the device only)
    const int nc = records ? (int)(((unsigned long long)*records + (unsigned)chunk_records - 1u) / (unsigned)chunk_records) : n_chunks;
    const int S = (nc + 15) >> 4;
    const int c_lo = min(nc, w * S), c_hi = min(nc, c_lo + S);
    uint32_t sum = 0;
    if (b < bins) {
        int c = c_lo;
        for (; c + 4 <= c_hi; c += 4) {                                // four independent loads in flight
            const uint32_t v0 = cnt[(size_t)c * bins + b], v1 = cnt[(size_t)(c + 1) * bins + b];
            const uint32_t v2 = cnt[(size_t)(c + 2) * bins + b], v3 = cnt[(size_t)(c + 3) * bins + b];
            sum += (v0 + v1) + (v2 + v3);
        }
        for (; c < c_hi; ++c) sum += cnt[(size_t)c * bins + b];
    }
    part[w][lane] = sum;
    __syncthreads();
    uint32_t run = 0, all = 0;
#pragma unroll
    for (int ww = 0; ww < 16; ++ww) { const uint32_t v = part[ww][lane]; run += ww < w ? v : 0u; all += v; }
    if (b < bins) {
        for (int c = c_lo; c < c_hi; ++c) {
            const uint32_t v = cnt[(size_t)c * bins + b];
            cnt[(size_t)c * bins + b] = run;
            run += v;
        }
        if (w == 0) tot[b] = all;
    }
    if (total_out && w == 0) {
        const uint32_t s = wave_inclusive_scan(b < bins ? all : 0u);
        if (lane == 63 && s) atomicAdd(total_out, s);
    }
}

// exclusive scan of src[0 .. n) (global) into dst[0 .. n) (LDS) by an NT-thread block; *total = the sum
template <int NT>
__device__ __forceinline__ void block_scan_array(const uint32_t* __restrict__ src, uint32_t* dst, int n, uint32_t* scratch, uint32_t* total)
{
    const int tid = (int)threadIdx.x;
    const int per = (n + NT - 1) / NT;
    const int b0 = tid * per;
    uint32_t sum = 0;
    for (int j = 0; j < per; ++j) if (b0 + j < n) sum += src[b0 + j];
    uint32_t ex = block_exclusive_scan<NT>(sum, scratch, total);
    for (int j = 0; j < per; ++j) if (b0 + j < n) { const uint32_t v = src[b0 + j]; dst[b0 + j] = ex; ex += v; }
}

// Stable rank of this lane's item among the items of the same bin that precede it in the wave's share of the chunk:
// equal bins inside one step are matched by ballots, the wave's cursor of the bin (LDS, private to the wave: its LDS
// operations execute in order) carries the count from step to step.
__device__ __forceinline__ uint32_t wave_rank_step(bool valid, uint32_t bin, int nbits, uint32_t* wcur, uint64_t lt)
{
    const uint64_t peers = wave_peers(valid, bin, nbits);
    const uint32_t rank = (uint32_t)__popcll(peers & lt);
    uint32_t old = 0;
    if (valid && rank == 0) old = atomicAdd(&wcur[bin], (uint32_t)__popcll(peers));
    old = (uint32_t)__shfl((int)old, valid ? __ffsll((unsigned long long)peers) - 1 : lane_id(), 64);
    return old + rank;
}

// ------------------------------------------------------------------------------------------
// pass 1, scatter: the chunk's records go to  base1[bin] + (records of the bin in earlier chunks) + (stable rank in the chunk)
// ------------------------------------------------------------------------------------------
template <int SRC, int TS, bool ATTR, bool BATCH>
__global__ __launch_bounds__(1024) void k_sort_scatter1(SortArgs a)
{
    constexpr int NT = 1024, NW = NT / 64, K = kSortK, CH = NT * K, D0 = 1 << (2 * TS);
    static_assert(D0 == NT, "one low-digit bin per thread");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_sort[];
    uint32_t* wcnt = lds_sort;                                         // [NW][D0]
    uint32_t* scratch = wcnt + NW * D0;                                // [16]
    const int tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6, chunk = (int)blockIdx.x;
    const uint64_t lt = lanemask_lt();
    for (int i = tid; i < NW * D0; i += NT) wcnt[i] = 0u;
    const ChunkRange cr = chunk_range<BATCH, CH>(a, chunk);
    const FrameConst& fc = BATCH ? a.frames[cr.sweep] : a.frame0;
    const long long base = cr.first + (long long)w * (K * 64) + lane;
    float4 p[K]; bool live[K];
    if (SRC == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long long i = base + k * 64;
            live[k] = i < cr.end;
            p[k] = a.xyzi[live[k] ? i : cr.first];
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) { live[k] = base + k * 64 < cr.end; p[k] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    const uint32_t colpre = a.cnt1[(size_t)chunk * D0 + tid];          // records of bin `tid` in earlier chunks (k_sort_scan)
    const uint32_t tot = a.tot1[tid];
    __syncthreads();
    // ---- 1. stable rank inside the wave's share, per-wave counts
    uint32_t key[K], rk[K]; float hh[K], vv[K]; bool ok[K], cok[K];
    uint32_t* wcur = wcnt + w * D0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const long long i = base + k * 64;
        const Binned b = bin_one<SRC, TS>(a, fc, live[k], p[k], i, (int)i + cr.orig0);
        ok[k] = b.valid; hh[k] = b.h; vv[k] = b.v; cok[k] = b.colour_ok;
        key[k] = b.cell | (b.tile << (2 * TS)) | ((uint32_t)cr.sweep << a.sweep_shift);
        rk[k] = wave_rank_step(b.valid, b.cell, 2 * TS, wcur, lt);
    }
    __syncthreads();
    // ---- 2. bin `tid`: global base + earlier chunks, then the waves of this chunk in order
    {
        uint32_t all;
        uint32_t g = block_exclusive_scan<NT>(tot, scratch, &all) + colpre;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) { const uint32_t c = wcnt[ww * D0 + tid]; wcnt[ww * D0 + tid] = g; g += c; }
    }
    __syncthreads();
    // ---- 3. records to their places
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (ok[k]) {
            const uint32_t cell = key[k] & (uint32_t)(D0 - 1);
            const uint32_t pos = wcur[cell] + rk[k];
            a.hv1[pos] = make_uint2(__float_as_uint(hh[k]), __float_as_uint(vv[k]));
            a.key1[pos] = key[k];
            if (ATTR) a.src1[pos] = (uint32_t)(base + k * 64) | (cok[k] ? 0x80000000u : 0u);   // source point; bit 31: all of R, G, B, intensity non-zero
        }
    }
}

// ------------------------------------------------------------------------------------------
// pass 2 (high digit = tile) over the records of pass 1
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void k_sort_count2(SortArgs a)
{
    constexpr int K = kSortK, CH = NT * K;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_sort[];
    uint32_t* hist = lds_sort;                                         // [T]
    const int tid = (int)threadIdx.x, chunk = (int)blockIdx.x;
    const uint32_t M = *a.total;
    if ((unsigned long long)chunk * CH >= M) return;                   // k_sort_scan only reads the rows of live chunks
    for (int i = tid; i < a.T; i += NT) hist[i] = 0u;
    const uint32_t base = (uint32_t)chunk * CH + (uint32_t)(tid >> 6) * (K * 64) + (uint32_t)(tid & 63);
    uint32_t key[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { const uint32_t i = base + k * 64; key[k] = a.key1[i < M ? i : M - 1u]; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) if (base + k * 64 < M) atomicAdd(&hist[(key[k] >> a.cell_bits) & a.tile_mask], 1u);
    __syncthreads();
    for (int i = tid; i < a.T; i += NT) a.cnt2[(size_t)chunk * a.T + i] = hist[i];
}

template <int NT, bool ATTR>
__global__ __launch_bounds__(NT) void k_sort_scatter2(SortArgs a)
{
    constexpr int NW = NT / 64, K = kSortK, CH = NT * K;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_sort[];
    const int T = a.T;
    uint32_t* gbase = lds_sort;                                        // [T]     first record of every tile
    uint32_t* wcnt = gbase + T;                                        // [NW][T]
    uint32_t* scratch = wcnt + NW * T;                                 // [16]
    const int tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6, chunk = (int)blockIdx.x;
    const uint64_t lt = lanemask_lt();
    const uint32_t M = *a.total;
    const bool live_chunk = (unsigned long long)chunk * CH < M;
    if (!live_chunk && chunk != 0) return;                             // workgroup 0 always publishes the tile bases
    for (int i = tid; i < NW * T; i += NT) wcnt[i] = 0u;
    const uint32_t base = (uint32_t)chunk * CH + (uint32_t)w * (K * 64) + (uint32_t)lane;
    uint2 hv[K]; uint32_t key[K], src[K], rk[K];
    if (live_chunk) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t i = base + k * 64, ic = i < M ? i : M - 1u;
            key[k] = a.key1[ic]; hv[k] = a.hv1[ic];
            if (ATTR) src[k] = a.src1[ic];
        }
    }
    __syncthreads();
    if (live_chunk) {
        uint32_t* wcur = wcnt + w * T;
#pragma unroll
        for (int k = 0; k < K; ++k)
            rk[k] = wave_rank_step(base + k * 64 < M, (key[k] >> a.cell_bits) & a.tile_mask, a.tile_bits, wcur, lt);
    }
    __syncthreads();
    {
        uint32_t all;
        block_scan_array<NT>(a.tot2, gbase, T, scratch, &all);
        __syncthreads();
        if (chunk == 0) {
            for (int i = tid; i < T; i += NT) a.tile_base[i] = gbase[i];
            if (tid == 0) { a.tile_base[T] = all; if (a.counters) atomicAdd(&a.counters[0], (unsigned long long)all); }
        }
        if (live_chunk) {
            for (int b = tid; b < T; b += NT) {
                uint32_t g = gbase[b] + a.cnt2[(size_t)chunk * T + b];
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) { const uint32_t c = wcnt[ww * T + b]; wcnt[ww * T + b] = g; g += c; }
            }
        }
    }
    __syncthreads();
    if (live_chunk) {
        const uint32_t* wcur = wcnt + w * T;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            if (base + k * 64 < M) {
                const uint32_t pos = wcur[(key[k] >> a.cell_bits) & a.tile_mask] + rk[k];
                a.hv2[pos] = hv[k];
                a.key2[pos] = key[k];
                if (ATTR) a.src2[pos] = src[k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_fuse_walk : one workgroup per 32x32 tile, one thread per cell
// ------------------------------------------------------------------------------------------
// FLAGS: bits 0-1 = ATTR (0 none, 1 colours from the cloud, 2 colours from gem_fuse's arrays), bit 2 = LOWEST (also maintain
// map_lowest, GPU:432-439).  MODE: bit 0 = variance increments between the sweeps (batched call with var_updates), bit 1 =
// count the touched cells per sweep (statistics).
constexpr int kWalkMaxSweeps = 512;

template <int TS, int FLAGS, int MODE>
__global__ __launch_bounds__(1024) void k_fuse_walk(WalkArgs a)
{
    constexpr int ATTR = FLAGS & 3;
    constexpr bool LOWEST = (FLAGS & 4) != 0;
    constexpr bool HAS_VU = (MODE & 1) != 0, COUNT_SWEEPS = (MODE & 2) != 0, KEYED = HAS_VU || COUNT_SWEEPS;
    constexpr int TE = 1 << TS, CELLS = TE * TE, NT = 1024;
    static_assert(CELLS == NT, "one cell per thread");
    __shared__ uint32_t cstart[CELLS], cend[CELLS];
    __shared__ float vu[HAS_VU ? kWalkMaxSweeps : 1];
    __shared__ uint32_t n_touched;
    const int tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6;
    const int tile = (int)blockIdx.x;
    const uint32_t rb = a.tile_base[tile], re = a.tile_base[tile + 1];
    if (rb == re && !a.dense) return;                                  // nothing reaches this tile and nothing is pending
    const int tr = tile / a.tiles_per_row, tc = tile - tr * a.tiles_per_row;
    const int row = (tr << TS) + (tid >> TS), col = (tc << TS) + (tid & (TE - 1));
    const int L = a.L;
    const bool owned = row >= a.row0 && row < a.row1 && col < L;
    const size_t g = owned ? (size_t)row * L + col : 0;
    const float e0 = a.elevation[g], s0 = a.variance[g];               // in flight behind the boundary search
    size_t lgeo = 0; float lw = 0.0f, lw0 = 0.0f;
    if constexpr (LOWEST) {                                            // map_lowest is indexed by the GEOGRAPHIC cell (GPU:430)
        int gr = row - a.start0, gc = col - a.start1;
        gr += gr < 0 ? L : 0; gc += gc < 0 ? L : 0;
        lgeo = owned ? (size_t)gr * L + gc : 0;
        lw0 = lw = a.lowest[lgeo];
    }
    cstart[tid] = 0u; cend[tid] = 0u;
    if (tid == 0) n_touched = 0u;
    if constexpr (HAS_VU) for (int i = tid; i < a.n_sweeps; i += NT) vu[i] = a.var_updates[i];
    __syncthreads();
    // ---- cell boundaries of the tile's run: the keys are sorted by cell, so a cell starts where the key's cell changes
    for (uint32_t p0 = rb + (uint32_t)w * 64u; p0 < re; p0 += NT) {    // wave-uniform
        const uint32_t p = p0 + (uint32_t)lane;
        const bool live = p < re;
        const uint32_t cell = live ? (a.key[p] & (uint32_t)(CELLS - 1)) : 0xfffffffeu;
        uint32_t prev = (uint32_t)__shfl_up((int)cell, 1, 64), next = (uint32_t)__shfl_down((int)cell, 1, 64);
        if (lane == 0)  prev = p > rb ? (a.key[p - 1] & (uint32_t)(CELLS - 1)) : 0xffffffffu;
        if (lane == 63) next = p + 1u < re ? (a.key[p + 1] & (uint32_t)(CELLS - 1)) : 0xffffffffu;
        if (live && cell != prev) cstart[cell] = p;
        if (live && cell != next) cend[cell] = p + 1u;
    }
    __syncthreads();
    const uint32_t first = cstart[tid], n = cend[tid] - first;

    float ce = e0, cs = s0;
    // Mapvar_update increments queued before this pass, then the one of sweep 0 (GPU:540-547)
    for (int k = 0; k < a.n_pending; ++k) if (cs != kInitVariance) cs += a.pending[k];
    uint32_t cur = 0;                                                  // "inside sweep cur, its increment applied"
    if constexpr (HAS_VU) { if (cs != kInitVariance) cs += vu[0]; }
    // from sweep `cur` to sweep `to`: the floor that ends every Fuse (GPU:533-534), then the next sweep's increment
    auto advance = [&](uint32_t to) {
        while (cur < to) {
            if (cs < a.var_floor) cs = a.var_floor;
            ++cur;
            if (cs != kInitVariance) cs += vu[cur];
        }
    };
    uint32_t wlast = 0xffffffffu, sweeps_seen = 0, last_sweep = 0xffffffffu;
    {   // the cell's own run, the records D steps ahead in flight (clamped address: never a branch round a load)
        constexpr int D = 4;
        const uint32_t nm1 = n ? n - 1u : 0u;
        const uint2* hp = a.hv + (n ? first : 0u);
        const uint32_t* kp = a.key + (n ? first : 0u);
        const uint32_t* sp = a.src + (n ? first : 0u);
        uint2 pre_hv[D]; uint32_t pre_k[D], pre_s[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const uint32_t j = min((uint32_t)k, nm1);
            pre_hv[k] = hp[j];
            if (KEYED) pre_k[k] = kp[j];
            if (ATTR) pre_s[k] = sp[j];
        }
        for (uint32_t i = 0; __ballot(i < n) != 0; i += D) {           // wave-uniform
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const uint2 cur_hv = pre_hv[k];
                const uint32_t cur_k = KEYED ? pre_k[k] : 0u, cur_s = ATTR ? pre_s[k] : 0u;
                const uint32_t idx = i + (uint32_t)k;
                const uint32_t j = min(idx + D, nm1);
                pre_hv[k] = hp[j];
                if (KEYED) pre_k[k] = kp[j];
                if (ATTR) pre_s[k] = sp[j];
                const bool live = idx < n;
                const float h = __uint_as_float(cur_hv.x), v = __uint_as_float(cur_hv.y);
                if constexpr (KEYED) {
                    const uint32_t sw = cur_k >> a.sweep_shift;
                    if constexpr (HAS_VU) { if (live) advance(sw); }
                    if constexpr (COUNT_SWEEPS) { if (live && sw != last_sweep) { ++sweeps_seen; last_sweep = sw; } }
                }
                float e2 = ce, s2 = cs;
                const bool taken = fuse_step(e2, s2, h, v, a.mahal, a.var_floor);
                const bool fl = live && (!LOWEST || h != -1.0f);       // GPU:482 (only LOWEST passes carry such records)
                ce = fl ? e2 : ce; cs = fl ? s2 : cs;
                if constexpr (LOWEST) { const float l2 = lowest_step(lw, h, v); lw = live ? l2 : lw; }
                if constexpr (ATTR != 0) { if (fl && taken && (cur_s & 0x80000000u)) wlast = cur_s & 0x7fffffffu; }
            }
        }
    }
    if constexpr (HAS_VU) advance((uint32_t)a.n_sweeps - 1u);
    if (cs < a.var_floor) cs = a.var_floor;                            // GPU:533-534, on every cell

    if (owned) {
        // only what changed goes back (a sweep touches a fraction of a tile's cells; whole-tile write-backs were most of the
        // write traffic of the tile kernels)
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
        const uint32_t mine = COUNT_SWEEPS ? sweeps_seen : (n ? 1u : 0u);
        const uint32_t s = wave_inclusive_scan(mine);
        if (lane == 63 && s) atomicAdd(&n_touched, s);
        __syncthreads();
        if (tid == 0 && n_touched) atomicAdd(&a.counters[1], (unsigned long long)n_touched);
    }
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
#define GEM_LAUNCH(k, grid, block, lds, st, ev, ...)                                              \
    do {                                                                                           \
        if ((ev).start) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)(lds), st, (ev).start, (ev).stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(k, grid, block, lds, st, __VA_ARGS__);                             \
    } while (0)

int sort_pass2_threads(int T)
{
    // LDS of k_sort_scatter2 = (T + T * NT / 64 + 16) words: 1024 threads while two workgroups fit a CU, fewer waves for big maps
    if (T <= 900) return 1024;
    if (T <= 1800) return 512;
    return 256;
}

static size_t scatter2_lds(int T, int nt) { return ((size_t)T * (1 + nt / 64) + 16) * 4; }

// more than 64 KiB of dynamic LDS needs an explicit opt-in, once per kernel and device
static hipError_t lds_opt_in(const void* fn, size_t lds, int slot)
{
    if (lds <= 64 * 1024) return hipSuccess;
    constexpr int kMaxDev = 64, kSlots = 8;
    static std::mutex mu;
    static size_t configured[kMaxDev][kSlots] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    if (dev < 0 || dev >= kMaxDev || lds > configured[dev][slot]) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < kMaxDev) configured[dev][slot] = lds;
    }
    return hipSuccess;
}

template <int SRC, bool ATTR, bool BATCH>
static hipError_t launch_sort_pass1(hipStream_t st, const SortArgs& a, LaunchEvents ev_count, LaunchEvents ev_scan, LaunchEvents ev_scatter)
{
    constexpr int TS = 5, D0 = 1 << (2 * TS);
    GEM_LAUNCH((k_sort_count1<SRC, TS, BATCH>), dim3(a.n_chunks1), dim3(1024), 0, st, ev_count, a);
    GEM_LAUNCH(k_sort_scan, dim3(D0 / 64), dim3(1024), 0, st, ev_scan, a.cnt1, a.tot1, D0, a.n_chunks1, (const uint32_t*)nullptr, 0, a.total);
    const size_t lds = ((size_t)16 * D0 + 16) * 4;
    hipError_t e = lds_opt_in((const void*)k_sort_scatter1<SRC, TS, ATTR, BATCH>, lds, (SRC ? 1 : 0) | (ATTR ? 2 : 0) | (BATCH ? 4 : 0));
    if (e != hipSuccess) return e;
    GEM_LAUNCH((k_sort_scatter1<SRC, TS, ATTR, BATCH>), dim3(a.n_chunks1), dim3(1024), lds, st, ev_scatter, a);
    return hipGetLastError();
}

template <int NT, bool ATTR>
static hipError_t launch_sort_pass2(hipStream_t st, const SortArgs& a, LaunchEvents ev_count, LaunchEvents ev_scan, LaunchEvents ev_scatter)
{
    constexpr int CH = NT * kSortK;
    const int grid = (int)((a.n + CH - 1) / CH);                       // upper bound: the live chunks are known on the device only
    hipError_t e = lds_opt_in((const void*)k_sort_count2<NT>, (size_t)a.T * 4, 0);
    if (e != hipSuccess) return e;
    GEM_LAUNCH((k_sort_count2<NT>), dim3(grid), dim3(NT), (size_t)a.T * 4, st, ev_count, a);
    GEM_LAUNCH(k_sort_scan, dim3((a.T + 63) / 64), dim3(1024), 0, st, ev_scan, a.cnt2, a.tot2, a.T, 0, (const uint32_t*)a.total, CH, (uint32_t*)nullptr);
    const size_t lds = scatter2_lds(a.T, NT);
    e = lds_opt_in((const void*)k_sort_scatter2<NT, ATTR>, lds, ATTR ? 1 : 0);
    if (e != hipSuccess) return e;
    GEM_LAUNCH((k_sort_scatter2<NT, ATTR>), dim3(grid > 0 ? grid : 1), dim3(NT), lds, st, ev_scatter, a);
    return hipGetLastError();
}

hipError_t launch_sort(hipStream_t st, const SortArgs& a, int src, bool attr, const LaunchEvents ev[6])
{
    if (a.n <= 0 || a.n_chunks1 <= 0) return hipErrorInvalidValue;
    const bool batch = a.n_sweeps > 1;
    hipError_t e;
#define GEM_P1(S, A, B) launch_sort_pass1<S, A, B>(st, a, ev[0], ev[1], ev[2])
    if (src == 0) e = attr ? (batch ? GEM_P1(0, true, true) : GEM_P1(0, true, false)) : (batch ? GEM_P1(0, false, true) : GEM_P1(0, false, false));
    else          e = attr ? (batch ? GEM_P1(1, true, true) : GEM_P1(1, true, false)) : (batch ? GEM_P1(1, false, true) : GEM_P1(1, false, false));
#undef GEM_P1
    if (e != hipSuccess) return e;
    const int nt = sort_pass2_threads(a.T);
    if (scatter2_lds(a.T, nt) > 160 * 1024) return hipErrorInvalidValue;
    if (nt == 1024) return attr ? launch_sort_pass2<1024, true>(st, a, ev[3], ev[4], ev[5]) : launch_sort_pass2<1024, false>(st, a, ev[3], ev[4], ev[5]);
    if (nt == 512)  return attr ? launch_sort_pass2<512, true>(st, a, ev[3], ev[4], ev[5]) : launch_sort_pass2<512, false>(st, a, ev[3], ev[4], ev[5]);
    return attr ? launch_sort_pass2<256, true>(st, a, ev[3], ev[4], ev[5]) : launch_sort_pass2<256, false>(st, a, ev[3], ev[4], ev[5]);
}

template <int FLAGS>
static hipError_t launch_walk_f(hipStream_t st, const WalkArgs& a, int mode, LaunchEvents ev)
{
    const dim3 grid(a.T), block(1024);
    switch (mode) {
    case 0:  GEM_LAUNCH((k_fuse_walk<5, FLAGS, 0>), grid, block, 0, st, ev, a); break;
    case 1:  GEM_LAUNCH((k_fuse_walk<5, FLAGS, 1>), grid, block, 0, st, ev, a); break;
    case 2:  GEM_LAUNCH((k_fuse_walk<5, FLAGS, 2>), grid, block, 0, st, ev, a); break;
    default: GEM_LAUNCH((k_fuse_walk<5, FLAGS, 3>), grid, block, 0, st, ev, a); break;
    }
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

} // namespace gem
