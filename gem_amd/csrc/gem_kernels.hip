// gem_kernels.hip -- gfx950 kernels of the GEM point-cloud -> elevation-grid hot path.
//
// Pipeline for one pass (replaces G_pointsprocess + G_fuse, gpu_process.cu:384-455 / 477-537):
//
//   k_bin_wave   one WAVE per "unit" of 64 consecutive points: coalesced float4 XYZI loads,
//                projection + variance + 2.5-D binning in registers, then a wave-local STABLE
//                grouping of the unit's points by map tile (16x16 or 32x32 cells) from ballots.
//                Emits 16-byte records {h, var, cell-in-tile, src} grouped by tile inside the unit's
//                own slice of the record arena (no global scan, no atomics) and one 16-bit
//                descriptor {start, count} per (sweep, tile, unit).
//   k_fuse_list  one workgroup per tile: ordered compaction of the tile's live descriptors,
//                records gathered into LDS in input order, arrival ranks per cell from an LDS atomic,
//                the cell's owner sorts its <= 7 slot numbers in registers and applies the
//                reference's non-associative recurrence in input order; any multiplicity beyond that
//                goes through per-wave in-order linked lists.  The tile is read once and written
//                back once with the variance floor applied.
//
// The reference's fusion is order dependent (variance floor inside the loop, Mahalanobis branch),
// so every step above preserves ascending point index per cell; there are no float atomics.
//
// Built with -ffp-contract=off (see gem_device.hpp).
#include "gem_kernels.hpp"
#include "gem_wave.hpp"

#include <hip/hip_ext.h>

#include <mutex>
#include <type_traits>

namespace gem {

// ------------------------------------------------------------------------------------------
// k_project : Process_points' kernel (GPU:384-455) for the GEM-compatible host-array entry.
// ------------------------------------------------------------------------------------------
// (`first`: the index of point 0 of these arrays in the caller's cloud -- gem_process_points hands a large cloud over in ranges)
__global__ __launch_bounds__(256) void k_project(FrameConst fc, int first, int n, float* __restrict__ x, float* __restrict__ y,
                                                 float* __restrict__ z, const int* __restrict__ orig, int write_back,
                                                 int* __restrict__ map_idx, float* __restrict__ var,
                                                 float* __restrict__ xt, float* __restrict__ yt, float* __restrict__ zt)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Projected r = project_point(fc, x[i], y[i], z[i], orig ? orig[i] : first + i);
        map_idx[i] = r.cell; var[i] = r.var; xt[i] = r.xt; yt[i] = r.yt; zt[i] = r.h;
        if (write_back && !r.accepted) { x[i] = -1.0f; y[i] = -1.0f; z[i] = -1.0f; }   // GPU:443-446
    }
}

// ------------------------------------------------------------------------------------------
// k_bin_wave : one WAVE = one unit of 64 consecutive points.  No LDS, no loops, no barriers.
// ------------------------------------------------------------------------------------------
// Descriptor word of (sweep, tile, unit), 16 bits:  start << 7 | count  (count 1..64, 0 = empty).
// Only the tiles a unit actually touches are written; k_fuse_list zeroes every word it consumes,
// so the table is all-zero again after each pass and never needs a clearing kernel.

// A binned record: {height, variance, cell in tile | colour flag << 31, index of the point}.  The last word is only read when colours are
// fused (ATTR != 0): without them the records are 12 bytes -- a quarter less to write, to read back and to stage through the LDS.
struct __attribute__((packed, aligned(4))) Rec3 { uint32_t x, y, z; };
__device__ __forceinline__ void rec_store3(uint4* rec, size_t i, uint32_t x, uint32_t y, uint32_t z)
{
    *reinterpret_cast<Rec3*>(reinterpret_cast<uint32_t*>(rec) + i * 3) = Rec3{x, y, z};
}
template <int W> __device__ __forceinline__ uint4 rec_load(const uint4* __restrict__ rec, uint32_t i)
{
    if constexpr (W == 4) return rec[i];
    else { const Rec3 r = *reinterpret_cast<const Rec3*>(reinterpret_cast<const uint32_t*>(rec) + (size_t)i * 3); return make_uint4(r.x, r.y, r.z, 0u); }
}
template <int W> __device__ __forceinline__ uint32_t rec_load_cell(const uint4* __restrict__ rec, uint32_t i)
{
    return reinterpret_cast<const uint32_t*>(rec)[(size_t)i * W + 2];
}

// one unit = 64 consecutive points, binned by one wave
template <int SRC, int TS, bool BATCH>
__device__ __forceinline__ void bin_unit(const BinArgs& a, int unit)
{
    constexpr int TE = 1 << TS;
    constexpr int U = 64;
    const int lane = lane_id();
    if (unit >= a.B) return;                               // whole wave leaves together
    if (unit == 0 && lane == 0) *a.srt_top = 0u;            // bump pointer of the sorted arena (dense tiles of k_fuse_list, same pass)

    int sweep = 0, unit_first = 0, orig0 = 0;
    long long base, sweep_begin = 0, sweep_end = a.n;
    if (BATCH) {
        int lo = 0, hi = a.n_sweeps;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.sweep_unit0[mid] <= unit) lo = mid; else hi = mid; }
        sweep = lo; unit_first = a.sweep_unit0[sweep];
        sweep_begin = a.sweep_first[sweep]; sweep_end = a.sweep_first[sweep + 1];
        orig0 = a.sweep_orig0 ? a.sweep_orig0[sweep] : 0;            // a big cloud cut into sweeps keeps its point numbering
        base = sweep_begin + (long long)(unit - unit_first) * U;
    } else {
        base = (long long)unit * U;
    }
    const FrameConst fc = BATCH ? a.frames[sweep] : a.frame0;
    const long long i = base + lane;

    bool valid = false;
    uint32_t tile = 0, cl = 0, src = 0; float hh = 0.0f, vv = 0.0f;
    if (SRC == 0 && fc.fast_laser && !a.rgb) {
        // (wave-uniform) a laser frame whose rotation variance is zero, no colours: projection + binning as straight-line code
        const float4 p = a.xyzi[i < sweep_end ? i : sweep_begin];
        int row, col;
        valid = project_bin_laser_fast(fc, p.x, p.y, p.z, i < sweep_end, a.keep_sentinel != 0, row, col, hh, vv);
        tile = (uint32_t)((row >> TS) * a.tiles_per_row + (col >> TS));
        cl = (uint32_t)(((row & (TE - 1)) << TS) | (col & (TE - 1)));
        src = (uint32_t)i;
        if (!valid) { tile = 0; cl = 0; }
    } else if (i < sweep_end) {
        int row, col; float h, v; bool colour_ok = false;
        if (SRC == 0) {
            const float4 p = a.xyzi[i];
            // (wave-uniform: a laser frame whose rotation variance is zero takes the short form of the variance, gem_device.hpp)
            const Projected r = fc.fast_laser ? project_point<kModelLaserFast>(fc, p.x, p.y, p.z, 0)
                                              : project_point(fc, p.x, p.y, p.z, a.orig ? a.orig[i] : (int)(i - sweep_begin) + orig0);
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
        // GPU:482: "point_index[i] != map_index || points_h[i] == -1" -> the point is skipped
        // (kept when the lowest scan points are tracked: GPU:430-439 sees the point, the LOWEST fuse variants skip its fusion)
        if (row >= fc.row0 && row < fc.row1 && (h != -1.0f || a.keep_sentinel)) {
            valid = true;
            tile = (uint32_t)((row >> TS) * a.tiles_per_row + (col >> TS));
            cl = (uint32_t)(((row & (TE - 1)) << TS) | (col & (TE - 1))) | (colour_ok ? 0x80000000u : 0u);
            hh = h; vv = v; src = (uint32_t)i;
        }
    }

    // group the wave's points by tile, stable in lane (= input) order
    const uint64_t peers = wave_peers_few(valid, tile, a.tile_bits);
    const uint64_t lt = lanemask_lt();
    const uint32_t rank = (uint32_t)__popcll(peers & lt);
    const uint32_t cnt = (uint32_t)__popcll(peers);
    const bool leader = valid && rank == 0;
    const uint32_t x = leader ? cnt : 0u;
    const uint32_t start_leader = wave_inclusive_scan(x) - x;       // groups laid out in order of first appearance
    const int my_leader = valid ? (__ffsll((unsigned long long)peers) - 1) : lane;
    const uint32_t start = (uint32_t)__shfl((int)start_leader, my_leader, 64);
    if (valid) {
        const size_t slot = (size_t)unit * U + start + rank;
        if (a.rec_words == 3) rec_store3(a.rec, slot, __float_as_uint(hh), __float_as_uint(vv), cl);     // (wave-uniform) no colours: the source index is dead
        else a.rec[slot] = make_uint4(__float_as_uint(hh), __float_as_uint(vv), cl, src);
    }
    if (leader) {
        // table layout [sweep][tile][unit in sweep]: the words one sweep writes stay within T * Bpad * 4 bytes
        a.seg[((size_t)sweep * a.T + tile) * a.Bpad + (unit - unit_first)] = (uint16_t)((start << kSegCountBits) | cnt);
        a.flag[(size_t)tile * a.n_sweeps + sweep] = a.epoch;       // "tile touched in this sweep" (same value from every writer)
        a.gflag[((size_t)sweep * a.T + tile) * (a.Bpad >> 5) + ((unit - unit_first) >> 5)] = a.epoch;   // "... by this group of 32 units"
    }

    if (a.counters) {
        const uint32_t nb = (uint32_t)__popcll(__ballot(valid));
        if (lane == 0 && nb) atomicAdd(&a.counters[0], (unsigned long long)nb);
    }
}

__device__ __forceinline__ void bin_stamp_begin(const BinArgs& a, int block)
{
    if (a.dbg && threadIdx.x == 0) {
        a.dbg[(size_t)block * 16] = (unsigned long long)__builtin_readcyclecounter(); a.dbg[(size_t)block * 16 + 15] = (unsigned long long)blockIdx.x + 1ull;
        a.dbg[(size_t)block * 16 + 12] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
        a.dbg[(size_t)block * 16 + 14] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) + 1ull;      // HW_REG_XCC_ID
    }
}
__device__ __forceinline__ void bin_stamp_end(const BinArgs& a, int block)
{
    if (a.dbg && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        a.dbg[(size_t)block * 16 + 1] = (unsigned long long)__builtin_readcyclecounter(); a.dbg[(size_t)block * 16 + 13] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
    }
}

template <int SRC, int TS, bool BATCH>
__device__ __forceinline__ void bin_wave_body(const BinArgs& a, int block)
{
    bin_stamp_begin(a, block);
    bin_unit<SRC, TS, BATCH>(a, (int)(block * 4 + (threadIdx.x >> 6)));
    bin_stamp_end(a, block);
}

template <int SRC, int TS, bool BATCH>
__global__ __launch_bounds__(256) void k_bin_wave(BinArgs a) { bin_wave_body<SRC, TS, BATCH>(a, (int)blockIdx.x); }

// ------------------------------------------------------------------------------------------
// k_fuse_list : one workgroup of NT threads per tile -- the lean fuse.
// ------------------------------------------------------------------------------------------
// No counting sort and no scan over cells.  Per sweep and per chunk of kChunkUnits units:
//   1. every thread reads UPT words of the tile's descriptor row; ONE packed block scan gives the
//      ORDERED compaction of the live descriptors (unit order == input order) and their record
//      prefix, so every record gets a deterministic LDS slot that is monotone in input order;
//   2. batches of <= PB records: the batch's descriptors are split into NW contiguous ranges, one
//      per wave; lanes < count gather the records (16 B / lane) into LDS, next loads in flight;
//   3a. FAST PATH (at most kRankMax records per cell and batch -- every LiDAR case): an LDS atomic
//      on the cell's row {slot[0..6], count} hands out arrival ranks; the cell's owner sorts its
//      <= 7 slot numbers with a register sorting network (slot order == input order) and runs the
//      reference's recurrence (GPU:480-531) from registers;
//   3b. GENERIC PATH (some cell got more): every wave appends its share of the batch, in order, to
//      its own per-cell linked list in LDS (head/tail[cell][wave], next[slot]; equal cells inside
//      one wave instruction are chained by a ballot match), and the owner walks list(wave 0),
//      list(wave 1), ... -- input order by construction, any multiplicity.
// The tile's elevation / variance are read once and written once.  LDS is sized by the tile:
// 16x16 tiles with PB = 1024 need ~31 KB (four workgroups per CU, VGPR-limited, so the latencies of
// one tile's phases hide behind other tiles), 32x32 tiles with PB = 2048 ~ 69 KB (two per CU).
constexpr int kChunkUnits = 2048;        // descriptor words scanned per block pass
constexpr int kRankMax    = 7;           // fast path: records per cell and batch
constexpr int fuse_list_max_batches(int pb) { return kChunkUnits * 64 / (pb - 64) + 2; }

// ------------------------------------------------------------------------------------------
// dense_tile : k_fuse_list's path for a DENSE tile (a depth camera's near field: tens of thousands of records of one
// sweep in 256 cells, chains of hundreds).  Batches in input order would keep only the few cells under the current
// image rows busy, and the tile would take the sum over batches of the longest chain.  Instead the chunk's records are
// counting-sorted by cell into the sorted arena -- stable: wave w takes the w-th contiguous share of the chunk's
// descriptors, one descriptor (<= 64 consecutive records) per step, equal cells inside a step ranked by a ballot
// match -- and every cell's owner streams its own contiguous run: the tile takes the LONGEST chain.
// Out of line on purpose: its registers must not count against the LiDAR paths of the caller.
// LDS on entry: dlc[nd] = {arena index of the first record, count} in input order, *gbase_p = the tile's base in the
// sorted arena (both written by the caller, no barrier yet); wc = [NW][CELLS] words (the rank-row area, left zeroed).
// ------------------------------------------------------------------------------------------
struct DenseResult { float e, s, lw; uint32_t n, last; };

template <int TS, int NT, int FLAGS>
__device__ __forceinline__ DenseResult dense_tile(const uint4* __restrict__ rec, uint4* __restrict__ srt_raw, uint32_t* wc, const uint2* dlc,
                                               uint32_t* scratch, const uint32_t* gbase_p, uint32_t nd, float e, float s, float lw,
                                               float mahal, float var_floor, unsigned long long* dbg)
{
    constexpr int ATTR = FLAGS & 3;
    constexpr int RW = ATTR != 0 ? 4 : 3;                                // words per binned record
    constexpr bool LOWEST = (FLAGS & 4) != 0;
#define GEM_DSTAMP(k) do { if (dbg && threadIdx.x == 0) dbg[k] = (unsigned long long)__builtin_readcyclecounter(); } while (0)
    GEM_DSTAMP(0);
    constexpr int CELLS = 1 << (2 * TS), NW = NT / 64;
    static_assert(CELLS == NT, "one cell per thread");
    using SRec = typename std::conditional<ATTR != 0, uint4, uint2>::type;   // sorted record: {h, v} (+ cell | colour flag, source index)
    SRec* const srt = reinterpret_cast<SRec*>(srt_raw);
    const int tid = (int)threadIdx.x, lane = lane_id(), w = tid >> 6;
    const uint64_t lt = lanemask_lt();
    for (int i = tid; i < NW * CELLS; i += NT) wc[i] = 0u;
    __syncthreads();
    const uint32_t dw0 = (nd * (uint32_t)w) / NW, dw1 = (nd * (uint32_t)(w + 1)) / NW;
    uint32_t* wcw = wc + w * CELLS;
    {   // count per (wave, cell): the cell words of the next eight descriptors are in flight while the current eight are counted
        constexpr int PFC = 8;
        uint32_t zA[PFC], cA[PFC], zB[PFC], cB[PFC];
        auto fetch = [&](uint32_t d, uint32_t (&z)[PFC], uint32_t (&cn)[PFC]) {
#pragma unroll
            for (int x = 0; x < PFC; ++x) {
                const uint2 de = dlc[min(d + x, nd - 1u)];
                cn[x] = d + x < dw1 ? de.y : 0u;
                z[x] = rec_load_cell<RW>(rec, de.x + ((uint32_t)lane < de.y ? (uint32_t)lane : 0u));
            }
        };
        auto count = [&](const uint32_t (&z)[PFC], const uint32_t (&cn)[PFC]) {
#pragma unroll
            for (int x = 0; x < PFC; ++x) if ((uint32_t)lane < cn[x]) atomicAdd(&wcw[z[x] & 0xffffu], 1u);
        };
        if (dw0 < dw1) {
            fetch(dw0, zA, cA);
            for (uint32_t d = dw0; d < dw1; d += 2 * PFC) {              // wave-uniform
                fetch(d + PFC, zB, cB);
                count(zA, cA);
                fetch(d + 2 * PFC, zA, cA);
                count(zB, cB);
            }
        }
    }
    __syncthreads();
    GEM_DSTAMP(1);                                                      // counted
    const uint32_t gbase = *gbase_p;
    uint32_t ctot = 0, cstart;
    {   // cell c's run starts at the cells' exclusive prefix; wave w writes behind the waves before it
        uint32_t cw[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) { cw[ww] = wc[ww * CELLS + tid]; ctot += cw[ww]; }
        uint32_t all;
        cstart = block_exclusive_scan<NT>(ctot, scratch, &all);
        uint32_t acc = cstart;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) { wc[ww * CELLS + tid] = acc; acc += cw[ww]; }
    }
    __syncthreads();
    GEM_DSTAMP(2);                                                      // cursors
    {   // placement: the records of the next four descriptors are in flight while the current four are ranked and stored
        constexpr int PFP = 4;
        uint4 rA[PFP], rB[PFP]; uint32_t cA[PFP], cB[PFP];
        auto fetch = [&](uint32_t d, uint4 (&r)[PFP], uint32_t (&cn)[PFP]) {
#pragma unroll
            for (int x = 0; x < PFP; ++x) {
                const uint2 de = dlc[min(d + x, nd - 1u)];
                cn[x] = d + x < dw1 ? de.y : 0u;
                r[x] = rec_load<RW>(rec, de.x + ((uint32_t)lane < de.y ? (uint32_t)lane : 0u));
            }
        };
        auto place = [&](const uint4 (&r)[PFP], const uint32_t (&cn)[PFP]) {
#pragma unroll
            for (int x = 0; x < PFP; ++x) {
                const bool on = (uint32_t)lane < cn[x];
                const uint32_t cell = r[x].z & 0xffffu;
                // per-bit match: a descriptor's 64 points spread over more cells than wave_peers_few's eight (measured: 70k vs 120k cycles)
                const uint64_t peers = wave_peers(on, cell, 2 * TS);
                const uint32_t rank = (uint32_t)__popcll(peers & lt);
                uint32_t old = 0;
                if (on && rank == 0) old = atomicAdd(&wcw[cell], (uint32_t)__popcll(peers));   // the wave's LDS operations execute in order
                old = (uint32_t)__shfl((int)old, on ? __ffsll((unsigned long long)peers) - 1 : lane);
                if (on) { if constexpr (ATTR != 0) srt[gbase + old + rank] = r[x]; else srt[gbase + old + rank] = make_uint2(r[x].x, r[x].y); }
            }
        };
        if (dw0 < dw1) {
            fetch(dw0, rA, cA);
            for (uint32_t d = dw0; d < dw1; d += 2 * PFP) {              // wave-uniform
                fetch(d + PFP, rB, cB);
                place(rA, cA);
                fetch(d + 2 * PFP, rA, cA);
                place(rB, cB);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GEM_DSTAMP(3);                                                      // placed
    DenseResult out; out.e = e; out.s = s; out.lw = lw; out.n = ctot; out.last = 0xffffffffu;
    {   // walk the cell's run, the records four steps ahead in flight (clamped address: never a branch round a load)
        constexpr int D = 4;
        const uint32_t n = ctot, nm1 = n ? n - 1u : 0u;
        const SRec* sp = srt + gbase + (n ? cstart : 0u);
        auto ld = [&](uint32_t i) -> SRec { return sp[min(i, nm1)]; };
        SRec pre[D];
#pragma unroll
        for (int k = 0; k < D; ++k) pre[k] = ld((uint32_t)k);
        for (uint32_t i = 0; __ballot(i < n) != 0; i += D) {             // wave-uniform
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const SRec cur = pre[k];
                const uint32_t idx = i + (uint32_t)k;
                pre[k] = ld(idx + D);
                float e2 = out.e, s2 = out.s;
                const bool taken = fuse_step(e2, s2, __uint_as_float(cur.x), __uint_as_float(cur.y), mahal, var_floor);
                const bool live = idx < n;
                const bool fl = live && (!LOWEST || __uint_as_float(cur.x) != -1.0f);      // GPU:482 (only LOWEST passes carry such records)
                out.e = fl ? e2 : out.e; out.s = fl ? s2 : out.s;
                if constexpr (LOWEST) { const float l2 = lowest_step(out.lw, __uint_as_float(cur.x), __uint_as_float(cur.y)); out.lw = live ? l2 : out.lw; }
                if constexpr (ATTR != 0) { if (fl && taken && (cur.z & 0x80000000u)) out.last = cur.w & 0x7fffffffu; }
            }
        }
    }
    GEM_DSTAMP(4);                                                      // this thread's chain done
    __syncthreads();
    GEM_DSTAMP(5);                                                      // all chains done
#undef GEM_DSTAMP
    {   // back to the fast path's invariant: every rank row has count 0
        uint4* zr = reinterpret_cast<uint4*>(wc);
        for (int c = tid; c < CELLS; c += NT) zr[c] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    return out;
}

#define GEM_CSWAP(a, b) do { const uint32_t lo_ = min(a, b), hi_ = max(a, b); a = lo_; b = hi_; } while (0)

// MODE 0: the LiDAR paths only (k_frame).  MODE 1: the same code, but a dense (tile, sweep) makes it hand the tile over --
// state in `st`, return true -- to a MODE 2 copy, which resumes at that sweep and has the dense path.  Two copies in one
// kernel keep the dense path's registers (and spills) out of the code every LiDAR tile runs.
template <int CPT> struct TileState { float e[CPT], s[CPT], lw[CPT]; uint32_t tmask, acc_nd, acc_P; int sweep; };

// FLAGS: bits 0-1 = ATTR (0 none, 1 colours from the cloud, 2 colours from gem_fuse's arrays), bit 2 = LOWEST (also maintain the
// map_lowest layer, GPU:432-439, for gem_raytracing)
constexpr int kFrameRunBits = 3;                    // k_frame: runs of 2^kFrameRunBits neighbouring tiles per XCD (fuse_list_body, RANKED; runs of 2 / 4 / 8: FETCH_SIZE 4.9 / 4.35 / 4.03 MB per C2 frame, 8.56 / 8.41 / 8.40 us per step)
constexpr int kFrameGridUnit = 8 << kFrameRunBits;  // ... its tile blocks come in multiples of this
template <int TS, int NT, int PB, int FLAGS, bool BATCH, int MODE, bool RANKED = false>
__device__ __forceinline__ bool fuse_list_body(const FuseArgs& a, int tile, unsigned char* lds_raw, TileState<(1 << (2 * TS)) / NT>& st)
{
    constexpr int ATTR = FLAGS & 3;
    constexpr bool LOWEST = (FLAGS & 4) != 0;
    // BATCH = false: one sweep, no per-sweep tables in device memory, no variance increments between sweeps --
    // the sweep loop below collapses and none of its scalar bookkeeping is compiled in
    const int NS = BATCH ? a.n_sweeps : 1;
    const float* const var_updates = BATCH ? a.var_updates : nullptr;
    constexpr int TE = 1 << TS;
    constexpr int CELLS = TE * TE;
    constexpr int NW = NT / 64;
    constexpr int CPT = CELLS / NT;
    constexpr int UPT = kChunkUnits / NT;            // descriptor words per thread and chunk (4 or 8)
    constexpr int DCAP = PB < kChunkUnits ? PB : kChunkUnits;      // a batch has at most one descriptor per record
    constexpr int MAXB = fuse_list_max_batches(PB);
    constexpr uint32_t Q = PB - 64;                  // batch b = descriptors whose record prefix lies in [b*Q, (b+1)*Q); a descriptor holds <= 64 records
    constexpr uint32_t NIL = 0xffffu;
    constexpr int XBYTES = CELLS * NW * 4 > CELLS * 16 ? CELLS * NW * 4 : CELLS * 16;
    static_assert(CELLS % NT == 0 && (UPT == 4 || UPT == 8) && (NW == 4 || NW == 8) && PB >= 512 && PB <= 4096, "geometry");

    // region X: fast path rows[CELLS] of 8 u16 {slot 0..6, count}; generic path head / tail[CELLS][NW]
    uint16_t* rowp    = reinterpret_cast<uint16_t*>(lds_raw);
    uint16_t* head    = rowp;
    uint16_t* tail    = head + CELLS * NW;
    // DMA (small batches): the records are copied HBM -> LDS as they are (global_load_lds_dwordx4, no registers in
    // between, all of a wave's loads in flight at once) into `stage`; otherwise their fields are split into s_h / s_v / s_src
    constexpr bool DMA = PB <= 1024;
    uint16_t* nxt     = reinterpret_cast<uint16_t*>(lds_raw + XBYTES);     // [PB]
    uint4*    stage   = reinterpret_cast<uint4*>(nxt + PB);                // [PB] (DMA)
    float*    s_h     = reinterpret_cast<float*>(nxt + PB);                // [PB] (!DMA)
    float*    s_v     = s_h + PB;                                          // [PB]
    uint32_t* s_src   = reinterpret_cast<uint32_t*>(s_v + PB);             // [PB] only when ATTR != 0
    uint32_t* dl_addr = DMA ? reinterpret_cast<uint32_t*>(stage + PB) : s_src + (ATTR ? PB : 0);   // [DCAP] arena index of the first record
    // words per binned record in the arena; in `stage` a record keeps a 16-byte slot either way (global_load_lds_dwordx3 writes lane l's
    // 12 bytes at base + 16 l: tools/ubench/lds_dma12.hip), the fourth word is then never written nor read
    constexpr int RW = ATTR != 0 ? 4 : 3;
    auto rec_h   = [&](uint32_t sl) -> float    { if constexpr (DMA) return __uint_as_float(stage[sl].x); else return s_h[sl]; };
    auto rec_v   = [&](uint32_t sl) -> float    { if constexpr (DMA) return __uint_as_float(stage[sl].y); else return s_v[sl]; };
    auto rec_src = [&](uint32_t sl) -> uint32_t { if constexpr (DMA && RW == 4) return (stage[sl].w & 0x7fffffffu) | (stage[sl].z & 0x80000000u); else if constexpr (DMA) return 0u; else return s_src[sl]; };
    uint32_t* dl_rc   = dl_addr + DCAP;                                    // [DCAP] record prefix << 9 | count
    uint32_t* bstart  = dl_rc + DCAP;                                      // [MAXB + 1] first descriptor of each batch
    uint32_t* scratch = bstart + MAXB + 1;                                 // [16]
    uint32_t* misc    = scratch + 16;                                      // [0] touched cells of the sweep, [1] fast-path overflow

    const int tid = (int)threadIdx.x;
    const int lane = lane_id();
    const int w = tid >> 6;
    // Block -> tile.  A single sweep (one launch per frame, latency matters): the map is robot-centric, so the tiles around
    // the sensor carry most of the records.  Tiles are ranked centre-first (rows and columns c, c-1, c+1, c-2, ... mod
    // tiles_per_row from the tile holding the map centre in storage coordinates) and rank r takes block
    // (r mod T/4) * 4 + r div (T/4): the heaviest quarter is dispatched first AND lands one per group of four consecutive
    // blocks, so the (up to four) workgroups resident on a CU are one heavy, one medium and two light tiles
    // (measured: 11.7 -> 10.6 us per C2 frame; plain centre-first order, which stacks heavy tiles on a CU: 12.1 us).
    // A batch keeps the identity mapping (every tile loops over all sweeps; the permutation cost 15 % there).
    const int tpr = a.tiles_per_row;
    const int block_in = (int)blockIdx.x;                                  // (profiling aid: stamp row word 15)
    int tr, tc;
    if constexpr (!BATCH) {
        const int q4 = (a.T + 3) >> 2;
        // RANKED (k_frame, whose tiles are all resident at once): centre-first in dispatch order, and XCD-AWARE -- workgroup b runs on
        // XCD b % 8, each XCD has its own L2, and four tiles that follow each other in a tile row share their 128-byte lines of the
        // layers: runs of 2^kFrameRunBits consecutive ranks go to ONE XCD (with plain rank = block the neighbours sat on eight different XCDs
        // and every shared line was fetched twice: FETCH_SIZE 5.6 MB per frame instead of 3.6, profiles/r05_c2_bench.txt)
        int rnk;
        if (RANKED) { const int x = tile & 7, i = tile >> 3; rnk = ((((i >> kFrameRunBits) << 3) + x) << kFrameRunBits) + (i & ((1 << kFrameRunBits) - 1)); }
        else rnk = (tile & 3) * q4 + (tile >> 2);
        if (rnk >= a.T) return false;
        const int bi = rnk / tpr, bj = rnk - bi * tpr;
        const int oi = (bi & 1) ? -((bi + 1) >> 1) : (bi >> 1);
        // columns: plain alternation c, c-1, c+1, ... or (RANKED) runs of neighbours on alternating sides: 0 1 2 3 | -1 -2 -3 -4 | 4 5 6 7 | ...
        int oj;
        if (RANKED) { const int cj = bj >> kFrameRunBits, t = bj & ((1 << kFrameRunBits) - 1); oj = (cj & 1) ? -(((cj - 1) >> 1) << kFrameRunBits) - 1 - t : ((cj >> 1) << kFrameRunBits) + t; }
        else oj = (bj & 1) ? -((bj + 1) >> 1) : (bj >> 1);
        tr = a.center_tr + oi; tr = tr < 0 ? tr + tpr : (tr >= tpr ? tr - tpr : tr);
        tc = a.center_tc + oj; tc = tc < 0 ? tc + tpr : (tc >= tpr ? tc - tpr : tc);
        tile = tr * tpr + tc;
    } else {
        if (tile >= a.T) return false;
        tr = tile / tpr; tc = tile - tr * tpr;
    }
    const int row_base = tr << TS, col_base = tc << TS;
    const int L = a.L;
    const uint32_t epoch = a.epoch;                                      // stamps the touched flags of this pass
    const uint64_t lt = lanemask_lt();
    int dbg_k = 0;
#define GEM_STAMP() do { if (a.dbg && tid == 0 && dbg_k < 12) a.dbg[(size_t)tile * 16 + dbg_k++] = (unsigned long long)__builtin_readcyclecounter(); } while (0)
    GEM_STAMP();                                                         // 0: start
    if (a.dbg && tid == 0) a.dbg[(size_t)tile * 16 + 12] = (unsigned long long)__builtin_amdgcn_s_memrealtime();   // (100 MHz, one clock for the chip)

    // which sweeps put a record into this tile?  flag[tile][sweep] == epoch (stamped by k_bin): one
    // coalesced load per 64 sweeps, turned into a wave-uniform bit mask
    const uint32_t* flagrow = a.flag + (size_t)tile * NS;
    auto sweep_mask = [&](int sbase) -> uint64_t {
        const int sidx = sbase + lane;
        return __ballot(sidx < NS && flagrow[sidx] == epoch);
    };
    auto row_ptr = [&](int sweep) -> uint16_t* { return a.seg + ((size_t)sweep * a.T + tile) * a.Bpad; };   // table layout [sweep][tile][unit in sweep]
    // A row is only read where k_bin stamped the group of 32 units (64 bytes of descriptor words) as live for
    // this tile: gflag[sweep][tile][group] == epoch.  A LiDAR ring crosses a tile in a few groups, so this cuts
    // the row traffic by an order of magnitude at the price of one small dependent load.
    auto load_gflag = [&](int sweep, int cbase) -> uint32_t {
        const int g = (cbase + tid * UPT) >> 5;
        return g < (a.Bpad >> 5) ? a.gflag[((size_t)sweep * a.T + tile) * (a.Bpad >> 5) + g] : 0u;
    };
    // the row words of a thread stay packed (one 16- or 8-byte register group) until they are needed: unpacking
    // at the load would make the compiler wait for the load right there
    using RowWords = typename std::conditional<UPT == 8, uint4, uint2>::type;
    auto load_row = [&](int sweep, int Bx, int cbase, uint32_t gf, bool& on) -> RowWords {
        const uint16_t* segx = row_ptr(sweep);
        const int u0 = cbase + tid * UPT;                                // rows are padded to 32 units
        // clamped address instead of a branch around the load: the number of loads in flight stays known
        on = gf == epoch && u0 < Bx;
        return *reinterpret_cast<const RowWords*>(segx + (on ? u0 : 0));  // `on` is applied when the words are unpacked
    };
    auto unpack_row = [&](const RowWords& q, bool on, uint32_t (&e)[UPT]) {
        e[0] = q.x & 0xffffu; e[1] = q.x >> 16; e[2] = q.y & 0xffffu; e[3] = q.y >> 16;
        if constexpr (UPT == 8) { e[4] = q.z & 0xffffu; e[5] = q.z >> 16; e[6] = q.w & 0xffffu; e[7] = q.w >> 16; }
        if (!on) {
#pragma unroll
            for (int j = 0; j < UPT; ++j) e[j] = 0;
        }
    };
    // group flags of the first chunk of sweep 0: issued before anything else so that their latency
    // overlaps the sweep-mask test
    uint32_t ev[UPT];
    RowWords evn;
    bool evn_on = false;
    const uint32_t gf0 = load_gflag(0, 0);
    uint32_t gfn = 0;
    int prefetched = -1;                                                 // sweep whose first chunk sits in evn

    // Batched call: the per-sweep tables are fetched once per block of 64 sweeps, one entry per lane, and read back with
    // v_readlane (a struct-of-pointers kernel argument carries no noalias information, so indexing them per sweep
    // would be a VECTOR load followed by a full wait -- three memory latencies per sweep and tile)
    int su0 = 0, su1 = 0; float vuv = 0.0f;
    auto sweep_tables = [&](int sbase) {
        if constexpr (BATCH) {
            const int sidx = sbase + lane < NS ? sbase + lane : NS - 1;
            su0 = a.sweep_unit0[sidx]; su1 = a.sweep_unit0[sidx + 1];
            vuv = var_updates ? var_updates[sidx] : 0.0f;
        }
    };
    auto units_of = [&](int sweep) -> int {                              // units of a sweep of the current block of 64
        if constexpr (BATCH) return __builtin_amdgcn_readlane(su1, sweep & 63) - __builtin_amdgcn_readlane(su0, sweep & 63);
        else return a.B_total;
    };
    const int sweep0 = MODE == 2 ? st.sweep : 0;                         // MODE 2 resumes where the MODE 1 copy stopped
    sweep_tables(sweep0 & ~63);

    // does this tile receive any point of this pass?
    uint64_t smask = sweep_mask(sweep0 & ~63);
    if constexpr (MODE != 2) {
        bool any_touched = smask != 0;
        for (int sb = 64; sb < NS && !any_touched; sb += 64) any_touched = sweep_mask(sb) != 0;
        if (!any_touched && !a.dense) return false;
    }
    // descriptor words of the first touched sweep: in flight before the tile itself is read, so that the
    // (larger, strided) tile loads do not sit in front of them in the memory pipeline
    // (Requesting a single sweep's whole row at once, next to its stamps -- one dependent round trip less -- was measured:
    //  no change beyond the box-to-box spread of k_frame, 9.3-9.6 us, and 2.8 MB more fetched per frame: the rows of the groups no
    //  point went to.)
    if constexpr (!BATCH) {                                              // one sweep: no branch around the load (see load_row)
        prefetched = 0;
        evn = load_row(0, a.B_total, 0, gf0, evn_on);
    } else if (MODE != 2 && smask != 0) {
        prefetched = __ffsll((unsigned long long)smask) - 1;
        evn = load_row(prefetched, units_of(prefetched), 0, prefetched == 0 ? gf0 : load_gflag(prefetched, 0), evn_on);
    }

    // ---- the single read of the tile ---------------------------------------------------------------
    float ce[CPT], cs[CPT], lw[CPT];
    float ce0[CPT], cs0[CPT];                                            // as loaded: only cells that changed are written back
    bool  owned[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int c = tid + NT * q;
        const int row = row_base + (c >> TS), col = col_base + (c & (TE - 1));
        owned[q] = row < a.row1 && row >= a.row0 && col < L;
        // unconditional loads (clamped address): a load inside a branch would make the number of loads in flight
        // unknown to the compiler, which then waits for ALL of them wherever it needs the descriptor words
        // (cells outside the map or the strip read cell 0; they receive no records and are never written back)
        const size_t g = owned[q] ? (size_t)row * L + col : 0;
        if constexpr (MODE == 2) { ce[q] = st.e[q]; cs[q] = st.s[q]; }
        else { ce[q] = a.elevation[g]; cs[q] = a.variance[g]; }
        ce0[q] = ce[q]; cs0[q] = cs[q];
        if constexpr (LOWEST) {
            // map_lowest is indexed by the GEOGRAPHIC cell (GPU:430 PointsToIndex), not by the circular-buffer cell
            int gr = row - a.start0, gc = col - a.start1;
            gr += gr < 0 ? L : 0; gc += gc < 0 ? L : 0;
            if constexpr (MODE == 2) lw[q] = st.lw[q]; else lw[q] = a.lowest[owned[q] ? (size_t)gr * L + gc : 0];
        }
    }

    {   // fast-path rows start with count 0; the owner leaves every row it consumed at count 0 again
        uint4* z = reinterpret_cast<uint4*>(rowp);
        for (int c = tid; c < CELLS; c += NT) z[c] = make_uint4(0, 0, 0, 0);
        if (tid == 0) misc[1] = 0;
    }
    GEM_STAMP();                                                         // 1: tile loads issued

    // colour / intensity of the last taken point with all four non-zero (GPU:487-494)
    auto write_attr = [&](int q, uint32_t last) {
        if constexpr (ATTR != 0) {
            const int c = tid + NT * q;
            const size_t g = (size_t)(row_base + (c >> TS)) * L + col_base + (c & (TE - 1));
            if (ATTR == 1) {
                const uint32_t cc = a.rgb[last];
                a.intensity[g] = a.xyzi[last].w;
                a.colorR[g] = (int)((cc >> 16) & 0xff); a.colorG[g] = (int)((cc >> 8) & 0xff); a.colorB[g] = (int)(cc & 0xff);
            } else {
                a.intensity[g] = a.f_I[last];
                a.colorR[g] = a.f_R[last]; a.colorG[g] = a.f_G[last]; a.colorB[g] = a.f_B[last];
            }
        }
    };

    uint32_t tmask = MODE == 2 ? st.tmask : 0u;                          // cells of this thread touched in this sweep (or pass)
    // Accumulate mode: when no variance increment separates the sweeps of a batched call (an aggregated cloud cut into sweeps,
    // or a batch without var_updates) the descriptors of consecutive sweeps are collected into ONE batch until it is full, so
    // that a tile that gets a few records from each of many sweeps pays the gather / rank / walk latencies once per PB records
    // instead of once per sweep (C5, 77 sweeps: k_fuse_list 1.46 ms -> see DESIGN.md).  Input order is preserved (sweep, unit).
    // Per-sweep cell counts need the sweeps apart, so counting per sweep turns it off.
    const bool acc_mode = BATCH && !var_updates && (!a.counters || a.count_per_pass);
    uint32_t acc_nd = MODE == 2 ? st.acc_nd : 0u, acc_P = MODE == 2 ? st.acc_P : 0u;   // descriptors / records collected so far (block-uniform)
    uint32_t scan_parity = 0;                                            // see block_exclusive_scan_alt
    bool pend_done = MODE == 2 && sweep0 > 0;                            // queued Mapvar_update increments: applied once, before the first record
    for (int sweep = sweep0; sweep < NS; ++sweep) {
        if (sweep != sweep0 && (sweep & 63) == 0) { smask = sweep_mask(sweep); sweep_tables(sweep); }
        const bool touched_sweep = (smask >> (sweep & 63)) & 1ull;       // block-uniform
        // a sweep that neither reaches this tile nor carries a variance increment changes nothing
        // (the floor below is idempotent and has been applied by an earlier sweep or is applied by a later one)
        if (!touched_sweep && !var_updates && sweep != 0 && sweep != NS - 1) continue;
        const int ub = BATCH ? __builtin_amdgcn_readlane(su0, sweep & 63) : 0;        // multiple of 32 (host pads sweeps)
        const int ue = BATCH ? __builtin_amdgcn_readlane(su1, sweep & 63) : a.B_total;
        const int B = ue - ub;

        // ---- Mapvar_update increments queued before this sweep (GPU:540-547): applied lazily, right before
        //      the first use of the variances in this sweep, so that the tile read (strided 64-byte segments,
        //      the slowest loads of the kernel) stays in flight behind the descriptor scan and the record gather
        bool incs_applied = false;
        auto apply_increments = [&]() {
            if (incs_applied) return;
            incs_applied = true;
            const bool pend_applied = pend_done;
            pend_done = true;
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                if (!pend_applied)
                    for (int k = 0; k < a.n_pending; ++k) if (cs[q] != kInitVariance) cs[q] += a.pending[k];
                if (var_updates) { const float u = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vuv), sweep & 63)); if (cs[q] != kInitVariance) cs[q] += u; }
            }
        };
        if (a.counters && tid == 0 && (sweep == 0 || !a.count_per_pass)) misc[0] = 0;
        if (!a.count_per_pass) tmask = 0;

        // one batch: m descriptors in dl_addr / dl_rc (written, barrier passed), records [slot0, slot0 + PB) -> LDS, then the chains
        auto run_batch = [&](const uint32_t m, const uint32_t slot0) {
            // ---- 2 + 3a. gather into LDS in input order (record k -> slot k - slot0) and take an
            //      arrival rank per cell for the fast path.  Wave w takes the w-th contiguous share of
            //      the batch's descriptors, one descriptor per step (lanes < count).
            const uint32_t dw0 = (m * (uint32_t)w) / NW, dw1 = (m * (uint32_t)(w + 1)) / NW;
            auto rank_one = [&](uint32_t cell, uint32_t sl) {
                const uint32_t old = atomicAdd(reinterpret_cast<uint32_t*>(rowp) + cell * 4 + 3, 0x10000u);
                const uint32_t rk = old >> 16;
                if (rk < (uint32_t)kRankMax) rowp[cell * 8 + rk] = (uint16_t)sl;
                else misc[1] = 1u;
            };
            if constexpr (DMA) {
                // lane i fetches the wave's i-th descriptor; each is then broadcast (v_readlane) and turned into ONE
                // global_load_lds_dwordx4 whose LDS base is the descriptor's first slot: nothing waits until all are issued
                for (uint32_t dbase = dw0; dbase < dw1; dbase += 64) {             // wave-uniform
                    const uint32_t nhere = min(64u, dw1 - dbase);
                    uint32_t my_rc = 0, my_addr = 0;
                    if ((uint32_t)lane < nhere) { my_rc = dl_rc[dbase + lane]; my_addr = dl_addr[dbase + lane]; }
                    for (uint32_t i = 0; i < nhere; ++i) {
                        const uint32_t rc = (uint32_t)__builtin_amdgcn_readlane((int)my_rc, (int)i);
                        const uint32_t adr = (uint32_t)__builtin_amdgcn_readlane((int)my_addr, (int)i);
                        if ((uint32_t)lane < (rc & 0x1ffu))
                        {
                            const auto* gsrc = (const __attribute__((address_space(1))) void*)(reinterpret_cast<const uint32_t*>(a.rec) + (size_t)(adr + (uint32_t)lane) * (ATTR != 0 ? 4 : 3));
                            auto* ldst = (__attribute__((address_space(3))) void*)(stage + ((rc >> 9) - slot0));
                            if constexpr (ATTR != 0) __builtin_amdgcn_global_load_lds(gsrc, ldst, 16, 0, 0);
                            else                     __builtin_amdgcn_global_load_lds(gsrc, ldst, 12, 0, 0);
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const uint32_t first = (dl_rc[0] >> 9) - slot0;                     // slots of the batch: [first, first + Pb)
                const uint32_t Pb = (dl_rc[m - 1] >> 9) + (dl_rc[m - 1] & 0x1ffu) - (dl_rc[0] >> 9);
                // PB / NT slots per thread: all cell reads, then all rank atomics, then all row writes in flight together
                constexpr int RK = PB / NT;
                uint32_t rcell[RK], rold[RK];
#pragma unroll
                for (int r = 0; r < RK; ++r) { const uint32_t k = (uint32_t)(tid + r * NT); rcell[r] = k < Pb ? (stage[first + k].z & 0xffffu) : 0u; }
#pragma unroll
                for (int r = 0; r < RK; ++r) { const uint32_t k = (uint32_t)(tid + r * NT); rold[r] = 0; if (k < Pb) rold[r] = atomicAdd(reinterpret_cast<uint32_t*>(rowp) + rcell[r] * 4 + 3, 0x10000u); }
#pragma unroll
                for (int r = 0; r < RK; ++r) {
                    const uint32_t k = (uint32_t)(tid + r * NT);
                    if (k < Pb) { const uint32_t rk = rold[r] >> 16; if (rk < (uint32_t)kRankMax) rowp[rcell[r] * 8 + rk] = (uint16_t)(first + k); else misc[1] = 1u; }
                }
            } else {
                // through registers, the loads of the next PF descriptors in flight while the current PF are being filed
                constexpr int PF = 4;
                uint4 rA[PF], rB[PF]; uint32_t sA[PF], sB[PF];       // slot == ~0: lane inactive
                auto fetch = [&](uint32_t dd, uint4 (&rr)[PF], uint32_t (&sl)[PF]) {
#pragma unroll
                    for (int x = 0; x < PF; ++x) {
                        sl[x] = 0xffffffffu; rr[x] = make_uint4(0, 0, 0, 0);
                        if (dd + x < dw1) {                          // wave-uniform
                            const uint32_t rc = dl_rc[dd + x];
                            if ((uint32_t)lane < (rc & 0x1ffu)) {
                                sl[x] = (rc >> 9) - slot0 + (uint32_t)lane;
                                rr[x] = rec_load<RW>(a.rec, dl_addr[dd + x] + (uint32_t)lane);
                            }
                        }
                    }
                };
                auto file = [&](const uint4 (&rr)[PF], const uint32_t (&sl)[PF]) {
#pragma unroll
                    for (int x = 0; x < PF; ++x) {
                        if (sl[x] != 0xffffffffu) {
                            const uint32_t cell = rr[x].z & 0xffffu;
                            s_h[sl[x]] = __uint_as_float(rr[x].x); s_v[sl[x]] = __uint_as_float(rr[x].y);
                            if (ATTR) s_src[sl[x]] = (rr[x].w & 0x7fffffffu) | (rr[x].z & 0x80000000u);
                            nxt[sl[x]] = (uint16_t)cell;             // the generic path reads the cell from here
                            rank_one(cell, sl[x]);
                        }
                    }
                };
                if (dw0 < dw1) {
                    fetch(dw0, rA, sA);
                    for (uint32_t dd = dw0; dd < dw1; dd += 2 * PF) {   // wave-uniform
                        fetch(dd + PF, rB, sB);
                        file(rA, sA);
                        fetch(dd + 2 * PF, rA, sA);
                        file(rB, sB);
                    }
                }
            }
            __syncthreads();
            GEM_STAMP();                                             // 4: records in LDS, ranked

            apply_increments();
            if (misc[1] == 0) {
                // ---- 3a. owner: sort <= 7 slot numbers, run the chains from registers.  The CPT cells
                //      of a thread are independent chains: everything is straight-line with selects so
                //      that their sorting networks and IEEE divisions interleave.
                uint4 row[CPT]; uint32_t n[CPT], nmax = 0;
#pragma unroll
                for (int q = 0; q < CPT; ++q) row[q] = *reinterpret_cast<const uint4*>(rowp + (tid + NT * q) * 8);
                uint32_t ps[CPT][kRankMax];
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    n[q] = row[q].w >> 16;
                    nmax = max(nmax, n[q]);
                    if (n[q] != 0) { reinterpret_cast<uint32_t*>(rowp + (tid + NT * q) * 8)[3] = 0u; tmask |= 1u << q; }
                    uint32_t p0 = row[q].x & 0xffffu, p1 = row[q].x >> 16, p2 = row[q].y & 0xffffu, p3 = row[q].y >> 16,
                             p4 = row[q].z & 0xffffu, p5 = row[q].z >> 16, p6 = row[q].w & 0xffffu, p7 = NIL;
                    if (n[q] < 1) p0 = NIL; if (n[q] < 2) p1 = NIL; if (n[q] < 3) p2 = NIL; if (n[q] < 4) p3 = NIL;
                    if (n[q] < 5) p4 = NIL; if (n[q] < 6) p5 = NIL; if (n[q] < 7) p6 = NIL;
                    GEM_CSWAP(p0, p1); GEM_CSWAP(p2, p3); GEM_CSWAP(p4, p5); GEM_CSWAP(p6, p7);
                    GEM_CSWAP(p0, p2); GEM_CSWAP(p1, p3); GEM_CSWAP(p4, p6); GEM_CSWAP(p5, p7);
                    GEM_CSWAP(p1, p2); GEM_CSWAP(p5, p6); GEM_CSWAP(p0, p4); GEM_CSWAP(p3, p7);
                    GEM_CSWAP(p1, p5); GEM_CSWAP(p2, p6);
                    GEM_CSWAP(p1, p4); GEM_CSWAP(p3, p6);
                    GEM_CSWAP(p2, p4); GEM_CSWAP(p3, p5);
                    GEM_CSWAP(p3, p4);
                    ps[q][0] = p0; ps[q][1] = p1; ps[q][2] = p2; ps[q][3] = p3; ps[q][4] = p4; ps[q][5] = p5; ps[q][6] = p6;
                }
                float hh[CPT][kRankMax], vv[CPT][kRankMax]; uint32_t sv[CPT][kRankMax];
#pragma unroll
                for (int i = 0; i < kRankMax; ++i) {
                    if (__ballot((uint32_t)i < nmax) == 0) break;    // wave-uniform
#pragma unroll
                    for (int q = 0; q < CPT; ++q) {
                        const uint32_t sl = (uint32_t)i < n[q] ? ps[q][i] : 0u;       // slot 0 is always a valid address
                        hh[q][i] = rec_h(sl); vv[q][i] = rec_v(sl);
                        if (ATTR) sv[q][i] = rec_src(sl);
                    }
                }
                uint32_t wlast[CPT];
#pragma unroll
                for (int q = 0; q < CPT; ++q) wlast[q] = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < kRankMax; ++i) {
                    if (__ballot((uint32_t)i < nmax) == 0) break;    // wave-uniform
#pragma unroll
                    for (int q = 0; q < CPT; ++q) {
                        float e2 = ce[q], s2 = cs[q];
                        const bool taken = fuse_step(e2, s2, hh[q][i], vv[q][i], a.mahal, a.var_floor);
                        const bool live = (uint32_t)i < n[q];
                        const bool fl = live && (!LOWEST || hh[q][i] != -1.0f);      // GPU:482 (only LOWEST passes carry such records)
                        ce[q] = fl ? e2 : ce[q]; cs[q] = fl ? s2 : cs[q];
                        if constexpr (LOWEST) { const float l2 = lowest_step(lw[q], hh[q][i], vv[q][i]); lw[q] = live ? l2 : lw[q]; }
                        if (ATTR) { if (fl && taken && (sv[q][i] & 0x80000000u)) wlast[q] = sv[q][i] & 0x7fffffffu; }
                    }
                }
                if (ATTR) {
#pragma unroll
                    for (int q = 0; q < CPT; ++q) if (wlast[q] != 0xffffffffu) write_attr(q, wlast[q]);
                }
            } else {
                // ---- 3b. generic path: per-wave in-order linked lists --------------------------
                const uint32_t k_lo = dl_rc[0] >> 9;
                const uint32_t Pb = (dl_rc[m - 1] >> 9) + (dl_rc[m - 1] & 0x1ffu) - k_lo;      // records of this batch
                {
                    uint4* z = reinterpret_cast<uint4*>(head);
                    for (int c = tid; c < CELLS * NW / 8; c += NT) z[c] = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
                }
                __syncthreads();
                if (tid == 0) misc[1] = 0u;
                {   // wave w appends records [w * per, (w + 1) * per) of the batch, 64 consecutive records per step
                    const uint32_t per = ((Pb + NW * 64u - 1u) / (NW * 64u)) * 64u;
                    const uint32_t kbeg = (uint32_t)w * per, kend = min(Pb, kbeg + per);
                    for (uint32_t kc = kbeg; kc < kend; kc += 64) {   // wave-uniform
                        const bool on = kc + (uint32_t)lane < kend;
                        const uint32_t sl = k_lo - slot0 + kc + (uint32_t)lane;
                        uint32_t cell = 0u;
                        if (on) { if constexpr (DMA) cell = stage[sl].z & 0xffffu; else cell = nxt[sl]; }
                        const uint64_t peers = wave_peers(on, cell, 2 * TS);
                        const uint64_t above = lane == 63 ? 0ull : (peers & (~0ull << (lane + 1)));
                        if (on) {
                            nxt[sl] = above ? (uint16_t)(sl + (uint32_t)(__ffsll((unsigned long long)above) - 1 - lane)) : (uint16_t)NIL;
                            const uint32_t hi = cell * NW + (uint32_t)w;
                            if ((peers & lt) == 0) {                  // first of its group: link behind the wave's list of this cell
                                if (head[hi] == NIL) head[hi] = (uint16_t)sl;
                                else nxt[tail[hi]] = (uint16_t)sl;
                            }
                            if (above == 0) tail[hi] = (uint16_t)sl;
                        }
                    }
                }
                __syncthreads();
                // walk: the CPT cells of a thread advance together (independent chains), one node per
                // cell and iteration, through list(wave 0), list(wave 1), ...
                {
                    uint32_t cur[CPT], wwq[CPT], wl[CPT], hdp[CPT][NW];
                    bool more = false;
#pragma unroll
                    for (int q = 0; q < CPT; ++q) {
                        const int c = tid + NT * q;
                        if constexpr (NW == 4) {
                            const uint2 hv = *reinterpret_cast<const uint2*>(head + c * NW);
                            hdp[q][0] = hv.x & 0xffffu; hdp[q][1] = hv.x >> 16; hdp[q][2] = hv.y & 0xffffu; hdp[q][3] = hv.y >> 16;
                        } else {
                            const uint4 hv = *reinterpret_cast<const uint4*>(head + c * NW);
                            hdp[q][0] = hv.x & 0xffffu; hdp[q][1] = hv.x >> 16; hdp[q][2] = hv.y & 0xffffu; hdp[q][3] = hv.y >> 16;
                            hdp[q][4] = hv.z & 0xffffu; hdp[q][5] = hv.z >> 16; hdp[q][6] = hv.w & 0xffffu; hdp[q][7] = hv.w >> 16;
                        }
                        // chain the wave lists of the cell: cur = first non-empty head, and each list's
                        // successor is the next non-empty head (resolved when its NIL end is reached)
                        cur[q] = NIL; wwq[q] = NW; wl[q] = 0xffffffffu;
#pragma unroll
                        for (int ww = NW - 1; ww >= 0; --ww) if (hdp[q][ww] != NIL) { cur[q] = hdp[q][ww]; wwq[q] = (uint32_t)ww; }
                        if (cur[q] != NIL) { tmask |= 1u << q; more = true; }
                    }
                    // software-pipelined: the node after the current one is resolved from registers and its LDS
                    // reads are issued before the current node's fusion step, so the two latencies overlap
                    float nh[CPT], nv[CPT]; uint32_t nn[CPT], nsv[CPT];
#pragma unroll
                    for (int q = 0; q < CPT; ++q) {
                        const uint32_t sl = cur[q] != NIL ? cur[q] : 0u;
                        nh[q] = rec_h(sl); nv[q] = rec_v(sl); nn[q] = nxt[sl];
                        if (ATTR) nsv[q] = rec_src(sl);
                    }
                    while (__ballot(more) != 0) {                    // wave-uniform
                        more = false;
#pragma unroll
                        for (int q = 0; q < CPT; ++q) {
                            const bool live = cur[q] != NIL;
                            const float h = nh[q], v = nv[q];
                            const uint32_t sv = ATTR ? nsv[q] : 0u;
                            uint32_t nx = nn[q];
                            if (live && nx == NIL) {                 // end of this wave's list: continue with the next non-empty one
                                uint32_t nw_ = NW;
#pragma unroll
                                for (int ww = NW - 1; ww >= 0; --ww) if ((uint32_t)ww > wwq[q] && hdp[q][ww] != NIL) { nx = hdp[q][ww]; nw_ = (uint32_t)ww; }
                                wwq[q] = nw_;
                            }
                            nx = live ? nx : NIL;
                            const uint32_t sl2 = nx != NIL ? nx : 0u;
                            nh[q] = rec_h(sl2); nv[q] = rec_v(sl2); nn[q] = nxt[sl2];
                            if (ATTR) nsv[q] = rec_src(sl2);
                            float e2 = ce[q], s2 = cs[q];
                            const bool taken = fuse_step(e2, s2, h, v, a.mahal, a.var_floor);
                            const bool fl = live && (!LOWEST || h != -1.0f);          // GPU:482 (only LOWEST passes carry such records)
                            ce[q] = fl ? e2 : ce[q]; cs[q] = fl ? s2 : cs[q];
                            if constexpr (LOWEST) { const float l2 = lowest_step(lw[q], h, v); lw[q] = live ? l2 : lw[q]; }
                            if (ATTR) { if (fl && taken && (sv & 0x80000000u)) wl[q] = sv & 0x7fffffffu; }
                            cur[q] = nx;
                            more |= cur[q] != NIL;
                        }
                    }
                    if (ATTR) {
#pragma unroll
                        for (int q = 0; q < CPT; ++q) if (wl[q] != 0xffffffffu) write_attr(q, wl[q]);
                    }
                }
                __syncthreads();
                {   // back to the fast path's invariant: every row has count 0
                    uint4* z = reinterpret_cast<uint4*>(rowp);
                    for (int c = tid; c < CELLS; c += NT) z[c] = make_uint4(0, 0, 0, 0);
                }
            }
            __syncthreads();
            GEM_STAMP();                                             // 5: walked
        };

        // a single sweep holds at most kChunkUnits units (longer clouds are cut into sweeps): exactly one chunk, and no
        // loop header at which the compiler would have to wait for every load in flight
        const int n_chunks = !touched_sweep ? 0 : (BATCH ? (B + kChunkUnits - 1) / kChunkUnits : 1);
        for (int ci = 0; ci < n_chunks; ++ci) {
            const int cbase = ci * kChunkUnits;
            // ---- 1. ordered compaction of the chunk's live descriptors ----------------------------
            const int u0 = cbase + tid * UPT;
            int next_sweep = -1;
            if (cbase == 0 && prefetched == sweep) unpack_row(evn, evn_on, ev);
            else { bool on; const RowWords q = load_row(sweep, B, cbase, load_gflag(sweep, cbase), on); unpack_row(q, on, ev); }
            if (cbase == 0) {                                            // next touched sweep of this 64-block: its group flags start flying now,
                const uint64_t later = (sweep & 63) == 63 ? 0ull : (smask >> ((sweep & 63) + 1));   // its row after the scan below
                if (later != 0) { next_sweep = sweep + 1 + (__ffsll((unsigned long long)later) - 1); gfn = load_gflag(next_sweep, 0); }
            }
            uint32_t packed = 0;                                         // live descriptors << 20 | records
            {
                uint32_t any = 0;
#pragma unroll
                for (int j = 0; j < UPT; ++j) {
                    const bool live = ev[j] != 0 && u0 + j < B;
                    if (!live) ev[j] = 0;
                    any |= ev[j];
                    packed += live ? ((1u << 20) | (ev[j] & kSegCountMask)) : 0u;
                }
            }
            if (a.dbg) { asm volatile("" :: "v"(packed)); GEM_STAMP(); }                      // 2: descriptor words arrived
            uint32_t tot;
            const uint32_t run = block_exclusive_scan_alt<NT>(packed, scratch, scan_parity++, &tot);
            const uint32_t nd = tot >> 20, P = tot & 0xfffffu;
            if (next_sweep >= 0) { evn = load_row(next_sweep, units_of(next_sweep), 0, gfn, evn_on); prefetched = next_sweep; }
            if (P == 0) continue;                                        // block-uniform
            if constexpr (MODE == 1 && DMA && CPT == 1) {
                if (ci == 0 && P > a.dense_min) {                        // block-uniform: a dense tile, nothing of this sweep consumed yet
#pragma unroll
                    for (int q = 0; q < CPT; ++q) { st.e[q] = ce[q]; st.s[q] = cs[q]; if constexpr (LOWEST) st.lw[q] = lw[q]; }
                    st.tmask = tmask; st.sweep = sweep; st.acc_nd = acc_nd; st.acc_P = acc_P;
                    return true;
                }
            }
            {   // consumed: one wide store of zeros over the words this thread read, so that the table is all-zero
                // again after the pass (its other words already are)
                uint32_t any = 0;
#pragma unroll
                for (int j = 0; j < UPT; ++j) any |= ev[j];
                if (any) {
                    uint16_t* rowx = row_ptr(sweep) + u0;
                    if constexpr (UPT == 8) *reinterpret_cast<uint4*>(rowx) = make_uint4(0, 0, 0, 0);
                    else                    *reinterpret_cast<uint2*>(rowx) = make_uint2(0, 0);
                }
            }
            const uint32_t nb = (P - 1u) / Q + 1u;                       // batches 0 .. nb-2 are non-empty (a descriptor holds < Q records)
            if (nb > 1) {
                for (uint32_t i = tid; i < nb; i += NT) bstart[i] = 0xffffffffu;
                if (tid == 0) bstart[nb] = nd;
                __syncthreads();
                uint32_t d = run >> 20, rs = run & 0xfffffu;
#pragma unroll
                for (int j = 0; j < UPT; ++j)
                    if (ev[j] != 0) { atomicMin(&bstart[rs / Q], d); ++d; rs += ev[j] & kSegCountMask; }
                __syncthreads();
            }

            // ---- batches.  In accumulate mode a batch first collects the descriptors of several sweeps (see acc_mode above):
            //      b == -1 runs the accumulated batch when this sweep cannot join it (too many records, or a dense tile);
            //      b >= 0 are this sweep's own batches, unless it was appended or takes the dense path.
            bool dense_now = false, fits = false;
            if constexpr (MODE == 2 && DMA && CPT == 1) dense_now = P > a.dense_min;
            if (acc_mode) fits = !dense_now && acc_P + P <= Q && acc_nd + nd <= (uint32_t)DCAP;
            const int b_first = (acc_P != 0 && !fits) ? -1 : 0;
            const int b_end = (dense_now || fits) ? 0 : (int)nb;
            for (int b = b_first; b < b_end; ++b) {
                uint32_t m, slot0;
                if (b < 0) {
                    m = acc_nd; slot0 = 0u; acc_nd = 0u; acc_P = 0u;     // the list already sits in dl_addr / dl_rc
                } else {
                    // (the last batch is empty when the final descriptor merely extends past a multiple of Q)
                    const uint32_t d_lo = nb > 1 ? bstart[b] : 0u;
                    if (d_lo == 0xffffffffu) break;                      // block-uniform
                    const uint32_t d_hi = nb > 1 ? min(bstart[b + 1], nd) : nd;
                    m = d_hi - d_lo; slot0 = (uint32_t)b * Q;
                    // this batch's descriptors, in order
                    uint32_t d = run >> 20, rs = run & 0xfffffu;
#pragma unroll
                    for (int j = 0; j < UPT; ++j) {
                        if (ev[j] != 0) {
                            const uint32_t cnt = ev[j] & kSegCountMask;
                            if (d >= d_lo && d < d_hi) {
                                dl_addr[d - d_lo] = (uint32_t)(ub + u0 + j) * (uint32_t)a.U + ((ev[j] >> kSegCountBits) & kSegStartMask);
                                dl_rc[d - d_lo] = (rs << 9) | cnt;
                            }
                            ++d; rs += cnt;
                        }
                    }
                }
                __syncthreads();
                GEM_STAMP();                                             // 3: descriptor list built
                run_batch(m, slot0);
            }
            if constexpr (MODE == 2 && DMA && CPT == 1) {
                if (dense_now) {                                         // block-uniform
                    // ---- DENSE TILE: see dense_tile().  The chunk's descriptor list goes to the stage area, the rest is
                    //      an out-of-line call so that its registers do not count against the LiDAR paths below.
                    static_assert(!DMA || PB * 16 >= kChunkUnits * 8, "the chunk's descriptor list lives in the stage area");
                    uint2* dlc = reinterpret_cast<uint2*>(stage);        // [nd] {arena index of the first record, count}
                    {
                        uint32_t d = run >> 20;
#pragma unroll
                        for (int j = 0; j < UPT; ++j)
                            if (ev[j] != 0) {
                                dlc[d++] = make_uint2((uint32_t)(ub + u0 + j) * (uint32_t)a.U + ((ev[j] >> kSegCountBits) & kSegStartMask),
                                                      ev[j] & kSegCountMask);
                            }
                        if (tid == 0) misc[2] = atomicAdd(a.srt_top, P);
                    }
                    apply_increments();
                    const DenseResult dr = dense_tile<TS, NT, FLAGS>(a.rec, a.srt, reinterpret_cast<uint32_t*>(rowp), dlc, scratch, misc + 2,
                                                                    nd, ce[0], cs[0], LOWEST ? lw[0] : 0.0f, a.mahal, a.var_floor,
                                                                    a.dbg && sweep == a.dbg_sweep ? a.dbg + (size_t)tile * 16 + 8 : nullptr);
                    ce[0] = dr.e; cs[0] = dr.s;
                    if constexpr (LOWEST) lw[0] = dr.lw;
                    if (dr.n) tmask |= 1u;
                    if (ATTR != 0 && dr.last != 0xffffffffu) write_attr(0, dr.last);
                    GEM_STAMP();
                    continue;
                }
            }
            if (fits) {
                // this sweep's descriptors join the accumulated batch (record prefixes continue where the batch stands)
                uint32_t d = acc_nd + (run >> 20), rs = acc_P + (run & 0xfffffu);
#pragma unroll
                for (int j = 0; j < UPT; ++j) {
                    if (ev[j] != 0) {
                        const uint32_t cnt = ev[j] & kSegCountMask;
                        dl_addr[d] = (uint32_t)(ub + u0 + j) * (uint32_t)a.U + ((ev[j] >> kSegCountBits) & kSegStartMask);
                        dl_rc[d] = (rs << 9) | cnt;
                        ++d; rs += cnt;
                    }
                }
                acc_nd += nd; acc_P += P;
            }
        }
        if (acc_P != 0 && sweep == NS - 1) {                             // block-uniform: the last accumulated batch of the pass
            __syncthreads();
            const uint32_t m = acc_nd;
            acc_nd = 0u; acc_P = 0u;
            run_batch(m, 0u);
        }

        // ---- variance floor at the end of every Fuse (GPU:533-534), on every cell -----------------
        apply_increments();
#pragma unroll
        for (int q = 0; q < CPT; ++q) if (cs[q] < a.var_floor) cs[q] = a.var_floor;

        if (a.counters && (!a.count_per_pass || sweep == NS - 1)) {
            __syncthreads();
            if (tmask) atomicAdd(&misc[0], (uint32_t)__popc(tmask));
            __syncthreads();
            if (tid == 0 && misc[0]) atomicAdd(&a.counters[1], (unsigned long long)misc[0]);
            __syncthreads();
        }
    }

    // ---- the single write-back of the tile ---------------------------------------------------------
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        if (owned[q]) {
            int c = tid + NT * q;
#if defined(__HIP_DEVICE_COMPILE__)
            // (recomputed from the thread's number HERE: the address of the tile's read, kept alive across the whole kernel for this
            //  store, was the one value the 80-register budget of k_frame spilled -- 8 bytes per thread, 5 MB of scratch traffic per frame)
            asm volatile("" : "+v"(c));
#endif
            const size_t g = (size_t)(row_base + (c >> TS)) * L + col_base + (c & (TE - 1));
            // A sweep touches a fraction of a tile's cells: writing the tile back whole was 1.8 MB of the 4.3 MB a C2 frame wrote
            // (profiles/r01f_c2_bench.txt).  (A tile resumed by the dense copy of the loop has lost its loaded values: written whole.)
            if (MODE == 2 || __float_as_uint(ce[q]) != __float_as_uint(ce0[q])) a.elevation[g] = ce[q];
            if (MODE == 2 || __float_as_uint(cs[q]) != __float_as_uint(cs0[q])) a.variance[g] = cs[q];
            if constexpr (LOWEST) {
                // (its geographic address recomputed like g: kept alive from the tile's read it was the value k_frame<4>'s budget spilled)
                int gr = row_base + (c >> TS) - a.start0, gc = col_base + (c & (TE - 1)) - a.start1;
                gr += gr < 0 ? L : 0; gc += gc < 0 ? L : 0;
                a.lowest[(size_t)gr * L + gc] = lw[q];
            }
        }
    }
    GEM_STAMP();                                                         // 6: stores issued
    if (a.dbg && tid == 0) {
        a.dbg[(size_t)tile * 16 + 13] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
        a.dbg[(size_t)tile * 16 + 15] = (unsigned long long)block_in + 1ull;
        a.dbg[(size_t)tile * 16 + 14] = (unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) + 1ull;      // HW_REG_XCC_ID: the XCD's clock is its own
    }
#undef GEM_STAMP
    return false;
}

// waves per SIMD the register budget of a k_fuse_list instantiation is set for.  The single-sweep forms fit 128 VGPRs (four waves);
// the BATCH forms carry the sweep loop's tables and, with colours / lowest scan points, spilled up to 46 VGPRs there (scratch
// traffic in the middle of the chains): they get the budget of one wave less.  32x32 tiles (maps whose 16x16 descriptor table
// would exceed 512 MB, or the tile_shift knob) are a fallback: whatever does not spill.  tests/test_code_objects.py holds every
// kernel of the library to zero VGPR spills.
constexpr int fuse_list_waves(int TS, int NT, int PB, int FLAGS, bool BATCH)
{
    if (TS == 4) return BATCH ? 3 : 4;
    if (PB > 2048) return (BATCH && NT == 256) ? 1 : 2;
    return BATCH ? 2 : 4;
}

template <int TS, int NT, int PB, int ATTR, bool BATCH>
__global__ __launch_bounds__(NT, fuse_list_waves(TS, NT, PB, ATTR, BATCH)) void k_fuse_list(FuseArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
    TileState<(1 << (2 * TS)) / NT> st;
    if constexpr (TS == 4 && PB <= 1024) {                               // 16x16 tiles: dense tiles are handed to the second copy
        if (fuse_list_body<TS, NT, PB, ATTR, BATCH, 1>(a, (int)blockIdx.x, lds_dyn, st)) {
            __syncthreads();
            fuse_list_body<TS, NT, PB, ATTR, BATCH, 2>(a, (int)blockIdx.x, lds_dyn, st);
        }
    } else {
        fuse_list_body<TS, NT, PB, ATTR, BATCH, 0>(a, (int)blockIdx.x, lds_dyn, st);
    }
}

// ------------------------------------------------------------------------------------------
// k_frame : the steady state of a stream of single sweeps in ONE launch per frame -- the fusion of the
// previous frame's records (blocks [0, T)) next to the binning of this frame's cloud (blocks [T, T + B/4)).
// Binning depends on the cloud and the pose only, so the two halves are independent; they use the two halves
// of the double-buffered arenas.  Saves one kernel boundary (~3 us here) per frame and lets the binning run on
// the CUs the fuse leaves idle.
// ------------------------------------------------------------------------------------------
// FLAGS: 0, or 4 = the fusion also maintains map_lowest (GPU:432-439; the adapter of the unmodified node turns it on for Raytracing).
// Round 5, from the launch's own time line (tools/frame_phases.py, profiles/r05_c2_frame_phases.txt: every workgroup's start and
// end on the chip-wide 100 MHz clock): the chip starts about 850 workgroups of this kernel per microsecond, a tile lives 3.5 us
// whatever it holds (three dependent memory round trips, five barrier-separated phases), and with 85 VGPRs / 31 KB of LDS five
// workgroups fit a CU -- 1280 of the launch's 1956, so the rest waited for the first tiles to END and the launch took two tile
// lifetimes.  Now: rounds of kFramePB = 768 records (24 KB) and at most 80 VGPRs -- six workgroups per CU, 1536 slots: every tile
// is resident from the start, the binning blocks follow as the light tiles leave -- and the tiles in plain centre-first order
// (the heaviest are dispatched first; the round-1 interleave by quarters spread them over the whole 1.5 us ramp): 9.1 -> 7.9 us.
// Measured and dropped: the binning blocks first (+0.9 us: every tile starts later), two or eight units per binning wave
// (+0.7 / +6 us: a unit is ~3 k cycles of a wave's issue, the blocks became the launch's tail), rounds of 512 records with
// 64 VGPRs and eight workgroups per CU (+3.5 us: the fuse body spills).
constexpr int kFramePB = 768;        // records per LDS round of k_frame's fuse half (k_fuse_list alone keeps 1024)
constexpr int kFrameWG = 6;          // workgroups per CU the register budget is set for
template <int FLAGS>
__global__ __launch_bounds__(256, kFrameWG) void k_frame(FuseArgs fa, BinArgs ba)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[];
    const int nf = (fa.T + kFrameGridUnit - 1) & ~(kFrameGridUnit - 1);     // fuse blocks (see the block -> tile mapping)
    if ((int)blockIdx.x < nf) { TileState<1> st; fuse_list_body<4, 256, kFramePB, FLAGS, false, 0, true>(fa, (int)blockIdx.x, lds_dyn, st); }
    else bin_wave_body<0, 4, false>(ba, (int)blockIdx.x - nf);
}

// ------------------------------------------------------------------------------------------
// dense / state kernels
// ------------------------------------------------------------------------------------------
// G_Init_map (GPU:198-214) and G_Clear_allmap (GPU:216-230; does not touch map_lowest)
__global__ __launch_bounds__(256) void k_init(LayerPtrs m, int cells, int clear_lowest)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += gridDim.x * blockDim.x) {
        m.intensity[i] = 0.0f; m.elevation[i] = kEmptyElevation; m.variance[i] = kInitVariance;
        m.traver[i] = -10.0f; m.rough[i] = 0.0f; m.slope[i] = 0.0f;
        if (clear_lowest) m.lowest[i] = 100.0f;
        m.colorR[i] = 0; m.colorG[i] = 0; m.colorB[i] = 0;
    }
}

// G_Clear_map (GPU:255-276): `count` rows (is_row) or columns starting at storage index `start`
__global__ __launch_bounds__(256) void k_clear_strip(LayerPtrs m, int L, int start, int count, int is_row)
{
    const int total = L * count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int c = is_row ? start * L + i : (i / count) * L + (i % count) + start;
        m.intensity[c] = 0.0f; m.elevation[c] = kEmptyElevation; m.variance[c] = kInitVariance;
        m.colorR[c] = 0; m.colorG[c] = 0; m.colorB[c] = 0;
    }
}

// G_Mapvar_update (GPU:540-547) for up to 4 queued increments, optionally followed by the
// variance floor of G_fuse's tail (GPU:533-534) -- used when no cloud is being fused.
__global__ __launch_bounds__(256) void k_dense_variance(float* __restrict__ variance, int cells, int n_pending,
                                                        float p0, float p1, float p2, float p3, int apply_floor, float var_floor)
{
    const float pend[4] = {p0, p1, p2, p3};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += gridDim.x * blockDim.x) {
        float s = variance[i];
        for (int k = 0; k < n_pending; ++k) if (s != kInitVariance) s += pend[k];
        if (apply_floor && s < var_floor) s = var_floor;
        variance[i] = s;
    }
}

// G_update_mapheight (GPU:1195-1202): loop-closure height shift of every cell that holds an elevation
__global__ __launch_bounds__(256) void k_update_height(float* __restrict__ elevation, int cells, float dz)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += gridDim.x * blockDim.x) {
        const float e = elevation[i];
        if (e != kEmptyElevation) elevation[i] = e + dz;
    }
}

// Device arrays -> the handle's pinned staging buffer (gem_capi.cpp: download_arrays): up to kCopyListMax pieces per launch, piece =
// blockIdx.y.  Stores to host memory over PCIe, a wave writing 1 KiB contiguous; one launch instead of one DMA command per array
// (whose fixed cost, not its rate, is what a frame's five 0.5 MB outputs pay).  Pieces are 16-byte aligned except for a tail.
__global__ __launch_bounds__(256) void k_copy_list(CopyList l)
{
    const CopyPiece pc = l.piece[blockIdx.y];
    if ((reinterpret_cast<uintptr_t>(pc.src) | reinterpret_cast<uintptr_t>(pc.dst)) & 15) {        // (block-uniform) a source that is only word-aligned
        const size_t n4 = pc.bytes / 4;
        const uint32_t* __restrict__ s4 = static_cast<const uint32_t*>(pc.src);
        uint32_t* __restrict__ d4 = static_cast<uint32_t*>(pc.dst);
        const bool words = ((reinterpret_cast<uintptr_t>(pc.src) | reinterpret_cast<uintptr_t>(pc.dst)) & 3) == 0;
        if (words) for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) d4[i] = s4[i];
        const size_t done = words ? n4 * 4 : 0;
        for (size_t i = done + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pc.bytes; i += (size_t)gridDim.x * blockDim.x)
            static_cast<unsigned char*>(pc.dst)[i] = static_cast<const unsigned char*>(pc.src)[i];
        return;
    }
    const size_t n16 = pc.bytes / 16;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const u32x4* __restrict__ s = static_cast<const u32x4*>(pc.src);
    u32x4* __restrict__ d = static_cast<u32x4*>(pc.dst);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(s[i], d + i);
    if (blockIdx.x == 0 && threadIdx.x < (pc.bytes & 15))
        static_cast<unsigned char*>(pc.dst)[n16 * 16 + threadIdx.x] = static_cast<const unsigned char*>(pc.src)[n16 * 16 + threadIdx.x];
}

// grid_map export (EM.cpp:98-111 reads the flat arrays with the GridMap *buffer* index; a
// grid_map::Matrix is an Eigen column-major float matrix; empty cells become NaN)
__global__ __launch_bounds__(256) void k_export_gridmap(const void* __restrict__ src, const float* __restrict__ elevation,
                                                        float* __restrict__ dst, int L, int is_int)
{
    const int cells = L * L;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += gridDim.x * blockDim.x) {
        const int col = i / L, row = i - col * L;            // dst is column-major: i = col*L + row
        const int g = row * L + col;                         // storage row-major
        float v = is_int ? (float)reinterpret_cast<const int*>(src)[g] : reinterpret_cast<const float*>(src)[g];
        if (elevation[g] == kEmptyElevation) v = __builtin_nanf("");
        dst[i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// The feed of ElevationMap::show (EM.cpp:85-149): the reference walks all L^2 cells on the host in grid_map's
// iteration order (linear index of the column-major matrix = buffer index), copies nine layers into visualMap_ where the cell
// holds an elevation AND a traversability, pushes a coloured point per such cell and paints the orthomosaic.  Here two small
// kernels do it on the resident layers: k_show_count counts the kept cells per block of 1024 linear indices, k_show_emit turns
// the counts of the blocks before it into its offset (order preserved: the point list equals the reference's) and writes the
// column-major layers (NaN elsewhere), the compacted points and the image.  Positions are grid_map's getPositionFromIndex in
// double (GridMapMath.cpp): mapPosition + (mapLength / 2 - resolution / 2) - resolution * unwrapped index.
// ------------------------------------------------------------------------------------------
struct ShowArgs {
    LayerPtrs m; int L, sx, sy;
    double off, res, px, py;                 // off = 0.5 * map_length - 0.5 * resolution
    uint32_t* block_count;                   // [blocks]
    float* visual; float* xyz; unsigned char* rgb; unsigned char* image; uint32_t* total;
};

__device__ __forceinline__ bool show_keep(const ShowArgs& a, size_t lin, size_t& index, int& ix, int& iy)
{
    ix = (int)(lin % (size_t)a.L); iy = (int)(lin / (size_t)a.L);       // EM.cpp:98-99: the buffer index of the linear index
    index = (size_t)ix * a.L + iy;                                      // EM.cpp:100
    const float tr = a.m.traver[index];
    return a.m.elevation[index] != kEmptyElevation && tr != -10.0f && !(tr != tr);   // EM.cpp:101
}

__global__ __launch_bounds__(1024) void k_show_count(ShowArgs a)
{
    __shared__ uint32_t scratch[16];
    const size_t lin = (size_t)blockIdx.x * 1024 + threadIdx.x, cells = (size_t)a.L * a.L;
    size_t index; int ix, iy;
    const bool keep = lin < cells && show_keep(a, lin, index, ix, iy);
    uint32_t total;
    block_exclusive_scan<1024>(keep ? 1u : 0u, scratch, &total);
    if (threadIdx.x == 0) a.block_count[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void k_show_emit(ShowArgs a)
{
    __shared__ uint32_t scratch[16];
    __shared__ uint32_t s_base;
    const size_t lin = (size_t)blockIdx.x * 1024 + threadIdx.x, cells = (size_t)a.L * a.L;
    // kept cells in the blocks before this one
    uint32_t part = 0;
    for (int b = (int)threadIdx.x; b < (int)blockIdx.x; b += 1024) part += a.block_count[b];
    uint32_t before;
    block_exclusive_scan<1024>(part, scratch, &before);
    if (threadIdx.x == 0) s_base = before;
    size_t index = 0; int ix = 0, iy = 0;
    const bool keep = lin < cells && show_keep(a, lin, index, ix, iy);
    uint32_t in_block;
    const uint32_t rank = block_exclusive_scan<1024>(keep ? 1u : 0u, scratch, &in_block);
    __syncthreads();
    const uint32_t n = s_base + rank;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *a.total = s_base + in_block;
    if (lin >= cells) return;
    const float nan = __builtin_nanf("");
    float vals[9] = {nan, nan, nan, nan, nan, nan, nan, nan, nan};
    if (keep) {
        const float cr = (float)a.m.colorR[index], cg = (float)a.m.colorG[index], cb = (float)a.m.colorB[index];    // EM.cpp:103-111
        vals[0] = a.m.elevation[index]; vals[1] = a.m.variance[index]; vals[2] = a.m.rough[index]; vals[3] = a.m.slope[index];
        vals[4] = a.m.traver[index]; vals[5] = cr; vals[6] = cg; vals[7] = cb; vals[8] = a.m.intensity[index];
        int ux = ix - a.sx, uy = iy - a.sy;                             // getIndexFromBufferIndex
        ux += ux < 0 ? a.L : 0; uy += uy < 0 ? a.L : 0;
        const unsigned char r8 = (unsigned char)(int)cr, g8 = (unsigned char)(int)cg, b8 = (unsigned char)(int)cb;
        if (a.xyz) {
            const double px = (a.px + a.off) + a.res * (double)(-ux);   // getPositionFromIndex (doubles)
            const double py = (a.py + a.off) + a.res * (double)(-uy);
            a.xyz[3 * (size_t)n + 0] = (float)px; a.xyz[3 * (size_t)n + 1] = (float)py; a.xyz[3 * (size_t)n + 2] = vals[0];   // EM.cpp:116-118
        }
        if (a.rgb) { a.rgb[3 * (size_t)n + 0] = r8; a.rgb[3 * (size_t)n + 1] = g8; a.rgb[3 * (size_t)n + 2] = b8; }
        if (a.image) { unsigned char* p = a.image + ((size_t)ux * a.L + uy) * 3; p[0] = b8; p[1] = g8; p[2] = r8; }          // EM.cpp:124-126
    }
    if (a.visual) {
#pragma unroll
        for (int l = 0; l < 9; ++l) a.visual[(size_t)l * cells + lin] = vals[l];
    }
}

hipError_t launch_show(hipStream_t st, const LayerPtrs& m, int L, int sx, int sy, double map_length, double resolution, double px, double py,
                       uint32_t* block_count, float* visual, float* xyz, unsigned char* rgb, unsigned char* image, uint32_t* total)
{
    ShowArgs a{};
    a.m = m; a.L = L; a.sx = sx; a.sy = sy; a.off = 0.5 * map_length - 0.5 * resolution; a.res = resolution; a.px = px; a.py = py;
    a.block_count = block_count; a.visual = visual; a.xyz = xyz; a.rgb = rgb; a.image = image; a.total = total;
    const int blocks = (int)(((size_t)L * L + 1023) / 1024);
    hipLaunchKernelGGL(k_show_count, dim3(blocks), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_show_emit, dim3(blocks), dim3(1024), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// k_map_feature : traversability stage that follows the fusion every frame (G_Mapfeature,
// GPU:549-670, with the Jacobi eigen-solver computerEigenvalue, GPU:66-187).  One thread per cell, one workgroup per 16x16 cells:
// plane fit over the valid cells of the 5x5 neighbourhood (bounds in unrolled coordinates, wrapped
// storage reads -- and, like the reference, STORAGE coordinates times the resolution as x / y), smallest
// eigenvector -> slope, |h - mean z| -> roughness, traver = 0.5 (1 - slope/0.6) + 0.5 (1 - rough/0.2).
// The neighbourhood is walked twice (means, then covariance) instead of staging 25 points per thread;
// both walks add in the reference's order.  The symmetric 3x3 matrix lives in six scalars: of the
// reference's six off-diagonal candidates only (0,1), (0,2), (1,2) can win its strict-greater scan.
// Trigonometry: the reference's float calls are evaluated in double and rounded (see oracle/ and DESIGN.md).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float f_sin(float x)            { return (float)sin((double)x); }
__device__ __forceinline__ float f_cos(float x)            { return (float)cos((double)x); }
__device__ __forceinline__ float f_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }
__device__ __forceinline__ float f_acos(float x)           { return (float)acos((double)x); }

// The Jacobi loop runs for the cells whose covariance has an off-diagonal entry of at least 0.01 -- slopes, steps, windows cut by
// the map's edge or by holes: a minority, scattered over every wave, each wave as slow as its slowest lane (up to 30 rotations of five
// double-precision libm calls).  The workgroup therefore COMPACTS them: the cells that rotate put their six matrix entries into LDS,
// the first `count` threads run the loop on dense waves and hand back what the cell needs (the diagonal and the eigenvectors' z
// components); the same operations on the same values per cell, in whatever lane.
struct JacobiOut { float d0, d1, d2, z0, z1, z2; };
__device__ __forceinline__ JacobiOut jacobi_3x3(float a00, float a11, float a22, float a01, float a02, float a12)
{
    // ---- computerEigenvalue (GPU:66-187), dbEps = 0.01, nJt = 30 ----
    float v00 = 1.f, v01 = 0.f, v02 = 0.f, v10 = 0.f, v11 = 1.f, v12 = 0.f, v20 = 0.f, v21 = 0.f, v22 = 1.f;
    int count = 0;
    while (true) {
        float mxv = a01; int pair = 0;                          // GPU:85-100 (signed start value, strict >)
        { const float d = fabsf(a01); if (d > mxv) { mxv = d; pair = 0; } }
        { const float d = fabsf(a02); if (d > mxv) { mxv = d; pair = 1; } }
        { const float d = fabsf(a12); if (d > mxv) { mxv = d; pair = 2; } }
        if (mxv < 0.01f) break;
        if (count > 30) break;
        ++count;
        // (p, q, r): pair 0 -> (0,1,2), 1 -> (0,2,1), 2 -> (1,2,0)
        float app, aqq, apq, arp, arq;
        if (pair == 0)      { app = a00; aqq = a11; apq = a01; arp = a02; arq = a12; }
        else if (pair == 1) { app = a00; aqq = a22; apq = a02; arp = a01; arq = a12; }
        else                { app = a11; aqq = a22; apq = a12; arp = a01; arq = a02; }
        const float ang = (float)(0.5 * (double)f_atan2(-2 * apq, aqq - app));             // GPU:116
        const float sn = f_sin(ang), cs = f_cos(ang), sn2 = f_sin(2 * ang), cs2 = f_cos(2 * ang);
        const float npp = app * cs * cs + aqq * sn * sn + 2 * apq * cs * sn;                // GPU:122-123
        const float nqq = app * sn * sn + aqq * cs * cs - 2 * apq * cs * sn;                // GPU:124-125
        const float npq = (float)(0.5 * (double)(aqq - app) * (double)sn2 + (double)(apq * cs2));   // GPU:126
        const float nrp = arq * sn + arp * cs;                                               // GPU:129-151
        const float nrq = arq * cs - arp * sn;
        if (pair == 0)      { a00 = npp; a11 = nqq; a01 = npq; a02 = nrp; a12 = nrq; }
        else if (pair == 1) { a00 = npp; a22 = nqq; a02 = npq; a01 = nrp; a12 = nrq; }
        else                { a11 = npp; a22 = nqq; a12 = npq; a01 = nrp; a02 = nrq; }
        // eigenvector columns p, q (GPU:154-161)
        float u0, u1, u2, w0, w1, w2;
        if (pair == 0)      { u0 = v00; u1 = v10; u2 = v20; w0 = v01; w1 = v11; w2 = v21; }
        else if (pair == 1) { u0 = v00; u1 = v10; u2 = v20; w0 = v02; w1 = v12; w2 = v22; }
        else                { u0 = v01; u1 = v11; u2 = v21; w0 = v02; w1 = v12; w2 = v22; }
        const float nu0 = w0 * sn + u0 * cs, nw0 = w0 * cs - u0 * sn;
        const float nu1 = w1 * sn + u1 * cs, nw1 = w1 * cs - u1 * sn;
        const float nu2 = w2 * sn + u2 * cs, nw2 = w2 * cs - u2 * sn;
        if (pair == 0)      { v00 = nu0; v10 = nu1; v20 = nu2; v01 = nw0; v11 = nw1; v21 = nw2; }
        else if (pair == 1) { v00 = nu0; v10 = nu1; v20 = nu2; v02 = nw0; v12 = nw1; v22 = nw2; }
        else                { v01 = nu0; v11 = nu1; v21 = nu2; v02 = nw0; v12 = nw1; v22 = nw2; }
    }
    return JacobiOut{a00, a11, a22, v20, v21, v22};
}

__global__ __launch_bounds__(256) void k_map_feature(const float* __restrict__ elevation, float* __restrict__ traver,
                                                     float* __restrict__ rough, float* __restrict__ slope,
                                                     int L, float res, int sx, int sy, int row0, int row1)
{
    // one workgroup per 16x16 block of storage cells; the block and its 2-cell halo (storage wrap-around) are staged
    // in LDS once: 400 loads per 256 cells instead of 50 per cell
    __shared__ float zt[20 * 20];
    __shared__ float jq[6][256];                                       // the rotating cells' matrices, then their results
    __shared__ uint32_t jn;
    const int tiles = (L + 15) >> 4;
    const int tr_ = (int)blockIdx.x / tiles, tc_ = (int)blockIdx.x - tr_ * tiles;
    const int R0 = tr_ << 4, C0 = tc_ << 4;
    for (int t = (int)threadIdx.x; t < 400; t += 256) {
        const int i = t / 20, j = t - i * 20;
        int px = R0 - 2 + i; px = px < 0 ? px + L : (px >= L ? px - L : px); px = px >= L ? px - L : px;
        int py = C0 - 2 + j; py = py < 0 ? py + L : (py >= L ? py - L : py); py = py >= L ? py - L : py;
        zt[t] = elevation[px * L + py];
    }
    if (threadIdx.x == 0) jn = 0u;
    __syncthreads();
    const int ly = (int)threadIdx.x >> 4, lx = (int)threadIdx.x & 15;
    const int cell_x = R0 + ly, cell_y = C0 + lx;
    const bool mine = cell_x < L && cell_y < L && cell_x >= row0 && cell_x < row1;      // (multi-GPU: only the owned strip)
    const int idx = cell_x * L + cell_y;
    const float height = mine ? zt[(ly + 2) * 20 + lx + 2] : kEmptyElevation;
    float mz = 0.0f;
    float a00 = 0.f, a11 = 0.f, a22 = 0.f, a01 = 0.f, a02 = 0.f, a12 = 0.f;
    bool fitted = false;                                                // n > 7: the plane fit exists
    if (height != kEmptyElevation) {                                    // GPU:581
        int gx = cell_x + L - sx; if (gx >= L) gx -= L;                 // unrolled index of the cell, GPU:587-588
        int gy = cell_y + L - sy; if (gy >= L) gy -= L;
        // per row / column of the window: inside the map in unrolled coordinates (GPU:587-593)?  storage coordinate
        // (wrapped, GPU:596-600) times the resolution
        bool rv[5], cv[5]; float xs[5], ys[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int ex = gx + k - 2, ey = gy + k - 2;
            rv[k] = ex >= 0 && ex < L; cv[k] = ey >= 0 && ey < L;
            int px = cell_x + k - 2; px = px < 0 ? px + L : (px >= L ? px - L : px);
            int py = cell_y + k - 2; py = py < 0 ? py + L : (py >= L ? py - L : py);
            xs[k] = (float)px * res; ys[k] = (float)py * res;
        }
        float mx = 0.0f, my = 0.0f;
        int n = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float z = zt[(ly + i) * 20 + lx + j];
                if (rv[i] && cv[j] && z != kEmptyElevation) { mx = mx + xs[i]; my = my + ys[j]; mz = mz + z; ++n; }
            }
        if (n > 7) {
            fitted = true;
            mx = mx / (float)n; my = my / (float)n; mz = mz / (float)n;
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const float z = zt[(ly + i) * 20 + lx + j];
                    if (rv[i] && cv[j] && z != kEmptyElevation) {       // GPU:624-635
                        const float dx = xs[i] - mx, dy = ys[j] - my, dz = z - mz;
                        a00 = a00 + dx * dx; a11 = a11 + dy * dy; a22 = a22 + dz * dz;
                        a01 = a01 + dx * dy; a02 = a02 + dx * dz; a12 = a12 + dy * dz;
                    }
                }
        }
    }
    // does the loop rotate at all?  (its own first test: the signed a01 or the largest magnitude reaches 0.01)
    float mxv = a01;
    { const float d = fabsf(a01); if (d > mxv) mxv = d; }
    { const float d = fabsf(a02); if (d > mxv) mxv = d; }
    { const float d = fabsf(a12); if (d > mxv) mxv = d; }
    const bool rotates = fitted && !(mxv < 0.01f);
    uint32_t slot = 0;
    {
        const uint64_t mk = __ballot(rotates);
        uint32_t base = 0;
        if (lane_id() == 0 && mk) base = atomicAdd(&jn, (uint32_t)__popcll(mk));
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        slot = base + (uint32_t)__popcll(mk & lanemask_lt());
        if (rotates) { jq[0][slot] = a00; jq[1][slot] = a11; jq[2][slot] = a22; jq[3][slot] = a01; jq[4][slot] = a02; jq[5][slot] = a12; }
    }
    __syncthreads();
    if (threadIdx.x < jn) {
        const int t = (int)threadIdx.x;
        const JacobiOut o = jacobi_3x3(jq[0][t], jq[1][t], jq[2][t], jq[3][t], jq[4][t], jq[5][t]);
        jq[0][t] = o.d0; jq[1][t] = o.d1; jq[2][t] = o.d2; jq[3][t] = o.z0; jq[4][t] = o.z1; jq[5][t] = o.z2;
    }
    __syncthreads();
    if (!mine) return;
    float r_out = 0.0f, s_out = 0.0f;
    if (height != kEmptyElevation) {
        float tr = -10.0f;                                              // GPU:660-666
        if (fitted) {
            float d0 = a00, d1 = a11, d2 = a22, z0 = 0.f, z1 = 0.f, z2 = 1.f;          // no rotation: the eigenvectors are the axes
            if (rotates) { d0 = jq[0][slot]; d1 = jq[1][slot]; d2 = jq[2][slot]; z0 = jq[3][slot]; z1 = jq[4][slot]; z2 = jq[5][slot]; }
            // z component of the eigenvector of the smallest eigenvalue (first minimum wins, GPU:168-186)
            float mn = d0, nz = z0;
            if (mn > d1) { mn = d1; nz = z1; }
            if (mn > d2) { mn = d2; nz = z2; }
            const float anz = nz > 0 ? nz : -nz;                        // GPU:647-650; acos(1) is exactly 0: spare flat cells the double acos
            const float Slope = anz == 1.0f ? 0.0f : f_acos(anz);
            const float Rough = fabsf(height - mz);
            tr = (float)(0.5 * (1.0 - (double)Slope / 0.6) + 0.5 * (1.0 - ((double)Rough / 0.2)));   // GPU:653
            s_out = Slope; r_out = Rough;
        }
        traver[idx] = tr;                                               // map_traver, GPU:658 / 665
    }
    rough[idx] = r_out;
    slope[idx] = s_out;
}

// ------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------
// Launch `k`; with a (start, stop) event pair the dispatch itself is time-stamped (hipExtLaunchKernelGGL),
// which measures the kernel alone -- no launch gap, no event-record overhead.
#define GEM_LAUNCH(k, grid, block, lds, st, ev, ...)                                              \
    do {                                                                                           \
        if ((ev).start || (ev).stop) hipExtLaunchKernelGGL(k, grid, block, (uint32_t)(lds), st, (ev).start, (ev).stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(k, grid, block, lds, st, __VA_ARGS__);                             \
    } while (0)

constexpr int kMaxDevices = 64;          // per-device launch configuration caches

static inline int grid_for(long long work, int block, int cap = 2048)
{
    long long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

hipError_t launch_project(hipStream_t st, const FrameConst& fc, int first, int n, float* x, float* y, float* z, const int* orig,
                          int write_back, int* map_idx, float* var, float* xt, float* yt, float* zt)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_project, dim3(grid_for(n, 256)), dim3(256), 0, st, fc, first, n, x, y, z, orig, write_back, map_idx, var, xt, yt, zt);
    return hipGetLastError();
}

template <int SRC>
static hipError_t launch_bin_wave(hipStream_t st, const BinArgs& a, int ts, LaunchEvents ev)
{
    const dim3 grid((a.B + 3) / 4), block(256);
    const bool batch = a.n_sweeps > 1;
    if (ts == 4) {
        if (batch) GEM_LAUNCH((k_bin_wave<SRC, 4, true>), grid, block, 0, st, ev, a);
        else       GEM_LAUNCH((k_bin_wave<SRC, 4, false>), grid, block, 0, st, ev, a);
    } else {
        if (batch) GEM_LAUNCH((k_bin_wave<SRC, 5, true>), grid, block, 0, st, ev, a);
        else       GEM_LAUNCH((k_bin_wave<SRC, 5, false>), grid, block, 0, st, ev, a);
    }
    return hipGetLastError();
}

hipError_t launch_bin(hipStream_t st, const BinArgs& a, int src, int ts, LaunchEvents ev)
{
    if (a.B <= 0) return hipSuccess;
    return src == 0 ? launch_bin_wave<0>(st, a, ts, ev) : launch_bin_wave<1>(st, a, ts, ev);
}

// ---- k_fuse_list -------------------------------------------------------------------------------------
static size_t fuse_list_lds(int cells, int nw, int pb, int attr)
{
    const size_t x = (size_t)cells * nw * 4 > (size_t)cells * 16 ? (size_t)cells * nw * 4 : (size_t)cells * 16;   // head + tail | rank rows
    const size_t dcap = pb < kChunkUnits ? pb : kChunkUnits;
    size_t b = x
             + (size_t)pb * 2                          // nxt
             + (pb <= 1024 ? (size_t)pb * 16 : (size_t)pb * 4 * (attr ? 3 : 2))   // stage | s_h, s_v (, s_src)
             + dcap * 4 * 2                            // dl_addr, dl_rc
             + (size_t)(fuse_list_max_batches(pb) + 1) * 4
             + 16 * 4 + 16;                            // scratch, misc
    return (b + 15) & ~(size_t)15;
}

// (tile shift, variant) -> threads per tile and records per LDS batch
//   16x16 tiles: 256 threads, 1024-record batches (~31 KB of LDS, 4 workgroups per CU)
//   32x32 tiles: variant 10 = 256 threads / 4096, 11 = 512 / 4096, 12 = 512 / 2048 (2 per CU, the default)
static void fuse_list_geometry(int ts, int variant, int* nt, int* pb)
{
    if (ts == 4) { *nt = 256; *pb = 1024; }
    else         { *nt = variant == 10 ? 256 : 512; *pb = (variant == 10 || variant == 11) ? 4096 : 2048; }
}

size_t fuse_lds_bytes(int ts, int variant, int attr)
{
    int nt, pb; fuse_list_geometry(ts, variant, &nt, &pb);
    return fuse_list_lds((1 << (2 * ts)), nt / 64, pb, attr);
}

template <int TS, int NT, int PB, bool BATCH>
static hipError_t launch_fuse_list_b(hipStream_t st, const FuseArgs& a, int attr, LaunchEvents ev)
{
    const size_t lds = fuse_list_lds(1 << (2 * TS), NT / 64, PB, attr);
    if (lds > 64 * 1024) {   // more than 64 KiB of dynamic LDS needs an explicit opt-in, once per kernel AND device
        static std::mutex mu;
        static size_t configured[kMaxDevices][3] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lk(mu);
        if (dev < 0 || dev >= kMaxDevices || lds > configured[dev][attr]) {
            const void* fn = attr == 0 ? (const void*)k_fuse_list<TS, NT, PB, 0, BATCH> : attr == 1 ? (const void*)k_fuse_list<TS, NT, PB, 1, BATCH>
                                                                                                       : (const void*)k_fuse_list<TS, NT, PB, 2, BATCH>;
            e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < kMaxDevices) configured[dev][attr] = lds;
        }
    }
    if (attr == 0)      GEM_LAUNCH((k_fuse_list<TS, NT, PB, 0, BATCH>), dim3((a.T + 3) & ~3), dim3(NT), lds, st, ev, a);
    else if (attr == 1) GEM_LAUNCH((k_fuse_list<TS, NT, PB, 1, BATCH>), dim3((a.T + 3) & ~3), dim3(NT), lds, st, ev, a);
    else                GEM_LAUNCH((k_fuse_list<TS, NT, PB, 2, BATCH>), dim3((a.T + 3) & ~3), dim3(NT), lds, st, ev, a);
    return hipGetLastError();
}

// the variants that also maintain map_lowest (16x16 tiles only)
template <bool BATCH>
static hipError_t launch_fuse_list_lowest(hipStream_t st, const FuseArgs& a, int attr, LaunchEvents ev)
{
    const size_t lds = fuse_list_lds(256, 4, 1024, attr & 3);
    const dim3 grid((a.T + 3) & ~3), block(256);
    if ((attr & 3) == 0)      GEM_LAUNCH((k_fuse_list<4, 256, 1024, 4, BATCH>), grid, block, lds, st, ev, a);
    else if ((attr & 3) == 1) GEM_LAUNCH((k_fuse_list<4, 256, 1024, 5, BATCH>), grid, block, lds, st, ev, a);
    else                      GEM_LAUNCH((k_fuse_list<4, 256, 1024, 6, BATCH>), grid, block, lds, st, ev, a);
    return hipGetLastError();
}

template <int TS, int NT, int PB>
static hipError_t launch_fuse_list(hipStream_t st, const FuseArgs& a, int attr, LaunchEvents ev)
{
    return a.n_sweeps > 1 ? launch_fuse_list_b<TS, NT, PB, true>(st, a, attr, ev) : launch_fuse_list_b<TS, NT, PB, false>(st, a, attr, ev);
}

hipError_t launch_fuse(hipStream_t st, const FuseArgs& a, int ts, int attr, int variant, LaunchEvents ev)
{
    if (a.T <= 0) return hipSuccess;
    if (attr & 4) {
        if (ts != 4) return hipErrorInvalidValue;
        return a.n_sweeps > 1 ? launch_fuse_list_lowest<true>(st, a, attr, ev) : launch_fuse_list_lowest<false>(st, a, attr, ev);
    }
    if (ts == 4) return launch_fuse_list<4, 256, 1024>(st, a, attr, ev);
    if (variant == 10) return launch_fuse_list<5, 256, 4096>(st, a, attr, ev);
    if (variant == 11) return launch_fuse_list<5, 512, 4096>(st, a, attr, ev);
    return launch_fuse_list<5, 512, 2048>(st, a, attr, ev);
}

// fuse of the previous frame + bin of this one (single sweeps on 16x16 tiles, no attributes)
hipError_t launch_frame(hipStream_t st, const FuseArgs& fa, const BinArgs& ba, int attr, LaunchEvents ev)
{
    const dim3 grid(((fa.T + kFrameGridUnit - 1) & ~(kFrameGridUnit - 1)) + (ba.B + 3) / 4), block(256);
    if (attr == 4)      GEM_LAUNCH((k_frame<4>), grid, block, fuse_list_lds(256, 4, kFramePB, 0), st, ev, fa, ba);
    else if (attr == 0) GEM_LAUNCH((k_frame<0>), grid, block, fuse_list_lds(256, 4, kFramePB, 0), st, ev, fa, ba);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// k_raytracing : the visibility clean-up, G_Raytracing (GPU:708-891; helpers GPU:672-706).
// A cell that holds an elevation and whose traversability is below the obstacle threshold walks AWAY from the map centre
// along the centre->cell ray (a DDA over cell borders); every crossed cell with a lowest scan point this frame bounds the
// obstacle's height by the line of sight from the sensor over that point; if elevation - 3 sigma is above the tightest
// bound the cell is deleted.  Quirks kept: map_lowest in GEOGRAPHIC cell order, the
// int robot_index, centre row / column cells never deleted, x-only abscissae, "no scan point" == 10.
// A thread writes only its own cell's elevation and reads only its own cell's elevation / variance: no ordering issue.
// The walks are long (up to ~2 L steps) and few (the obstacle cells): each is split over G lanes (k_raytracing, below).
// ------------------------------------------------------------------------------------------
// The DDA divides by the same two direction components at every step (GPU:831-832 and twins).  hipcc expands an IEEE float division
// into v_div_scale x 2, v_rcp, two Newton steps on the reciprocal, the quotient, two residual corrections (the last one
// v_div_fmas) and v_div_fixup; for operands far from the ends of the exponent range -- here |a| in [0.5, L + 1], |b| in
// [1 / (1.5 L), 1] -- the scale factors are 1, v_div_fmas is a plain fma and v_div_fixup passes the quotient through.  So the
// refined reciprocal is computed ONCE per cell and every step runs only the quotient and its two corrections: the same
// operations in the same order on the same values, i.e. the same bits, in five dependent instructions instead of ten.
__device__ __forceinline__ float refined_rcp(float b)
{
    const float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float div_by(float a, float b, float r)     // a / b, r = refined_rcp(b); see above for the operand ranges
{
    float q = a * r;
    float e = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e, r, q);
    e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, r, q);
}

// The cells that walk: obstacle cells (GPU:712) off the centre row and column (GPU:760-791: there the bound is computed and never
// applied).  Most of a map's cells do not, and a wave lasts as long as its longest ray: the walkers are compacted into a list first
// (one ballot and one atomic per wave; the order is irrelevant, every cell only touches itself) and k_raytracing runs on full waves.
__device__ __forceinline__ int ray_robot_index(int L) { return (L % 2 == 0) ? (int)(float)(L / 2 - 0.5) : (int)(float)(L / 2); }   // GPU:733, 739: float -> int

// The same pass over the cells takes G_Clear_maplowest (GPU:232-239) with it: the walks read a SNAPSHOT of the lowest scan points,
// written here next to the reset of the layer itself -- one launch and one pass over the layer less than clearing behind the walks
// (k_clear_lowest was 3.8-5.6 us of the node's 45 us frame).  It also zeroes the OTHER call parity's counter of walkers.
// One atomic per WORKGROUP of 1024 cells claims the list slots of its sixteen waves: with one per wave, a map where most waves hold
// a walker (C2 with the reference's reject filter: walkers all over the map) sent 5 600 atomics to one address and the kernel took
// 31 us for 7 MB of traffic.  (The order of the list is irrelevant: the walks only lower running minima.)
constexpr int kRayListThreads = 1024;
__global__ __launch_bounds__(kRayListThreads) void k_ray_list(LayerPtrs m, int L, int start0, int start1, float obstacle_threshold, int row0, int row1,
                                                              uint32_t* __restrict__ list, uint32_t* __restrict__ count, uint32_t* __restrict__ count_next,
                                                              float* __restrict__ lowest_snapshot)
{
    __shared__ uint32_t wave_cnt[kRayListThreads / 64];
    __shared__ uint32_t block_base;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool walks = false;
    if (i == 0) *count_next = 0u;
    if (i < L * L) {
        lowest_snapshot[i] = m.lowest[i];
        m.lowest[i] = 10.0f;
        const int cell_x = i / L, cell_y = i - cell_x * L;
        if (cell_x >= row0 && cell_x < row1 && m.traver[i] < obstacle_threshold && m.elevation[i] != kEmptyElevation) {    // GPU:712
            int ob0 = cell_x + L - start0; ob0 -= ob0 >= L ? L : 0;                              // GPU:672-675 (% L)
            int ob1 = cell_y + L - start1; ob1 -= ob1 >= L ? L : 0;
            const int robot_index = ray_robot_index(L);
            walks = ob0 != robot_index && ob1 != robot_index;
        }
    }
    const uint64_t mk = __ballot(walks);
    const int w = (int)(threadIdx.x >> 6);
    if (lane_id() == 0) wave_cnt[w] = (uint32_t)__popcll(mk);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int k = 0; k < kRayListThreads / 64; ++k) total += wave_cnt[k];
        block_base = total ? atomicAdd(count, total) : 0u;
    }
    __syncthreads();
    if (!walks) return;
    uint32_t base = block_base;
    for (int k = 0; k < w; ++k) base += wave_cnt[k];
    list[base + (uint32_t)__popcll(mk & lanemask_lt())] = (uint32_t)i;
}

// One ray is walked by G adjacent lanes, each a run of S crossings of the ray's MAJOR axis (the one with the larger |increment|):
// a wave lasts as long as its longest lane, the walkers fill a third of the chip's SIMDs with one wave each, and a 600-cell walk at
// ~30 instructions a step was the whole 30 us of this kernel.  A lane can start in the middle of a walk because the walk is a merge
// of two increasing sequences, the border distances dn_x(k) = (k + 1/2) inc_x / dir_x and dn_y(k) likewise (the bounds are
// half-integers, exact in float; the quotients are correctly rounded, so increasing, and a whole cell apart): after the step
// that makes the K-th major crossing the walk has made every minor crossing with dn_m(j) <= dn_M(K - 1) (equal distances step
// together, GPU:855), stands in cell (ob_M + K inc_M, ob_m + k_m inc_m) and remembers `later` = dn_M(K - 1).  k_m is found from an
// estimate corrected by evaluating dn_m itself, the very expression the walk compares.  Lane g walks from that state until its
// major index reaches (g + 1) S or the walk leaves the map; the lanes' bounds are folded by a minimum (the order is irrelevant).
template <int RD, int G>
__global__ __launch_bounds__(256) void k_raytracing(LayerPtrs m, int L, int start0, int start1, float sensor_z,
                                                   const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
                                                   const float* __restrict__ lowest_snapshot)
{
    static_assert(G >= 1 && G <= 64 && (G & (G - 1)) == 0, "lanes per ray: a power of two inside a wave");
    const uint32_t n_rays = *count;
    const int robot_index = ray_robot_index(L);
    const float robot_f = (float)robot_index;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t / G < n_rays; t += gridDim.x * blockDim.x) {   // a ray's G lanes stay together
        const int g = (int)(t % G);
        const int i = (int)list[t / G];
        const int cell_x = i / L, cell_y = i - cell_x * L;
        const float obstacle_ele = m.elevation[i];
        int ob0 = cell_x + L - start0; ob0 -= ob0 >= L ? L : 0;                                      // GPU:672-675 (% L)
        int ob1 = cell_y + L - start1; ob1 -= ob1 >= L ? L : 0;
        const float inc0 = (float)(ob0 - robot_index), inc1 = (float)(ob1 - robot_index);
        const int inc_x = inc0 > 0 ? 1 : -1, inc_y = inc1 > 0 ? 1 : -1;                              // (neither is 0: k_ray_list)
        const float dis = sqrtf(inc0 * inc0 + inc1 * inc1);                                          // GPU:793
        const float dir0 = inc0 / dis, dir1 = inc1 / dis;
        float threshold;                                                                             // GPU:798-802, double arithmetic
        if (fabsf(inc0) > fabsf(inc1)) { const double q = 0.5 / (double)inc0 * (double)inc1; threshold = (float)sqrt(0.5 * 0.5 + q * q); }
        else                           { const double q = 0.5 / (double)inc1 * (double)inc0; threshold = (float)sqrt(0.5 * 0.5 + q * q); }
        float bound_x = (float)inc_x / 2, bound_y = (float)inc_y / 2;                               // GPU:808-809
        const float rcp0 = refined_rcp(dir0), rcp1 = refined_rcp(dir1);
        float dir_num_x = div_by(bound_x, dir0, rcp0), dir_num_y = div_by(bound_y, dir1, rcp1), later = 0.0f;
        float restrict_ele = obstacle_ele;
        int c0 = ob0, c1 = ob1;
        // this lane's run of the walk
        const bool maj_x = fabsf(inc0) >= fabsf(inc1);
        const int obM = maj_x ? ob0 : ob1, incM = maj_x ? inc_x : inc_y;
        int cM_end = -2;                                                   // (never reached: the last lane walks off the map)
        if (G > 1) {
            const int nM = incM > 0 ? L - 1 - obM : obM;                   // major crossings that stay inside; crossing number nM leaves
            const int S = (nM + G) / G;                                    // ceil((nM + 1) / G)
            const int K = g * S;
            cM_end = g == G - 1 ? -2 : obM + (K + S) * incM;
            if (K > 0) {
                const int obm = maj_x ? ob1 : ob0, incm = maj_x ? inc_y : inc_x;
                const float dirM = maj_x ? dir0 : dir1, rcpM = maj_x ? rcp0 : rcp1, dirm = maj_x ? dir1 : dir0, rcpm = maj_x ? rcp1 : rcp0;
                const float fincM = (float)incM, fincm = (float)incm;
                const float vprev = div_by(((float)K - 0.5f) * fincM, dirM, rcpM);               // dn_M(K - 1): the step that got here
                const float dnM = div_by(((float)K + 0.5f) * fincM, dirM, rcpM);
                int j = (int)floorf(vprev * fabsf(dirm) - 0.5f);                                 // about the last j with dn_m(j) <= vprev
                j = j < -1 ? -1 : (j > 2 * L ? 2 * L : j);
                while (div_by(((float)(j + 1) + 0.5f) * fincm, dirm, rcpm) <= vprev) ++j;
                while (j >= 0 && div_by(((float)j + 0.5f) * fincm, dirm, rcpm) > vprev) --j;
                const int km = j + 1;
                const float bm = ((float)km + 0.5f) * fincm;
                const float dnm = div_by(bm, dirm, rcpm);
                const int cM = obM + K * incM, cm = obm + km * incm;
                const float bM = ((float)K + 0.5f) * fincM;
                c0 = maj_x ? cM : cm;              c1 = maj_x ? cm : cM;
                bound_x = maj_x ? bM : bm;         bound_y = maj_x ? bm : bM;
                dir_num_x = maj_x ? dnM : dnm;     dir_num_y = maj_x ? dnm : dnM;
                later = vprev;
            }
        }
        // GPU:819-880.  The reference's three branches (step in y / step in x / step in both when the two border distances tie)
        // are one straight-line body here: the crossed-cell test uses the smaller distance (dir_num_x when they tie, as in the
        // reference's last branch), and each axis advances under a select.  Every value is computed by the reference's own
        // expression (the two divisions per step through div_by, above); lanes of a wave do not serialise over the three variants.
        // The walk itself (c0, c1, the border distances) never depends on what it reads: the lowest scan point of a crossed cell only
        // lowers `restrict_ele`, a running minimum.  The walk runs RD steps ahead -- RD loads in flight, addresses clamped instead
        // of branched round -- and the bounds are folded in afterwards (a minimum: any order).
        bool inside = c0 >= 0 && c0 < L && c1 >= 0 && c1 < L;
        while (__ballot(inside) != 0) {                                   // wave-uniform; lanes that are done idle
            float low[RD]; int hc0[RD]; bool hit[RD];
#pragma unroll
            for (int u = 0; u < RD; ++u) {
                const bool step_y = dir_num_x > dir_num_y;                 // GPU:821
                const bool step_x = dir_num_x < dir_num_y;                 // GPU:838; neither: both axes (GPU:855)
                const float step = step_y ? dir_num_y : dir_num_x;
                hit[u] = inside && step - later > threshold && c0 != ob0 && c1 != ob1;       // GPU:823-830 and twins
                hc0[u] = c0;
                low[u] = lowest_snapshot[hit[u] ? (uint32_t)(c0 * L + c1) : 0u];
                later = inside ? step : later;
                const float nbx = bound_x + (float)inc_x, nby = bound_y + (float)inc_y;
                const float ndx = div_by(nbx, dir0, rcp0), ndy = div_by(nby, dir1, rcp1);
                if (inside && !step_y) { c0 += inc_x; bound_x = nbx; dir_num_x = ndx; }
                if (inside && !step_x) { c1 += inc_y; bound_y = nby; dir_num_y = ndy; }
                inside = inside && (unsigned)c0 < (unsigned)L && (unsigned)c1 < (unsigned)L;
                if (G > 1) inside = inside && (maj_x ? c0 : c1) != cM_end;
            }
#pragma unroll
            for (int u = 0; u < RD; ++u) {
                if (hit[u] && low[u] != 10.0f) {                           // GPU:681-689
                    const float x1 = (float)(hc0[u] - ob0), x2 = (float)hc0[u] - robot_f;            // GPU:691-706
                    const float e = low[u] + (sensor_z - low[u]) / x2 * x1;
                    if (e < restrict_ele) restrict_ele = e;
                }
            }
        }
#pragma unroll
        for (int o = 1; o < G; o <<= 1) {                                  // the ray's lanes: adjacent, all here
            const float other = __shfl_xor(restrict_ele, o);
            if (other < restrict_ele) restrict_ele = other;
        }
        if (g == 0 && obstacle_ele - 3 * sqrtf(m.variance[i]) > restrict_ele) m.elevation[i] = kEmptyElevation;   // GPU:884-885
    }
}

// AoS ingest (first half): the reference walks the PCL cloud on the host and copies x, y, z, r, g, b, intensity
// into seven stack arrays (SPB.cpp:160-169).  Here the point structs go to the device as they are and this kernel pulls the
// fields apart: byte offsets of four-byte fields inside a struct of `step` bytes (PointXYZRGBICT.hpp:28-46: x y z pad | rgb |
// covariance | intensity | travers, 32 bytes; PCL's rgb is the bytes b, g, r, a).
__global__ __launch_bounds__(256) void k_unpack_aos(const unsigned char* __restrict__ src, int n, int step, int ox, int oy, int oz, int oi, int orgb,
                                                    float4* __restrict__ xyzi, uint32_t* __restrict__ rgb)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned char* p = src + (size_t)i * step;
        float4 v;
        v.x = *reinterpret_cast<const float*>(p + ox); v.y = *reinterpret_cast<const float*>(p + oy);
        v.z = *reinterpret_cast<const float*>(p + oz); v.w = oi >= 0 ? *reinterpret_cast<const float*>(p + oi) : 0.0f;
        xyzi[i] = v;
        if (rgb) rgb[i] = *reinterpret_cast<const uint32_t*>(p + orgb) & 0x00ffffffu;      // a<<24 | r<<16 | g<<8 | b  ->  0x00RRGGBB
    }
}

hipError_t launch_unpack_aos(hipStream_t st, const void* src, int n, int step, int ox, int oy, int oz, int oi, int orgb, float4* xyzi, uint32_t* rgb)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_unpack_aos, dim3(grid_for(n, 256)), dim3(256), 0, st, static_cast<const unsigned char*>(src), n, step, ox, oy, oz, oi, orgb, xyzi, rgb);
    return hipGetLastError();
}

// list: L * L words (the walking cells); counts: two words, the walkers of even / odd calls (this call's is zero on entry: zeroed at
// allocation and by the call before); snapshot: L * L floats
hipError_t launch_raytracing(hipStream_t st, const LayerPtrs& m, int L, int start0, int start1, float sensor_z, float obstacle_threshold,
                             int row0, int row1, uint32_t* list, uint32_t* counts, int parity, float* lowest_snapshot, int depth, int lanes)
{
    const dim3 grid((L * L + kRayListThreads - 1) / kRayListThreads), block(256);
    uint32_t* count = counts + (parity & 1), *count_next = counts + ((parity + 1) & 1);
    hipLaunchKernelGGL(k_ray_list, grid, dim3(kRayListThreads), 0, st, m, L, start0, start1, obstacle_threshold, row0, row1, list, count, count_next, lowest_snapshot);
    auto k = lanes >= 16 ? (depth >= 8 ? k_raytracing<8, 16> : k_raytracing<4, 16>)
           : lanes >= 8  ? (depth >= 8 ? k_raytracing<8, 8>  : k_raytracing<4, 8>)
           : lanes >= 4  ? (depth >= 8 ? k_raytracing<8, 4>  : k_raytracing<4, 4>)
                         : (depth >= 8 ? k_raytracing<8, 1>  : k_raytracing<4, 1>);
    const int g_lanes = lanes >= 16 ? 16 : lanes >= 8 ? 8 : lanes >= 4 ? 4 : 1;
    const long long want = ((long long)L * L * g_lanes + 255) / 256;
    const dim3 ray_grid((unsigned)(want < 4096 ? want : 4096));           // the walkers are a fraction of the cells: grid-stride
    hipLaunchKernelGGL(k, ray_grid, block, 0, st, m, L, start0, start1, sensor_z, (const uint32_t*)list, (const uint32_t*)count, (const float*)lowest_snapshot);
    return hipGetLastError();
}

hipError_t launch_init(hipStream_t st, const LayerPtrs& m, int cells, int clear_lowest)
{
    hipLaunchKernelGGL(k_init, dim3(grid_for(cells, 256)), dim3(256), 0, st, m, cells, clear_lowest);
    return hipGetLastError();
}

hipError_t launch_clear_strip(hipStream_t st, const LayerPtrs& m, int L, int start, int count, int is_row)
{
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_clear_strip, dim3(grid_for((long long)L * count, 256)), dim3(256), 0, st, m, L, start, count, is_row);
    return hipGetLastError();
}

hipError_t launch_dense_variance(hipStream_t st, float* variance, int cells, int n_pending, const float* pending,
                                 int apply_floor, float var_floor)
{
    float p[4] = {0, 0, 0, 0};
    for (int i = 0; i < n_pending && i < 4; ++i) p[i] = pending[i];
    hipLaunchKernelGGL(k_dense_variance, dim3(grid_for(cells, 256)), dim3(256), 0, st, variance, cells, n_pending,
                       p[0], p[1], p[2], p[3], apply_floor, var_floor);
    return hipGetLastError();
}

hipError_t launch_update_height(hipStream_t st, float* elevation, int cells, float dz)
{
    hipLaunchKernelGGL(k_update_height, dim3(grid_for(cells, 256)), dim3(256), 0, st, elevation, cells, dz);
    return hipGetLastError();
}

hipError_t launch_map_feature(hipStream_t st, const float* elevation, float* traver, float* rough, float* slope,
                              int L, float res, int sx, int sy, int row0, int row1)
{
    const int tiles = (L + 15) / 16;
    hipLaunchKernelGGL(k_map_feature, dim3(tiles * tiles), dim3(256), 0, st, elevation, traver, rough, slope, L, res, sx, sy, row0, row1);
    return hipGetLastError();
}

hipError_t launch_copy_list(hipStream_t st, const CopyList& l)
{
    if (l.n <= 0) return hipSuccess;
    size_t longest = 0;
    for (int i = 0; i < l.n; ++i) longest = l.piece[i].bytes > longest ? l.piece[i].bytes : longest;
    const size_t blocks = (longest / 16 + 255) / 256;                   // one 16-byte store per lane: the link wants many stores in flight
    hipLaunchKernelGGL(k_copy_list, dim3((unsigned)(blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks)), (unsigned)l.n), dim3(256), 0, st, l);
    return hipGetLastError();
}

hipError_t launch_export_gridmap(hipStream_t st, const void* src, const float* elevation, float* dst, int L, int is_int)
{
    hipLaunchKernelGGL(k_export_gridmap, dim3(grid_for((long long)L * L, 256)), dim3(256), 0, st, src, elevation, dst, L, is_int);
    return hipGetLastError();
}

} // namespace gem
