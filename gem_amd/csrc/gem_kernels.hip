// gem_kernels.hip -- gfx950 kernels of the GEM point-cloud -> elevation-grid hot path.
//
// Pipeline for one cloud (replaces G_pointsprocess + G_fuse, gpu_process.cu:384-455 / 477-537):
//
//   k_bin   one WAVE per "unit" of 64*IPT consecutive points: coalesced float4 XYZI loads,
//           projection + variance + 2.5-D binning in registers, then a wave-local STABLE
//           counting sort of the unit's points by map tile (32x32 or 64x64 cells).  Emits
//           16-byte records {cell-in-tile, h, var, src} grouped by tile inside the unit's own
//           slice of the record arena (no global scan needed) and a (start,count) descriptor
//           per (tile, unit).
//   k_fuse  one workgroup per tile: stages the tile's elevation/variance in LDS, gathers the
//           tile's records unit by unit (ascending unit == ascending input index), stable-sorts
//           them by cell in LDS (per-wave histograms + ballot ranking), then one lane per cell
//           walks its points in input order applying the reference's non-associative
//           recurrence, and the tile is written back once with the variance floor applied.
//
// The reference's fusion is order dependent (variance floor inside the loop, Mahalanobis branch),
// so every step above preserves ascending point index per cell; there are no float atomics.
//
// Built with -ffp-contract=off (see gem_device.hpp).
#include "gem_kernels.hpp"

namespace gem {

// ------------------------------------------------------------------------------------------
// small wave / block helpers (wave = 64 lanes)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ uint64_t lanemask_lt()
{
    const int l = lane_id();
    return l == 0 ? 0ull : (~0ull >> (64 - l));
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (l >= d) v += t;
    }
    return v;
}

// exclusive scan over a 256-thread block; scratch = 5 uint32 in LDS.  Returns prefix, *total = block sum.
__device__ __forceinline__ uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* scratch, uint32_t* total)
{
    const int w = (int)(threadIdx.x >> 6);
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane_id() == 63) scratch[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t s = scratch[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();     // scratch may be reused by the caller right away
    *total = tot;
    return base + inc - v;
}

// Stable rank of each valid lane among the lanes holding the same key, plus a running base kept
// in an LDS table: pos = table[key]++ applied in lane order.  Iterates once per DISTINCT key in
// the wave (<= 64), independent of how many lanes share a key.  All table traffic goes through
// the group's leader lane, so there is no intra-wave race.
__device__ __forceinline__ uint32_t wave_stable_place(bool valid, uint32_t key, uint32_t* table)
{
    const int l = lane_id();
    const uint64_t lt = lanemask_lt();
    uint64_t remaining = __ballot(valid);
    uint32_t pos = 0;
    while (remaining) {                                   // wave-uniform loop
        const int leader = __ffsll((unsigned long long)remaining) - 1;
        const uint32_t k0 = (uint32_t)__shfl((int)key, leader, 64);
        const bool mine = valid && key == k0;
        const uint64_t m = __ballot(mine);
        uint32_t b = 0;
        if (l == leader) { b = table[k0]; table[k0] = b + (uint32_t)__popcll(m); }
        b = (uint32_t)__shfl((int)b, leader, 64);
        if (mine) pos = b + (uint32_t)__popcll(m & lt);
        remaining &= ~m;
    }
    return pos;
}

// ------------------------------------------------------------------------------------------
// k_project : Process_points' kernel (GPU:384-455) for the GEM-compatible host-array entry.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_project(FrameConst fc, int n, float* __restrict__ x, float* __restrict__ y,
                                                 float* __restrict__ z, const int* __restrict__ orig, int write_back,
                                                 int* __restrict__ map_idx, float* __restrict__ var,
                                                 float* __restrict__ xt, float* __restrict__ yt, float* __restrict__ zt)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const Projected r = project_point(fc, x[i], y[i], z[i], orig ? orig[i] : i);
        map_idx[i] = r.cell; var[i] = r.var; xt[i] = r.xt; yt[i] = r.yt; zt[i] = r.h;
        if (write_back && !r.accepted) { x[i] = -1.0f; y[i] = -1.0f; z[i] = -1.0f; }   // GPU:443-446
    }
}

// ------------------------------------------------------------------------------------------
// k_bin
// ------------------------------------------------------------------------------------------
template <int IPT, int SRC, int TS, bool BATCH>
__global__ __launch_bounds__(64) void k_bin(BinArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_cnt[];      // [T]
    constexpr int TE = 1 << TS;
    constexpr int U = 64 * IPT;
    const int lane = lane_id();
    const int unit = (int)blockIdx.x;
    const int T = a.T;

    // which sweep does this unit belong to, and which points does it cover
    int sweep = 0;
    long long base, sweep_begin = 0, sweep_end = a.n;
    if (BATCH) {
        int lo = 0, hi = a.n_sweeps;                 // sweep_unit0[lo] <= unit < sweep_unit0[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.sweep_unit0[mid] <= unit) lo = mid; else hi = mid; }
        sweep = lo;
        sweep_begin = a.sweep_first[sweep]; sweep_end = a.sweep_first[sweep + 1];
        base = sweep_begin + (long long)(unit - a.sweep_unit0[sweep]) * U;
    } else {
        base = (long long)unit * U;
    }
    const FrameConst fc = BATCH ? a.frames[sweep] : a.frame0;
    const long long left = sweep_end - base;
    const int npts = left < U ? (int)left : U;

    for (int t = lane; t < T; t += 64) lds_cnt[t] = 0;
    __syncthreads();

    uint32_t tile[IPT], cl[IPT], src[IPT];
    float hh[IPT], vv[IPT];
    uint32_t n_binned = 0;

#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int o = j * 64 + lane;
        tile[j] = (uint32_t)kInvalidTile; cl[j] = 0; hh[j] = 0.0f; vv[j] = 0.0f; src[j] = 0;
        if (o < npts) {
            const long long i = base + o;
            int cell; float h, v; bool colour_ok = false;
            if (SRC == 0) {
                const float4 p = a.xyzi[i];
                const Projected r = project_point(fc, p.x, p.y, p.z, a.orig ? a.orig[i] : (int)(i - sweep_begin));
                cell = r.cell; h = r.h; v = r.var;
                if (a.rgb) {
                    const uint32_t c = a.rgb[i];
                    colour_ok = ((c >> 16) & 0xff) != 0 && ((c >> 8) & 0xff) != 0 && (c & 0xff) != 0 && p.w != 0.0f;
                }
            } else {
                cell = a.f_index[i]; h = a.f_height[i]; v = a.f_var[i];
                if (cell >= fc.L * fc.L) cell = -1;
                if (a.f_R) colour_ok = a.f_R[i] != 0 && a.f_G[i] != 0 && a.f_B[i] != 0 && a.f_I[i] != 0.0f;
            }
            // GPU:482: "point_index[i] != map_index || points_h[i] == -1" -> the point is skipped
            if (cell >= 0 && h != -1.0f) {
                const int row = cell / fc.L, col = cell - row * fc.L;
                if (row >= fc.row0 && row < fc.row1) {
                    tile[j] = (uint32_t)((row >> TS) * a.tiles_per_row + (col >> TS));
                    cl[j] = (uint32_t)(((row & (TE - 1)) << TS) | (col & (TE - 1))) | (colour_ok ? 0x80000000u : 0u);
                    hh[j] = h; vv[j] = v; src[j] = (uint32_t)i;
                    atomicAdd(&lds_cnt[tile[j]], 1u);
                    ++n_binned;
                }
            }
        }
    }
    __syncthreads();

    // wave-level exclusive scan of the T tile counts (blocked: lane owns K consecutive tiles)
    const int K = (T + 63) / 64;
    const int t0 = lane * K;
    uint32_t local = 0;
    for (int k = 0; k < K; ++k) { const int t = t0 + k; if (t < T) local += lds_cnt[t]; }
    uint32_t run = wave_inclusive_scan(local) - local;
    uint32_t* seg = a.seg;
    for (int k = 0; k < K; ++k) {
        const int t = t0 + k;
        if (t < T) {
            const uint32_t c = lds_cnt[t];
            lds_cnt[t] = run;                                       // becomes the running base of tile t
            seg[(size_t)t * a.B + unit] = run | (c << 16);          // start (16 bit) | count (16 bit)
            run += c;
        }
    }
    __syncthreads();

    // stable placement, chunk by chunk in input order, into this unit's slice of the arena
    uint4* rec = a.rec + (size_t)unit * U;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const bool valid = tile[j] != (uint32_t)kInvalidTile;
        const uint32_t pos = wave_stable_place(valid, tile[j], lds_cnt);
        if (valid) rec[pos] = make_uint4(cl[j], __float_as_uint(hh[j]), __float_as_uint(vv[j]), src[j]);
    }

    if (a.counters) {
        uint32_t s = n_binned;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d, 64);
        if (lane == 0 && s) atomicAdd(&a.counters[0], (unsigned long long)s);
    }
}

// ------------------------------------------------------------------------------------------
// k_fuse
// ------------------------------------------------------------------------------------------
// Thread t owns cells {t + 256*q} of the tile for the whole kernel: their (elevation, variance)
// live in registers from the single read to the single write-back.  LDS holds the per-batch sort:
//   wc[4][CELLS] u32 | cstart[CELLS] u16 | ccount[CELLS] u16 | s_h[PB] f32 | s_v[PB] f32
//   | s_src[PB] u32 | scratch[8] u32 | touched[CELLS/32] u32 | prefix[Bpad+1] u32 | ustart[Bpad] u16
template <int TS, int R, int ATTR>
__global__ __launch_bounds__(256) void k_fuse(FuseArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int TE = 1 << TS;
    constexpr int CELLS = TE * TE;
    constexpr int PB = 256 * R;
    constexpr int CPT = CELLS / 256;                 // cells per thread

    uint32_t* wc      = reinterpret_cast<uint32_t*>(lds_raw);              // [4][CELLS]
    uint16_t* cstart  = reinterpret_cast<uint16_t*>(wc + 4 * CELLS);
    uint16_t* ccount  = cstart + CELLS;
    float*    s_h     = reinterpret_cast<float*>(ccount + CELLS);
    float*    s_v     = s_h + PB;
    uint32_t* s_src   = reinterpret_cast<uint32_t*>(s_v + PB);
    uint32_t* scratch = s_src + PB;
    uint32_t* touched = scratch + 8;
    uint32_t* prefix  = touched + CELLS / 32;
    uint16_t* ustart  = reinterpret_cast<uint16_t*>(prefix + a.Bpad + 1);

    const int tid = (int)threadIdx.x;
    const int lane = lane_id();
    const int w = tid >> 6;
    const int tile = (int)blockIdx.x;
    const int tr = tile / a.tiles_per_row, tc = tile - tr * a.tiles_per_row;
    const int row_base = tr << TS, col_base = tc << TS;
    const int L = a.L;
    const uint32_t* seg = a.seg + (size_t)tile * a.B_total;

    // ---- 0. does this tile receive any point at all? ---------------------------------------------
    if (!a.dense) {
        uint32_t any = 0;
        for (int u = tid; u < a.B_total; u += 256) any |= seg[u] >> 16;
        if (!__syncthreads_or((int)any)) return;
    }

    // ---- 1. the single read of the tile ---------------------------------------------------------
    float ce[CPT], cs[CPT];
    bool  owned[CPT];
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        const int c = tid + 256 * q;
        const int row = row_base + (c >> TS), col = col_base + (c & (TE - 1));
        owned[q] = row < a.row1 && row >= a.row0 && col < L;
        ce[q] = kEmptyElevation; cs[q] = kInitVariance;
        if (owned[q]) {
            const size_t g = (size_t)row * L + col;
            ce[q] = a.elevation[g]; cs[q] = a.variance[g];
        }
    }

    for (int sweep = 0; sweep < a.n_sweeps; ++sweep) {
        const int ub = a.sweep_unit0 ? a.sweep_unit0[sweep] : 0;
        const int ue = a.sweep_unit0 ? a.sweep_unit0[sweep + 1] : a.B_total;
        const int B = ue - ub;

        // ---- 2. Mapvar_update increments queued before this sweep (GPU:540-547) ------------------
#pragma unroll
        for (int q = 0; q < CPT; ++q) {
            if (sweep == 0)
                for (int k = 0; k < a.n_pending; ++k) if (cs[q] != kInitVariance) cs[q] += a.pending[k];
            if (a.var_updates) { if (cs[q] != kInitVariance) cs[q] += a.var_updates[sweep]; }
        }

        // ---- 3. which units feed this tile: exclusive scan of their counts ----------------------
        const int UPT = (B + 255) / 256;
        const int u0 = tid * UPT;
        uint32_t local = 0;
        for (int k = 0; k < UPT; ++k) { const int u = u0 + k; if (u < B) local += seg[ub + u] >> 16; }
        uint32_t P;
        uint32_t run = block_exclusive_scan_256(local, scratch, &P);
        if (P != 0) {                                                   // block-uniform
            for (int k = 0; k < UPT; ++k) {
                const int u = u0 + k;
                if (u < B) { const uint32_t s = seg[ub + u]; prefix[u] = run; ustart[u] = (uint16_t)(s & 0xffffu); run += s >> 16; }
            }
            if (tid == 255) prefix[B] = P;
            if (tid < CELLS / 32) touched[tid] = 0;
            __syncthreads();
        }

        // ---- 4. batches of PB points, in input order --------------------------------------------
        for (uint32_t bbase = 0; bbase < P; bbase += PB) {
            const uint32_t Pb = min((uint32_t)PB, P - bbase);
            const uint32_t span = ((Pb + 255u) / 256u) * 64u;            // contiguous k-range per wave
            const uint32_t nchunk = span / 64u;                          // <= R

            for (int c = tid; c < 4 * CELLS; c += 256) wc[c] = 0;
            __syncthreads();

            uint32_t cell[R], srcv[R]; float hh[R], vv[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                cell[r] = 0xffffffffu; srcv[r] = 0; hh[r] = 0.0f; vv[r] = 0.0f;
                const uint32_t kl = (uint32_t)w * span + (uint32_t)r * 64u + (uint32_t)lane;
                if ((uint32_t)r < nchunk && kl < Pb) {
                    const uint32_t k = bbase + kl;
                    int lo = 0, hi = B;                                  // prefix[lo] <= k < prefix[hi]
                    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (prefix[mid] <= k) lo = mid; else hi = mid; }
                    const size_t ridx = (size_t)(ub + lo) * a.U + ustart[lo] + (k - prefix[lo]);
                    const uint4 rr = a.rec[ridx];
                    cell[r] = rr.x; hh[r] = __uint_as_float(rr.y); vv[r] = __uint_as_float(rr.z); srcv[r] = rr.w;
                    atomicAdd(&wc[w * CELLS + (rr.x & 0xffffu)], 1u);
                }
            }
            __syncthreads();

            // exclusive scan in (cell-major, wave-minor) order
            {
                const int c0 = tid * CPT;
                uint32_t loc = 0;
#pragma unroll
                for (int q = 0; q < CPT; ++q)
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) loc += wc[ww * CELLS + c0 + q];
                uint32_t tot;
                uint32_t rn = block_exclusive_scan_256(loc, scratch, &tot);
#pragma unroll
                for (int q = 0; q < CPT; ++q) {
                    const int c = c0 + q;
                    const uint32_t st = rn;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) { const uint32_t x = wc[ww * CELLS + c]; wc[ww * CELLS + c] = rn; rn += x; }
                    cstart[c] = (uint16_t)st;
                    ccount[c] = (uint16_t)(rn - st);
                    if (a.counters && rn != st) atomicOr(&touched[c >> 5], 1u << (c & 31));
                }
            }
            __syncthreads();

            // stable placement: each wave walks its chunks in order
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if ((uint32_t)r < nchunk) {                               // wave-uniform
                    const bool valid = cell[r] != 0xffffffffu;
                    const uint32_t pos = wave_stable_place(valid, cell[r] & 0xffffu, wc + w * CELLS);
                    if (valid) {
                        s_h[pos] = hh[r]; s_v[pos] = vv[r];
                        if (ATTR) s_src[pos] = (srcv[r] & 0x7fffffffu) | (cell[r] & 0x80000000u);
                    }
                }
            }
            __syncthreads();

            // one lane per cell walks its points in input order (GPU:480-531)
#pragma unroll
            for (int q = 0; q < CPT; ++q) {
                const int c = tid + 256 * q;
                const uint32_t cnt = ccount[c];
                if (cnt) {
                    float e = ce[q], s = cs[q];
                    const uint32_t st = cstart[c];
                    uint32_t last = 0xffffffffu;
                    for (uint32_t p = st; p < st + cnt; ++p) {
                        const bool taken = fuse_step(e, s, s_h[p], s_v[p], a.mahal, a.var_floor);
                        if (ATTR) { const uint32_t sv = s_src[p]; if (taken && (sv & 0x80000000u)) last = sv & 0x7fffffffu; }
                    }
                    ce[q] = e; cs[q] = s;
                    if (ATTR && last != 0xffffffffu) {
                        // colour / intensity of the last taken point with all four non-zero (GPU:487-494)
                        const int row = row_base + (c >> TS), col = col_base + (c & (TE - 1));
                        const size_t g = (size_t)row * L + col;
                        if (ATTR == 1) {
                            const uint32_t cc = a.rgb[last];
                            a.intensity[g] = a.xyzi[last].w;
                            a.colorR[g] = (int)((cc >> 16) & 0xff); a.colorG[g] = (int)((cc >> 8) & 0xff); a.colorB[g] = (int)(cc & 0xff);
                        } else {
                            a.intensity[g] = a.f_I[last];
                            a.colorR[g] = a.f_R[last]; a.colorG[g] = a.f_G[last]; a.colorB[g] = a.f_B[last];
                        }
                    }
                }
            }
            __syncthreads();
        }

        // ---- 5. variance floor at the end of every Fuse (GPU:533-534), on every cell -------------
#pragma unroll
        for (int q = 0; q < CPT; ++q) if (cs[q] < a.var_floor) cs[q] = a.var_floor;

        if (a.counters && P != 0) {
            if (tid < CELLS / 32) {
                const uint32_t n = (uint32_t)__popc(touched[tid]);
                if (n) atomicAdd(&a.counters[1], (unsigned long long)n);
            }
            __syncthreads();
        }
    }

    // ---- 6. the single write-back of the tile ----------------------------------------------------
#pragma unroll
    for (int q = 0; q < CPT; ++q) {
        if (owned[q]) {
            const int c = tid + 256 * q;
            const size_t g = (size_t)(row_base + (c >> TS)) * L + col_base + (c & (TE - 1));
            a.elevation[g] = ce[q];
            a.variance[g] = cs[q];
        }
    }
}

// ------------------------------------------------------------------------------------------
// dense / state kernels
// ------------------------------------------------------------------------------------------
// G_Init_map (GPU:198-214) and G_Clear_allmap (GPU:216-230; does not touch map_lowest)
__global__ __launch_bounds__(256) void k_init(LayerPtrs m, int cells, int clear_lowest)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += gridDim.x * blockDim.x) {
        m.intensity[i] = 0.0f; m.elevation[i] = kEmptyElevation; m.variance[i] = kInitVariance;
        m.traver[i] = -10.0f;
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
// host-side launchers
// ------------------------------------------------------------------------------------------
static inline int grid_for(long long work, int block, int cap = 2048)
{
    long long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

hipError_t launch_project(hipStream_t st, const FrameConst& fc, int n, float* x, float* y, float* z, const int* orig,
                          int write_back, int* map_idx, float* var, float* xt, float* yt, float* zt)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_project, dim3(grid_for(n, 256)), dim3(256), 0, st, fc, n, x, y, z, orig, write_back, map_idx, var, xt, yt, zt);
    return hipGetLastError();
}

template <int IPT, int SRC>
static hipError_t launch_bin_ts(hipStream_t st, const BinArgs& a, int ts)
{
    const size_t lds = (size_t)a.T * sizeof(uint32_t);
    const bool batch = a.n_sweeps > 1;
    if (ts == 5) {
        if (batch) hipLaunchKernelGGL((k_bin<IPT, SRC, 5, true>),  dim3(a.B), dim3(64), lds, st, a);
        else       hipLaunchKernelGGL((k_bin<IPT, SRC, 5, false>), dim3(a.B), dim3(64), lds, st, a);
    } else {
        if (batch) hipLaunchKernelGGL((k_bin<IPT, SRC, 6, true>),  dim3(a.B), dim3(64), lds, st, a);
        else       hipLaunchKernelGGL((k_bin<IPT, SRC, 6, false>), dim3(a.B), dim3(64), lds, st, a);
    }
    return hipGetLastError();
}

hipError_t launch_bin(hipStream_t st, const BinArgs& a, int ipt, int src, int ts)
{
    if (a.B <= 0) return hipSuccess;
    if (src == 0) {
        switch (ipt) {
        case 1:  return launch_bin_ts<1, 0>(st, a, ts);
        case 2:  return launch_bin_ts<2, 0>(st, a, ts);
        default: return launch_bin_ts<4, 0>(st, a, ts);
        }
    }
    switch (ipt) {
    case 1:  return launch_bin_ts<1, 1>(st, a, ts);
    case 2:  return launch_bin_ts<2, 1>(st, a, ts);
    default: return launch_bin_ts<4, 1>(st, a, ts);
    }
}

size_t fuse_lds_bytes(int ts, int r, int bpad)
{
    const size_t cells = (size_t)1 << (2 * ts);
    const size_t pb = 256 * (size_t)r;
    size_t b = cells * 4 * 4            // wc
             + cells * 2 * 2            // cstart, ccount
             + pb * 4 * 3               // s_h, s_v, s_src
             + 8 * 4                    // scratch
             + cells / 32 * 4           // touched
             + ((size_t)bpad + 1) * 4   // prefix
             + (size_t)bpad * 2;        // ustart
    return (b + 15) & ~(size_t)15;
}

template <int TS, int R>
static hipError_t launch_fuse_attr(hipStream_t st, const FuseArgs& a, int attr, size_t lds)
{
    // more than 64 KiB of dynamic LDS needs an explicit opt-in, once per kernel
    static size_t configured[3] = {0, 0, 0};
    if (lds > 64 * 1024 && lds > configured[attr]) {
        const void* fn = attr == 0 ? (const void*)k_fuse<TS, R, 0> : attr == 1 ? (const void*)k_fuse<TS, R, 1> : (const void*)k_fuse<TS, R, 2>;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured[attr] = lds;
    }
    if (attr == 0)      hipLaunchKernelGGL((k_fuse<TS, R, 0>), dim3(a.T), dim3(256), lds, st, a);
    else if (attr == 1) hipLaunchKernelGGL((k_fuse<TS, R, 1>), dim3(a.T), dim3(256), lds, st, a);
    else                hipLaunchKernelGGL((k_fuse<TS, R, 2>), dim3(a.T), dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_fuse(hipStream_t st, const FuseArgs& a, int ts, int attr)
{
    if (a.T <= 0) return hipSuccess;
    if (ts == 5) return launch_fuse_attr<5, kFuseR32>(st, a, attr, fuse_lds_bytes(5, kFuseR32, a.Bpad));
    return launch_fuse_attr<6, kFuseR64>(st, a, attr, fuse_lds_bytes(6, kFuseR64, a.Bpad));
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

hipError_t launch_export_gridmap(hipStream_t st, const void* src, const float* elevation, float* dst, int L, int is_int)
{
    hipLaunchKernelGGL(k_export_gridmap, dim3(grid_for((long long)L * L, 256)), dim3(256), 0, st, src, elevation, dst, L, is_int);
    return hipGetLastError();
}

} // namespace gem
