// Micro-benchmark (MI355X): what do device-scope integer atomics to scattered map cells cost at the
// size of one LiDAR sweep?  Decides between the per-cell slot design and the tile-sort design.
//   hipcc --offload-arch=gfx950 -O3 -o atomics tools/ubench/atomics.hip && ./atomics
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_empty(int) {}

__global__ void k_copy16(const float4* __restrict__ in, uint4* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float4 p = in[i]; out[i] = make_uint4(__float_as_uint(p.x), __float_as_uint(p.y), __float_as_uint(p.z), i); }
}

template <int SCOPE>
__global__ void k_atomic_ret(const int* __restrict__ cell, unsigned* __restrict__ count, unsigned* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int c = cell[i];
        if (c >= 0) out[i] = __hip_atomic_fetch_add(&count[c], 1u, __ATOMIC_RELAXED, SCOPE);
    }
}

__global__ void k_atomic_noret(const int* __restrict__ cell, unsigned* __restrict__ count, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const int c = cell[i]; if (c >= 0) __hip_atomic_fetch_add(&count[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

// the slot kernel: load point, (fake) projection, returning atomic, scattered 16-byte slot write
template <int K>
__global__ void k_slot(const float4* __restrict__ pts, const int* __restrict__ cell, unsigned* __restrict__ count,
                       uint4* __restrict__ slots, unsigned* __restrict__ ovf, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float4 p = pts[i];
        const int c = cell[i];
        if (c >= 0) {
            const unsigned r = __hip_atomic_fetch_add(&count[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r < K) slots[(size_t)c * K + r] = make_uint4(__float_as_uint(p.z), __float_as_uint(p.w), (unsigned)i, 0u);
            else atomicAdd(ovf, 1u);
        }
    }
}

// the per-cell kernel: dense read of count, gather of the cell's slot records, tiny index sort,
// serial chain (three divisions per step like the Kalman recurrence), read-modify-write of 2 layers
template <int K>
__global__ void k_cell(unsigned* __restrict__ count, const uint4* __restrict__ slots, float* __restrict__ elev,
                       float* __restrict__ var, int cells)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cells) return;
    const unsigned n = count[c];
    if (n == 0) return;
    count[c] = 0;
    uint4 r[K];
    const unsigned m = n < K ? n : K;
#pragma unroll
    for (int j = 0; j < K; ++j) if (j < m) r[j] = slots[(size_t)c * K + j];
    float e = elev[c], s = var[c];
    // selection in index order without dynamic register indexing
    unsigned last = 0;
    for (unsigned t = 0; t < m; ++t) {
        unsigned best = 0xffffffffu; float h = 0, v = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) if (j < m) { const unsigned idx = r[j].z + 1u; if (idx > last && idx < best) { best = idx; h = __uint_as_float(r[j].x); v = __uint_as_float(r[j].y); } }
        last = best;
        const float sf = s < 1e-4f ? 1e-4f : s;
        const float mm = fabsf(h - e) / sqrtf(sf);
        const float en = (sf * h + v * e) / (sf + v);
        const float sn = (v * sf) / (v + sf);
        if (e == -10.0f || (mm > 5.0f && e < h)) { e = h; s = v; } else if (mm <= 5.0f) { e = en; s = sn; } else s = sf;
    }
    elev[c] = e; var[c] = s < 1e-4f ? 1e-4f : s;
}

template <typename F>
static void timeit(const char* name, hipStream_t st, int reps, F launch)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) launch(nullptr, nullptr);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) launch(nullptr, nullptr);
    CK(hipEventRecord(b, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    // per-dispatch time stamps
    std::vector<float> d;
    hipEvent_t s, e; CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
    for (int i = 0; i < 50; ++i) { launch(s, e); CK(hipStreamSynchronize(st)); float t; CK(hipEventElapsedTime(&t, s, e)); d.push_back(t * 1e3f); }
    std::sort(d.begin(), d.end());
    printf("%-34s back-to-back %7.2f us/launch   dispatch median %6.2f us  min %6.2f us\n", name, ms * 1e3f / reps, d[d.size() / 2], d[0]);
}

int main()
{
    const int L = 600, N = 64 * 2048, cells = L * L;
    std::vector<int> cell(N);
    std::vector<float4> pts(N);
    srand(2);
    int inmap = 0;
    for (int b = 0; b < 64; ++b)
        for (int a = 0; a < 2048; ++a) {
            const double el = (-24.8 + 26.8 * b / 63.0) * M_PI / 180.0;
            double r = el < 0 ? 1.73 / tan(-el) : 80.0; if (r > 80) r = 80;
            r += 0.02 * ((rand() % 2001) / 1000.0 - 1.0);
            const double az = 2 * M_PI * a / 2048.0;
            const float x = (float)(r * cos(az) * cos(el)) + 0.3f, y = (float)(r * sin(az) * cos(el)) - 0.2f;
            const int ix = (int)(300.0f - x / 0.05f), iy = (int)(300.0f - y / 0.05f);
            const int i = b * 2048 + a;
            pts[i] = make_float4(x, y, 0.01f * (rand() % 100), 3e-4f + 1e-6f * (rand() % 100));
            cell[i] = (ix >= 0 && ix < L && iy >= 0 && iy < L) ? ix * L + iy : -1;
            inmap += cell[i] >= 0;
        }
    { std::vector<int> h(cells, 0); int mx = 0, touched = 0; for (int c : cell) if (c >= 0) { if (h[c]++ == 0) ++touched; mx = std::max(mx, h[c]); }
      printf("points %d, in map %d, touched cells %d, max per cell %d\n", N, inmap, touched, mx); }

    hipStream_t st; CK(hipStreamCreate(&st));
    int* d_cell; float4* d_pts; unsigned *d_count, *d_out, *d_ovf; uint4 *d_slots, *d_o16; float *d_e, *d_v;
    constexpr int K = 16;
    CK(hipMalloc(&d_cell, N * 4)); CK(hipMalloc(&d_pts, N * 16)); CK(hipMalloc(&d_count, cells * 4)); CK(hipMalloc(&d_out, N * 4));
    CK(hipMalloc(&d_ovf, 4)); CK(hipMalloc(&d_slots, (size_t)cells * K * 16)); CK(hipMalloc(&d_o16, N * 16));
    CK(hipMalloc(&d_e, cells * 4)); CK(hipMalloc(&d_v, cells * 4));
    CK(hipMemcpy(d_cell, cell.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pts, pts.data(), N * 16, hipMemcpyHostToDevice));
    CK(hipMemset(d_count, 0, cells * 4)); CK(hipMemset(d_ovf, 0, 4));
    { std::vector<float> e(cells, -10.0f); CK(hipMemcpy(d_e, e.data(), cells * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_v, e.data(), cells * 4, hipMemcpyHostToDevice)); }

    const dim3 gp(N / 256), gc((cells + 255) / 256), blk(256);
#define L_(k, grid, ...) [&](hipEvent_t s, hipEvent_t e) { if (s) hipExtLaunchKernelGGL(k, grid, blk, 0, st, s, e, 0, __VA_ARGS__); else hipLaunchKernelGGL(k, grid, blk, 0, st, __VA_ARGS__); }
    timeit("empty (1 block)", st, 500, L_(k_empty, dim3(1), 0));
    timeit("empty (512 blocks)", st, 500, L_(k_empty, gp, 0));
    timeit("copy16 load->store", st, 500, L_(k_copy16, gp, d_pts, d_o16, N));
    timeit("atomic returning, agent", st, 500, L_((k_atomic_ret<__HIP_MEMORY_SCOPE_AGENT>), gp, d_cell, d_count, d_out, N));
    timeit("atomic returning, workgroup scope", st, 500, L_((k_atomic_ret<__HIP_MEMORY_SCOPE_WORKGROUP>), gp, d_cell, d_count, d_out, N));
    timeit("atomic non-returning, agent", st, 500, L_(k_atomic_noret, gp, d_cell, d_count, N));
    // slot pipeline: k_slot then k_cell (k_cell resets the counts)
    CK(hipMemset(d_count, 0, cells * 4));
    auto both = [&](hipEvent_t s, hipEvent_t e) {
        if (s) { hipExtLaunchKernelGGL((k_slot<K>), gp, blk, 0, st, s, e, 0, d_pts, d_cell, d_count, d_slots, d_ovf, N); hipLaunchKernelGGL((k_cell<K>), gc, blk, 0, st, d_count, d_slots, d_e, d_v, cells); }
        else { hipLaunchKernelGGL((k_slot<K>), gp, blk, 0, st, d_pts, d_cell, d_count, d_slots, d_ovf, N); hipLaunchKernelGGL((k_cell<K>), gc, blk, 0, st, d_count, d_slots, d_e, d_v, cells); }
    };
    timeit("k_slot + k_cell (dispatch = k_slot)", st, 500, both);
    auto both2 = [&](hipEvent_t s, hipEvent_t e) {
        if (s) { hipLaunchKernelGGL((k_slot<K>), gp, blk, 0, st, d_pts, d_cell, d_count, d_slots, d_ovf, N); hipExtLaunchKernelGGL((k_cell<K>), gc, blk, 0, st, s, e, 0, d_count, d_slots, d_e, d_v, cells); }
        else { hipLaunchKernelGGL((k_slot<K>), gp, blk, 0, st, d_pts, d_cell, d_count, d_slots, d_ovf, N); hipLaunchKernelGGL((k_cell<K>), gc, blk, 0, st, d_count, d_slots, d_e, d_v, cells); }
    };
    timeit("k_slot + k_cell (dispatch = k_cell)", st, 500, both2);
    unsigned ovf; CK(hipMemcpy(&ovf, d_ovf, 4, hipMemcpyDeviceToHost)); printf("overflowed points (all runs): %u\n", ovf);
    return 0;
}
