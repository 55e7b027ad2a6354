// chain.hip -- what ONE step of the per-cell recurrence costs a wave that is alone on its SIMD, and where the cycles go.
// k_fuse_block's heaviest block spends ~850 shader cycles per chain step on ~65 instructions (tools/block_phases.py) whether one or
// three workgroups share its CU: this strips the loop down, variant by variant, on synthetic records in LDS.
//   V0  the plain loop as the kernel has it: record {h, v} + sweep from LDS a step ahead, the next sweep's increment from LDS
//   V1  ... the increment a constant (no dependent LDS read)
//   V2  ... records from registers too (no LDS in the loop)
//   V3  V0 with scalar quotients (no v_pk_*)
//   V4  V0 without the two transcendentals (timing only)
//   V5  only the LDS reads and the loop
//   V6  V0 with the rare-path branch removed (timing only)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-fno-slp-vectorize] tools/ubench/chain.hip -o tools/ubench/bin/chain
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kSteps = 4096;
typedef float v2f __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256) void k_chain(const float* __restrict__ init, float* __restrict__ out, unsigned long long* __restrict__ ticks, int steps, float fl_, float thr)
{
    __shared__ uint2 st_hv[2048 + 64];
    __shared__ uint16_t st_sw[2048 + 64];
    __shared__ float vu[520];
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < 2048 + 64; i += 256) {
        const float h = 0.5f + 0.001f * (float)((i * 37) % 101), v = 4e-4f + 1e-6f * (float)((i * 13) % 17);
        st_hv[i] = make_uint2(__float_as_uint(h), __float_as_uint(v));
        st_sw[i] = (uint16_t)((i / 3) % 32);
    }
    for (int i = tid; i < 520; i += 256) vu[i] = 1e-6f * (float)(1 + i % 3);
    __syncthreads();
    float ce = init[tid], cs = init[256 + tid];
    uint32_t cur = 0;
    const float band = 1e-5f * fabsf(thr);
    const uint32_t cf = (uint32_t)((tid * 29) % 1500);
    const uint32_t cn = (uint32_t)steps;
    const uint2* ph = st_hv + cf + 1u;
    const uint16_t* ps = st_sw + cf + 1u;
    uint2 nx = st_hv[cf]; uint32_t nx_sw = st_sw[cf];
    float un = vu[cur + 1u], un2 = vu[cur + 2u], un3 = vu[cur + 3u];
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz
    for (uint32_t i = 0; i < (uint32_t)steps; ++i) {
        uint2 r = nx; uint32_t swr = nx_sw;
        if (V != 2) { nx = *ph; nx_sw = *ps; ph += 1; ps += 1; if (((i + 1u) & 511u) == 0u) { ph -= 512; ps -= 512; } }
        else { r.x += 1u; }
        const bool live = i < cn;
        const float h = __uint_as_float(r.x), v = __uint_as_float(r.y);
        if (V == 5) { ce += h; cs += v + (float)swr; continue; }
        bool rare = false;
        uint32_t sw = live ? (swr & 31u) : cur;
        if (sw < cur) sw = cur;                                          // (synthetic sweeps wrap: keep them monotone)
        if (V == 7) {                                                    // the three-level advance of the kernel
            const uint32_t gap = sw - cur;
            const float c1 = (cs < fl_ ? fl_ : cs) + un;
            cs = gap >= 1u ? c1 : cs;
            const float c2 = cs + un2;
            cs = gap >= 2u ? c2 : cs;
            const float c3 = cs + un3;
            cs = gap >= 3u ? c3 : cs;
            cur += min(gap, 3u);
            cur &= 31u;
            rare = gap > 40u;
            un = vu[cur + 1u]; un2 = vu[cur + 2u]; un3 = vu[cur + 3u];
        }
        const uint32_t gap = V == 7 ? 0u : sw - cur;
        const bool adv = gap != 0u;
        const float ca = (cs < fl_ ? fl_ : cs) + un;
        cs = adv ? ca : cs;
        cur += adv ? 1u : 0u;
        cur &= 31u;
        rare = gap > 40u;
        if (V == 0 || V == 3 || V == 4 || V == 6) un = vu[cur + 1u];
        const float sf = cs < fl_ ? fl_ : cs;
        const float rs = V == 4 ? sf * 3.0f : __builtin_amdgcn_rsqf(sf);
        const float m = fabsf(h - ce) * rs;
        const float D = sf + v;
        const float N1 = sf * h + v * ce, N2 = v * sf;
        rare = (rare | (fabsf(m - thr) <= band) | !(fabsf(N1) >= 8.673617379884035e-19f)) & live;
        const float r0 = V == 4 ? D * 0.5f : __builtin_amdgcn_rcpf(D);
        const float rr = __builtin_fmaf(__builtin_fmaf(-D, r0, 1.0f), r0, r0);
        float en, sn;
        if (V == 3) {
            float q = N1 * rr;  float t = __builtin_fmaf(-D, q, N1);  q = __builtin_fmaf(t, rr, q);  t = __builtin_fmaf(-D, q, N1);  en = __builtin_fmaf(t, rr, q);
            q = N2 * rr;        t = __builtin_fmaf(-D, q, N2);        q = __builtin_fmaf(t, rr, q);  t = __builtin_fmaf(-D, q, N2);  sn = __builtin_fmaf(t, rr, q);
        } else {
            const v2f N = {N1, N2}, rr2 = {rr, rr}, nD2 = {-D, -D};
            v2f q = N * rr2;
            v2f t = __builtin_elementwise_fma(nD2, q, N);
            q = __builtin_elementwise_fma(t, rr2, q);
            t = __builtin_elementwise_fma(nD2, q, N);
            q = __builtin_elementwise_fma(t, rr2, q);
            en = q.x; sn = q.y;
        }
        const bool outlier = m > thr;
        const bool replace = (ce == -10.0f) | (outlier & (ce < h));
        float e2 = replace ? h : (outlier ? ce : en);
        float s2 = replace ? v : (outlier ? sf : sn);
        if (V != 6) {
            if (__builtin_expect(__ballot(rare) != 0, 0)) { e2 = ce + 1.0f / (cs + h); s2 = sqrtf(cs + v); }
        }
        ce = live ? e2 : ce; cs = live ? s2 : cs;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    out[tid] = ce + cs + (float)cur + un2 + un3;
    if (tid == 0) { ticks[2 * blockIdx.x] = t1 - t0; ticks[2 * blockIdx.x + 1] = rt1 - rt0; }
}

template <int V>
static void run(const char* name, int threads, int blocks)
{
    float* d_init; float* d_out; unsigned long long* d_t;
    hipMalloc(&d_init, 512 * 4); hipMalloc(&d_out, 256 * 4); hipMalloc(&d_t, 16 * blocks);
    std::vector<float> init(512);
    for (int i = 0; i < 256; ++i) { init[i] = 0.52f + 0.0001f * i; init[256 + i] = 3e-4f; }
    hipMemcpy(d_init, init.data(), 512 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_chain<V>, dim3(blocks), dim3(threads), 0, 0, d_init, d_out, d_t, kSteps, 1e-4f, 5.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> t(2 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_chain<V>, dim3(blocks), dim3(threads), 0, 0, d_init, d_out, d_t, kSteps, 1e-4f, 5.0f);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(t.data(), d_t, 16 * blocks, hipMemcpyDeviceToHost);
    double tmin = 1e30, tmax = 0, tsum = 0, rmin = 1e30, rmax = 0;
    for (int b = 0; b < blocks; ++b) {
        const double tk = (double)t[2 * b] / kSteps, rt = (double)t[2 * b + 1] * 10.0 / kSteps;      // ns / step from the 100 MHz counter
        tmin = tk < tmin ? tk : tmin; tmax = tk > tmax ? tk : tmax; tsum += tk; rmin = rt < rmin ? rt : rmin; rmax = rt > rmax ? rt : rmax;
    }
    printf("%-58s threads %3d blocks %4d: s_memtime ticks / step min %6.1f mean %6.1f max %6.1f | realtime ns / step min %6.1f max %6.1f | kernel %.1f us = %.1f ns / step\n",
           name, threads, blocks, tmin, tsum / blocks, tmax, rmin, rmax, ms * 1e3, ms * 1e6 / kSteps);
    hipFree(d_init); hipFree(d_out); hipFree(d_t);
}

int main()
{
    for (int threads : {64, 256}) {
        run<0>("V0 plain loop (records + sweep + increment from LDS)", threads, 1);
        run<1>("V1 increment constant", threads, 1);
        run<2>("V2 no LDS in the loop", threads, 1);
        run<3>("V3 V0 with scalar quotients (no v_pk)", threads, 1);
        run<4>("V4 V0 without rsq / rcp", threads, 1);
        run<5>("V5 LDS reads + loop only", threads, 1);
        run<6>("V6 V0 without the rare branch", threads, 1);
        run<7>("V7 V0 with the three-level advance (5 LDS reads)", threads, 1);
    }
    run<0>("V0, one block per CU", 256, 256);
    run<0>("V0, every CU busy (3 blocks per CU)", 256, 768);
    run<0>("V0, 6 blocks per CU", 256, 1536);
    run<3>("V3, every CU busy", 256, 768);
    return 0;
}
