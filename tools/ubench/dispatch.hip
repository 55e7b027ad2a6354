// dispatch.hip -- how fast the chip starts and retires SHORT workgroups: C5's walk is 22 500 workgroups of 256 threads that live
// ~11 k cycles each (block_phases) and the kernel takes 113 us -- on average fewer than four of a CU's ten slots are occupied.
// Is that the dispatcher?  Grid of G workgroups x 256 threads with L bytes of LDS whose threads (a) do nothing, (b) wait for D
// DEPENDENT global loads (a pointer chase through a 256 MB table: every hop a miss) and cross B barriers.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/dispatch.hip -o tools/ubench/bin/dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void k_short(const unsigned* __restrict__ table, unsigned mask, int depth, int barriers, unsigned* sink)
{
    extern __shared__ unsigned lds[];
    unsigned x = (blockIdx.x * 2654435761u + (threadIdx.x >> 6) * 977u) & mask;      // one line per wave and hop: latency, not bandwidth
    for (int d = 0; d < depth; ++d) {
        x = table[x] & mask;
        if (d < barriers) { lds[threadIdx.x] = x; __syncthreads(); x ^= lds[(threadIdx.x + 64) & 255] & 0xff00u; x &= mask; }
    }
    if (x == 0xffffffffu) *sink = x;
}

int main()
{
    const unsigned words = 64u << 20;                        // 256 MB
    unsigned* table; unsigned* sink;
    hipMalloc(&table, (size_t)words * 4); hipMalloc(&sink, 4);
    std::vector<unsigned> h(words);
    unsigned s = 12345u;
    for (unsigned i = 0; i < words; ++i) { s = s * 1664525u + 1013904223u; h[i] = s >> 4; }
    hipMemcpy(table, h.data(), (size_t)words * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int lds : {7680, 16384}) for (int grid : {5632, 22500}) for (int depth : {0, 1, 2, 3, 4}) {
        std::vector<float> t;
        for (int rep = 0; rep < 7; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_short, dim3(grid), dim3(256), lds, 0, table, words - 1, depth, depth, sink);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3f);
        }
        std::sort(t.begin(), t.end());
        printf("lds %5d  grid %5d  dependent loads + barriers %d: %7.1f us  (%.0f workgroups / us)\n", lds, grid, depth, t[3], grid / t[3]);
    }
    return 0;
}
