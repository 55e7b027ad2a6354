// What does global_load_lds_dwordx3 write where?  Lane l fetches words {3l, 3l+1, 3l+2}; the LDS image is dumped.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint32_t* g, uint32_t* out, int misalign)
{
    __shared__ uint32_t s[512];
    for (int i = threadIdx.x; i < 512; i += 64) s[i] = 0xdeadbeefu;
    __syncthreads();
    if (threadIdx.x < 40)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + threadIdx.x * 3),
                                         (__attribute__((address_space(3))) void*)(s + 4 + misalign), 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = s[i];
}
int main()
{
    std::vector<uint32_t> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
    uint32_t *g, *o;
    hipMalloc(&g, 4096); hipMalloc(&o, 2048);
    hipMemcpy(g, h.data(), 4096, hipMemcpyHostToDevice);
    for (int mis = 0; mis < 4; ++mis) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o, mis);
        std::vector<uint32_t> r(512);
        hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
        printf("base word %d:", 4 + mis);
        for (int i = 0; i < 200; ++i) { if (r[i] == 0xdeadbeefu) printf(" ."); else printf(" %u", r[i] - 1000); }
        printf("\n");
    }
    return 0;
}
