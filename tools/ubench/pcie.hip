// pcie.hip -- how to move a frame's arrays between device memory and the library's pinned staging buffer: DMA commands
// (hipMemcpyAsync) against copy kernels that store to / load from host memory, whole and in pieces, and what it costs the host to
// learn that a piece has arrived (hipEventSynchronize against polling a word the kernel writes behind its data).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/pcie.hip -o tools/ubench/bin/pcie
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_copy(u32x4* __restrict__ d, const u32x4* __restrict__ s, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(s[i], d + i);
}

// ... and a word behind the data: the last workgroup to finish publishes `seq`
__global__ __launch_bounds__(256) void k_copy_flag(u32x4* __restrict__ d, const u32x4* __restrict__ s, size_t n16, unsigned* counter, volatile unsigned* flag, unsigned seq)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(s[i], d + i);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            *counter = 0;
            __threadfence_system();
            __hip_atomic_store(const_cast<unsigned*>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void k_tiny(unsigned* p) { if (threadIdx.x == 0) p[0] += 1; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t total = 9ull * 600 * 600 * 4;            // Map_feature's nine layers of a 600 x 600 map
    unsigned char *dev, *pin; unsigned *cnt; volatile unsigned* flags;
    hipMalloc(&dev, total + 4096); hipMalloc(&cnt, 256);
    hipHostMalloc(&pin, total + 4096, hipHostMallocDefault);
    hipHostMalloc((void**)&flags, 4096, hipHostMallocDefault);
    hipMemset(dev, 7, total); hipMemset(cnt, 0, 256); std::memset(pin, 0, total); std::memset((void*)flags, 0, 4096);
    std::vector<unsigned char> user(total);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t ev[32]; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const int reps = 21;

    // 1. host cost of learning that a tiny kernel has finished
    { std::vector<double> a, b;
      for (int r = 0; r < reps; ++r) {
          double s = now_us(); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, cnt + 8); hipEventRecord(ev[0], st); hipEventSynchronize(ev[0]); a.push_back(now_us() - s);
          s = now_us(); hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, cnt + 8); hipStreamSynchronize(st); b.push_back(now_us() - s);
      }
      std::printf("tiny kernel + hipEventRecord + hipEventSynchronize %.1f us; + hipStreamSynchronize %.1f us\n", med(a), med(b)); }

    for (int pieces : {1, 2, 4, 8, 9, 16}) {
        const size_t pb = ((total / pieces) + 255) & ~(size_t)255;
        std::vector<double> dma, dma_cp, kern, kern_cp, flag_cp, h2d, k_h2d;
        for (int r = 0; r < reps; ++r) {
            // a. DMA commands, one wait at the end
            double s = now_us();
            for (int p = 0; p < pieces; ++p) { const size_t o = p * pb, b = o + pb <= total ? pb : total - o; hipMemcpyAsync(pin + o, dev + o, b, hipMemcpyDeviceToHost, st); }
            hipStreamSynchronize(st); dma.push_back(now_us() - s);
            // b. DMA commands + event per piece + memcpy to the user's array as pieces arrive
            s = now_us();
            for (int p = 0; p < pieces; ++p) { const size_t o = p * pb, b = o + pb <= total ? pb : total - o; hipMemcpyAsync(pin + o, dev + o, b, hipMemcpyDeviceToHost, st); hipEventRecord(ev[p], st); }
            for (int p = 0; p < pieces; ++p) { const size_t o = p * pb, b = o + pb <= total ? pb : total - o; hipEventSynchronize(ev[p]); std::memcpy(user.data() + o, pin + o, b); }
            dma_cp.push_back(now_us() - s);
            // c. copy kernels, one wait
            s = now_us();
            for (int p = 0; p < pieces; ++p) { const size_t o = p * pb, b = o + pb <= total ? pb : total - o; hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, st, (u32x4*)(pin + o), (const u32x4*)(dev + o), b / 16); }
            hipStreamSynchronize(st); kern.push_back(now_us() - s);
            // d. copy kernels + events + memcpy
            s = now_us();
            for (int p = 0; p < pieces; ++p) { const size_t o = p * pb, b = o + pb <= total ? pb : total - o; hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, st, (u32x4*)(pin + o), (const u32x4*)(dev + o), b / 16); hipEventRecord(ev[p], st); }
            for (int p = 0; p < pieces; ++p) { const size_t o = p * pb, b = o + pb <= total ? pb : total - o; hipEventSynchronize(ev[p]); std::memcpy(user.data() + o, pin + o, b); }
            kern_cp.push_back(now_us() - s);
            // e. copy kernels that publish a word + polling + memcpy
            const unsigned seq = (unsigned)(r * 100 + pieces * 7 + 1);
            s = now_us();
            for (int p = 0; p < pieces; ++p) { const size_t o = p * pb, b = o + pb <= total ? pb : total - o;
                hipLaunchKernelGGL(k_copy_flag, dim3(1024), dim3(256), 0, st, (u32x4*)(pin + o), (const u32x4*)(dev + o), b / 16, cnt, flags + 16 * p, seq); }
            for (int p = 0; p < pieces; ++p) { const size_t o = p * pb, b = o + pb <= total ? pb : total - o;
                while (__atomic_load_n(const_cast<unsigned*>(flags + 16 * p), __ATOMIC_ACQUIRE) != seq) __builtin_ia32_pause();
                std::memcpy(user.data() + o, pin + o, b); }
            flag_cp.push_back(now_us() - s);
            hipStreamSynchronize(st);
            // f. uploads: DMA, and a kernel that reads host memory
            s = now_us();
            for (int p = 0; p < pieces; ++p) { const size_t o = p * pb, b = o + pb <= total ? pb : total - o; hipMemcpyAsync(dev + o, pin + o, b, hipMemcpyHostToDevice, st); }
            hipStreamSynchronize(st); h2d.push_back(now_us() - s);
            s = now_us();
            for (int p = 0; p < pieces; ++p) { const size_t o = p * pb, b = o + pb <= total ? pb : total - o; hipLaunchKernelGGL(k_copy, dim3(1024), dim3(256), 0, st, (u32x4*)(dev + o), (const u32x4*)(pin + o), b / 16); }
            hipStreamSynchronize(st); k_h2d.push_back(now_us() - s);
        }
        bool ok = true; for (size_t i = 0; i < total; i += 4097) ok &= user[i] == 7;
        std::printf("%2d pieces of %7zu B: D2H DMA %6.1f | DMA+events+memcpy %6.1f | kernels %6.1f | kernels+events+memcpy %6.1f | kernels+flags+memcpy %6.1f || H2D DMA %6.1f | H2D kernels %6.1f  us  %s\n",
                    pieces, pb, med(dma), med(dma_cp), med(kern), med(kern_cp), med(flag_cp), med(h2d), med(k_h2d), ok ? "" : "DATA WRONG");
    }
    return 0;
}
