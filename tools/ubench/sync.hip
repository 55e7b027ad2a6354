// sync.hip -- what the host pays to learn that a stream is idle: the time from the launch of a kernel that spins for 20 us to the
// return of the wait, minus those 20 us (launch latency + completion latency), for the ways HIP offers to wait.  The driver's bench
// command synchronises every 20 sweeps (~160 us of kernels): every microsecond here is 0.05 us per step there.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/sync.hip -o tools/ubench/bin/sync
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_spin(unsigned long long ticks, unsigned* sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned x = threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu) *sink = x;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    unsigned* sink; hipMalloc(&sink, 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t ev, evb; hipEventCreateWithFlags(&ev, hipEventDisableTiming); hipEventCreateWithFlags(&evb, hipEventDisableTiming | hipEventBlockingSync);
    const char* names[] = {"hipStreamSynchronize", "hipEventRecord + hipEventSynchronize", "hipEventRecord + spin on hipEventQuery", "spin on hipStreamQuery",
                           "hipEventRecord(BlockingSync) + hipEventSynchronize", "hipDeviceSynchronize"};
    for (int spin_us : {20, 160}) {
        for (int v = 0; v < 6; ++v) {
            std::vector<double> over, launch;
            for (int rep = 0; rep < 60; ++rep) {
                hipDeviceSynchronize();
                const double t0 = now_us();
                hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s, 100ull * spin_us, sink);
                const double t1 = now_us();
                switch (v) {
                case 0: hipStreamSynchronize(s); break;
                case 1: hipEventRecord(ev, s); hipEventSynchronize(ev); break;
                case 2: hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) {} break;
                case 3: while (hipStreamQuery(s) == hipErrorNotReady) {} break;
                case 4: hipEventRecord(evb, s); hipEventSynchronize(evb); break;
                default: hipDeviceSynchronize(); break;
                }
                const double t2 = now_us();
                if (rep >= 10) { over.push_back(t2 - t0 - spin_us); launch.push_back(t1 - t0); }
            }
            std::sort(over.begin(), over.end()); std::sort(launch.begin(), launch.end());
            std::printf("kernel of %3d us, %-52s launch call %5.2f us, launch-to-return minus the kernel %6.2f us (median; min %6.2f, max %6.2f)\n",
                        spin_us, names[v], launch[launch.size() / 2], over[over.size() / 2], over.front(), over.back());
        }
    }
    return 0;
}
