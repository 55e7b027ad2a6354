// handover.hip -- what it costs a kernel to start behind a kernel of ANOTHER stream (event record + hipStreamWaitEvent), against
// following it on the same stream.  The walks of the sorted pipelines wait for their sort that way and start 11-13 us after it ends
// (profiles/r04_c*_timeline_overlapped.txt); consecutive kernels of one stream start 0.1 us apart.
// Kernel A (stream 1, ~20 us of spinning) stamps its end, kernel B (stream 2) stamps its start: B_start - A_end in device time
// (s_memrealtime, 100 MHz), median of 50.  Variants: the event's flags; whether stream 2 is idle or busy with a kernel C that ends
// just before A does (the walk before); B launched with hipExtLaunchKernelGGL; the wait enqueued long before / right before A ends.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/handover.hip -o tools/ubench/bin/handover
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void k_spin(unsigned long long ticks, unsigned long long* stamp_end, unsigned* sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned x = threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeefu) *sink = x;
    if (threadIdx.x == 0 && blockIdx.x == 0 && stamp_end) *stamp_end = __builtin_amdgcn_s_memrealtime();
}
__global__ void k_stamp(unsigned long long* stamp_start, unsigned* data, unsigned* sink)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *stamp_start = __builtin_amdgcn_s_memrealtime();
    if (data && data[blockIdx.x * blockDim.x + threadIdx.x] == 0xdeadbeefu) *sink = 1;
}
__global__ void k_touch(unsigned* data, unsigned n) { for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) data[i] += 1u; }

int main()
{
    unsigned long long* st; unsigned* sink; unsigned* data;
    hipHostMalloc((void**)&st, 4096, hipHostMallocDefault);
    hipMalloc(&sink, 4); const unsigned nd = 8u << 20; hipMalloc(&data, nd * 4); hipMemset(data, 0, nd * 4);
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    struct V { const char* name; unsigned flags; int busy2; int ext; int dirty; int same; };
    const V vs[] = {
        {"same stream (no event)", 0, 0, 0, 0, 1},
        {"same stream, A dirties 32 MB", 0, 0, 0, 1, 1},
        {"event default flags, stream 2 idle", hipEventDefault, 0, 0, 0, 0},
        {"event DisableTiming, stream 2 idle", hipEventDisableTiming, 0, 0, 0, 0},
        {"event DisableTiming|DisableSystemFence, stream 2 idle", hipEventDisableTiming | hipEventDisableSystemFence, 0, 0, 0, 0},
        {"event DisableTiming, stream 2 busy until ~2 us before", hipEventDisableTiming, 1, 0, 0, 0},
        {"event DisableTiming|DisableSystemFence, stream 2 busy", hipEventDisableTiming | hipEventDisableSystemFence, 1, 0, 0, 0},
        {"event DisableTiming, A dirties 32 MB, stream 2 busy", hipEventDisableTiming, 1, 0, 1, 0},
        {"event DisableTiming|DisableSystemFence, A dirties 32 MB, stream 2 busy", hipEventDisableTiming | hipEventDisableSystemFence, 1, 0, 1, 0},
        {"stop event of A's dispatch (hipExtLaunch), stream 2 busy", hipEventDisableTiming, 1, 1, 0, 0},
    };
    // ... and the steady state of the overlapped pipelines: the event stream 2 waits for completed LONG before stream 2's own kernel C
    // (the walk before) ends -- how long after C does B start then?
    for (int late_us : {5, 20, 60}) {
        for (unsigned flags : {(unsigned)hipEventDisableTiming, (unsigned)(hipEventDisableTiming | hipEventDisableSystemFence)}) {
            hipEvent_t ev; hipEventCreateWithFlags(&ev, flags);
            std::vector<double> gap;
            for (int rep = 0; rep < 50; ++rep) {
                st[0] = st[1] = st[2] = 0;
                const unsigned long long a_ticks = 2000;
                hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s2, a_ticks + 100ull * late_us, st + 2, sink);      // C: ends late_us after A
                hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s1, a_ticks, st, sink); hipEventRecord(ev, s1);
                hipStreamWaitEvent(s2, ev, 0);
                hipLaunchKernelGGL(k_stamp, dim3(1024), dim3(256), 0, s2, st + 1, data, sink);
                hipStreamSynchronize(s1); hipStreamSynchronize(s2);
                if (rep >= 5) gap.push_back(((double)st[1] - (double)st[2]) * 0.01);
            }
            std::sort(gap.begin(), gap.end());
            printf("event %s, completed ~%2d us before stream 2's own kernel C ends          B starts %6.2f us after C ends (median; min %6.2f, max %6.2f)\n",
                   flags & hipEventDisableSystemFence ? "DisableTiming|DisableSystemFence" : "DisableTiming                   ", late_us, gap[gap.size() / 2], gap.front(), gap.back());
            hipEventDestroy(ev);
        }
    }
    for (const V& v : vs) {
        hipEvent_t ev; hipEventCreateWithFlags(&ev, v.flags ? v.flags : hipEventDefault);
        std::vector<double> gap;
        for (int rep = 0; rep < 50; ++rep) {
            st[0] = st[1] = 0;
            const unsigned long long a_ticks = 2000;                       // 20 us
            if (v.same) {
                if (v.dirty) hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, s1, data, nd);
                hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s1, a_ticks, st, sink);
                hipLaunchKernelGGL(k_stamp, dim3(1024), dim3(256), 0, s1, st + 1, data, sink);
                hipStreamSynchronize(s1);
            } else {
                if (v.busy2) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s2, a_ticks - 200, (unsigned long long*)nullptr, sink);   // "the walk before": ends ~2 us before A
                if (v.dirty) hipLaunchKernelGGL(k_touch, dim3(2048), dim3(256), 0, s1, data, nd);
                if (v.ext) hipExtLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s1, nullptr, ev, 0, a_ticks, st, sink);
                else { hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s1, a_ticks, st, sink); hipEventRecord(ev, s1); }
                hipStreamWaitEvent(s2, ev, 0);
                hipLaunchKernelGGL(k_stamp, dim3(1024), dim3(256), 0, s2, st + 1, data, sink);
                hipStreamSynchronize(s1); hipStreamSynchronize(s2);
            }
            if (rep >= 5) gap.push_back(((double)st[1] - (double)st[0]) * 0.01);      // 100 MHz ticks -> us
        }
        std::sort(gap.begin(), gap.end());
        printf("%-72s  B starts %6.2f us after A ends (median; min %6.2f, max %6.2f)\n", v.name, gap[gap.size() / 2], gap.front(), gap.back());
        hipEventDestroy(ev);
    }
    return 0;
}
