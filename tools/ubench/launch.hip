// launch.hip -- what a kernel boundary costs on this machine, and what it depends on ("an empty kernel costs
// 2.7-3.0 us here against the guide's 1.45 us").  Back-to-back launches of an EMPTY kernel on one stream, N launches between two
// host synchronisations; reports the wall time per launch (host clock over the whole loop: what a stream of single sweeps pays)
// and the GPU-side time between the first dispatch's start and the last one's end (events).
//   variants: kernel argument bytes (0 / 64 / 1024 / 3072: k_frame passes FuseArgs + BinArgs by value, ~1.2 KB),
//             null stream | blocking stream | non-blocking stream,
//             hipLaunchKernelGGL | hipExtLaunchKernelGGL (with and without time-stamp events) | + hipGetLastError,
//             grid of 1 workgroup | 2000 workgroups of 256 threads (k_frame's size).
// Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/launch.hip -o tools/ubench/bin/launch
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <vector>

template <int BYTES> struct Blob { unsigned char b[BYTES > 0 ? BYTES : 1]; };

template <int BYTES> __global__ void k_empty(Blob<BYTES> blob, int* sink) { if (sink && blob.b[0] == 123 && threadIdx.x == 9999) *sink = 1; }
__global__ void k_empty0(int* sink) { if (sink && threadIdx.x == 9999) *sink = 1; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <typename F>
static void run(const char* name, hipStream_t st, int n, F launch)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) launch(i);
    hipStreamSynchronize(st);
    double best_wall = 1e30, best_gpu = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        const double t0 = now_us();
        hipEventRecord(e0, st);
        for (int i = 0; i < n; ++i) launch(i);
        hipEventRecord(e1, st);
        hipStreamSynchronize(st);
        const double wall = (now_us() - t0) / n;
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (wall < best_wall) best_wall = wall;
        if (ms * 1e3 / n < best_gpu) best_gpu = ms * 1e3 / n;
    }
    printf("%-78s wall %6.2f us/launch   gpu (event to event) %6.2f us/launch\n", name, best_wall, best_gpu);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
    const int N = 4000;
    hipStream_t s_block, s_nonblock;
    hipStreamCreate(&s_block);
    hipStreamCreateWithFlags(&s_nonblock, hipStreamNonBlocking);
    int* sink = nullptr; hipMalloc(&sink, 4);
    Blob<64> b64{}; Blob<1024> b1k{}; Blob<3072> b3k{};
    printf("# empty kernel, %d back-to-back launches per measurement, best of 5\n", N);
    run("null stream, no args, 1 workgroup", nullptr, N, [&](int) { hipLaunchKernelGGL(k_empty0, dim3(1), dim3(64), 0, nullptr, sink); });
    run("blocking stream, no args, 1 workgroup", s_block, N, [&](int) { hipLaunchKernelGGL(k_empty0, dim3(1), dim3(64), 0, s_block, sink); });
    run("non-blocking stream, no args, 1 workgroup", s_nonblock, N, [&](int) { hipLaunchKernelGGL(k_empty0, dim3(1), dim3(64), 0, s_nonblock, sink); });
    run("non-blocking stream, 64-byte args, 1 workgroup", s_nonblock, N, [&](int) { hipLaunchKernelGGL(k_empty<64>, dim3(1), dim3(64), 0, s_nonblock, b64, sink); });
    run("non-blocking stream, 1 KB args, 1 workgroup", s_nonblock, N, [&](int) { hipLaunchKernelGGL(k_empty<1024>, dim3(1), dim3(64), 0, s_nonblock, b1k, sink); });
    run("non-blocking stream, 3 KB args, 1 workgroup", s_nonblock, N, [&](int) { hipLaunchKernelGGL(k_empty<3072>, dim3(1), dim3(64), 0, s_nonblock, b3k, sink); });
    run("non-blocking stream, 1 KB args, 2000 workgroups x 256", s_nonblock, N, [&](int) { hipLaunchKernelGGL(k_empty<1024>, dim3(2000), dim3(256), 0, s_nonblock, b1k, sink); });
    run("non-blocking stream, 1 KB args, 2000 x 256, 31 KB dynamic LDS", s_nonblock, N, [&](int) { hipLaunchKernelGGL(k_empty<1024>, dim3(2000), dim3(256), 31 * 1024, s_nonblock, b1k, sink); });
    run("  + hipGetLastError after every launch", s_nonblock, N, [&](int) { hipLaunchKernelGGL(k_empty<1024>, dim3(2000), dim3(256), 0, s_nonblock, b1k, sink); (void)hipGetLastError(); });
    run("  hipExtLaunchKernelGGL, no events", s_nonblock, N, [&](int) { hipExtLaunchKernelGGL(k_empty<1024>, dim3(2000), dim3(256), 0, s_nonblock, nullptr, nullptr, 0, b1k, sink); });
    {
        std::vector<hipEvent_t> ea(N + 200), eb(N + 200);
        for (auto& e : ea) hipEventCreate(&e);
        for (auto& e : eb) hipEventCreate(&e);
        run("  hipExtLaunchKernelGGL with start / stop events (gem_set_timing)", s_nonblock, N,
            [&](int i) { hipExtLaunchKernelGGL(k_empty<1024>, dim3(2000), dim3(256), 0, s_nonblock, ea[i], eb[i], 0, b1k, sink); });
        hipStreamSynchronize(s_nonblock);
        float ms = 0, tot = 0; int cnt = 0;
        for (int i = 100; i < N; ++i) if (hipEventElapsedTime(&ms, ea[i], eb[i]) == hipSuccess) { tot += ms; ++cnt; }
        printf("    dispatch time stamps of the empty 2000 x 256 kernel itself: %.2f us average\n", cnt ? tot * 1e3 / cnt : 0.0);
        float gap = 0; int gc = 0;
        for (int i = 101; i < N; ++i) if (hipEventElapsedTime(&ms, eb[i - 1], ea[i]) == hipSuccess) { gap += ms; ++gc; }
        printf("    end of one dispatch to start of the next: %.2f us average\n", gc ? gap * 1e3 / gc : 0.0);
    }
    return 0;
}
