/*
 * cuda_runtime.h -- CPU stand-in for the CUDA runtime, TEST INFRASTRUCTURE ONLY (oracle/ref_build).
 *
 * Lets g++ compile the reference's own gpu_process.cu (from /root/reference, where it lies) into
 * oracle/_ref/libgem_ref.so: kernels become plain functions, a launch runs every (block, thread) of the grid
 * SEQUENTIALLY in index order on the calling thread, "device" memory is host memory.  The reference's kernels use no
 * shared memory and no barriers, so sequential execution is one of the schedules a GPU could produce (the one in which
 * thread i finishes before thread i + 1 starts); G_fuse loops over the points per CELL, in input order, so its result
 * does not depend on the schedule at all.
 * The launch syntax `k<<<g, b>>>(args)` is rewritten to GEMREF_LAUNCH(g, b, k(args)) by build_ref.py on the way to the
 * compiler (into a temporary file that is deleted again); nothing else of the reference's text is touched.
 */
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>

#define __global__
#define __device__
#define __host__
#define __constant__
#define __forceinline__ inline

struct gemref_uint3 { unsigned x, y, z; };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
extern gemref_uint3 threadIdx, blockIdx, blockDim, gridDim;

template <class F> static inline void gemref_launch(dim3 g, dim3 b, F&& body)
{
    gridDim = {g.x, g.y, g.z}; blockDim = {b.x, b.y, b.z};
    for (unsigned bz = 0; bz < g.z; ++bz) for (unsigned by = 0; by < g.y; ++by) for (unsigned bx = 0; bx < g.x; ++bx)
        for (unsigned tz = 0; tz < b.z; ++tz) for (unsigned ty = 0; ty < b.y; ++ty) for (unsigned tx = 0; tx < b.x; ++tx) {
            blockIdx = {bx, by, bz}; threadIdx = {tx, ty, tz};
            body();
        }
}
#define GEMREF_LAUNCH(g, b, call) gemref_launch(dim3(g), dim3(b), [&]() { call; })

typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3 };
static inline cudaError_t cudaMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? 0 : 2; }
static inline cudaError_t cudaFree(void* p) { free(p); return 0; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return 0; }
static inline cudaError_t cudaMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
template <class T> static inline cudaError_t cudaMemcpyToSymbol(T& sym, const void* s, size_t n, size_t off = 0, cudaMemcpyKind = cudaMemcpyHostToDevice)
{ memcpy(reinterpret_cast<char*>(&sym) + off, s, n); return 0; }
template <class T> static inline cudaError_t cudaMemcpyFromSymbol(void* d, const T& sym, size_t n, size_t off = 0, cudaMemcpyKind = cudaMemcpyDeviceToHost)
{ memcpy(d, reinterpret_cast<const char*>(&sym) + off, n); return 0; }
static inline cudaError_t cudaDeviceSynchronize() { return 0; }
static inline cudaError_t cudaGetLastError() { return 0; }
static inline const char* cudaGetErrorString(cudaError_t) { return "no error (CPU stand-in)"; }

/* device intrinsics the reference uses */
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int atomicCAS(int* a, int compare, int val) { const int old = *a; if (old == compare) *a = val; return old; }
static inline int atomicAdd(int* a, int v) { const int old = *a; *a = old + v; return old; }
static inline float atomicAdd(float* a, float v) { const float old = *a; *a = old + v; return old; }
static inline int atomicMin(int* a, int v) { const int old = *a; if (v < old) *a = v; return old; }
static inline int atomicMax(int* a, int v) { const int old = *a; if (v > old) *a = v; return old; }
