// gem_wave.hpp -- wave64 / workgroup primitives shared by the gfx950 kernels (internal header).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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

// Wave-wide inclusive scans on the DPP network (row_shr within rows of 16 lanes, then row_bcast
// across rows): six VALU instructions instead of six LDS-crossbar permutes.
#define GEM_DPP(old, src, ctrl, rowmask) \
    (uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(src), (ctrl), (rowmask), 0xf, false)

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v)
{
    v += GEM_DPP(0, v, 0x111, 0xf);      // row_shr:1
    v += GEM_DPP(0, v, 0x112, 0xf);      // row_shr:2
    v += GEM_DPP(0, v, 0x114, 0xf);      // row_shr:4
    v += GEM_DPP(0, v, 0x118, 0xf);      // row_shr:8
    v += GEM_DPP(0, v, 0x142, 0xa);      // row_bcast:15 -> rows 1 and 3
    v += GEM_DPP(0, v, 0x143, 0xc);      // row_bcast:31 -> rows 2 and 3
    return v;
}

__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v)    // identity 0
{
    v = max(v, GEM_DPP(0, v, 0x111, 0xf));
    v = max(v, GEM_DPP(0, v, 0x112, 0xf));
    v = max(v, GEM_DPP(0, v, 0x114, 0xf));
    v = max(v, GEM_DPP(0, v, 0x118, 0xf));
    v = max(v, GEM_DPP(0, v, 0x142, 0xa));
    v = max(v, GEM_DPP(0, v, 0x143, 0xc));
    return v;
}

// value of the previous lane (0 for lane 0): wave_shr:1
__device__ __forceinline__ uint32_t wave_prev(uint32_t v) { return GEM_DPP(0, v, 0x138, 0xf); }

// exclusive scan over an NT-thread block (NT/64 <= 16 waves); scratch = 16 uint32 in LDS.
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* scratch, uint32_t* total)
{
    constexpr int NW = NT / 64;
    const int w = (int)(threadIdx.x >> 6);
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane_id() == 63) scratch[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint32_t s = scratch[i];
        if (i < w) base += s;
        tot += s;
    }
    __syncthreads();     // scratch may be reused by the caller right away
    *total = tot;
    return base + inc - v;
}

// The same with ONE barrier: consecutive calls alternate between the two halves of `scratch` (parity), so a wave that runs
// ahead into the next call writes the other half while a slower wave may still be reading this one; by the time a half is
// written again every wave has passed the barrier of the call in between.  NT/64 <= 8.
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan_alt(uint32_t v, uint32_t* scratch, uint32_t parity, uint32_t* total)
{
    constexpr int NW = NT / 64;
    static_assert(NW <= 8, "two halves of 8 words");
    uint32_t* sc = scratch + (parity & 1u) * 8u;
    const int w = (int)(threadIdx.x >> 6);
    const uint32_t inc = wave_inclusive_scan(v);
    if (lane_id() == 63) sc[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const uint32_t s = sc[i];
        if (i < w) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

// "match-any" on a 64-lane wave: the mask of valid lanes holding the same key as this lane,
// from one ballot per key bit (no loop over distinct keys, no LDS, no divergence).
// rank among equal keys in lane order = popc(peers & lanemask_lt); group size = popc(peers).
__device__ __forceinline__ uint64_t wave_peers(bool valid, uint32_t key, int nbits)
{
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < nbits; ++b) {                     // wave-uniform trip count
        const bool bit = (key >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return valid ? peers : 0ull;
}

// The same when a wave holds only a few distinct keys (the 64 consecutive points of a LiDAR ring or of an
// image row fall into 2-8 tiles): one ballot per distinct key, taken from the first lane still unmatched.
// More than 8 distinct keys (a random cloud) fall back to the per-bit form.
__device__ __forceinline__ uint64_t wave_peers_few(bool valid, uint32_t key, int nbits)
{
    uint64_t remaining = __ballot(valid), peers = 0;
    for (int it = 0; it < 8 && remaining != 0; ++it) {   // wave-uniform
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, __ffsll((unsigned long long)remaining) - 1);
        const bool mine = valid && key == k;
        const uint64_t m = __ballot(mine);
        peers = mine ? m : peers;
        remaining &= ~m;
    }
    return remaining != 0 ? wave_peers(valid, key, nbits) : peers;
}

// the map_lowest side output (GPU:432-439) in input order: lowest = min(lowest, h); if (h == lowest) lowest += 3 * var
__device__ __forceinline__ float lowest_step(float lw, float h, float v)
{
    const float l2 = fminf(h, lw);                                                  // GPU:434 atomicMin (GPU:372-382)
    return h == l2 ? l2 + 3.0f * v : l2;                                            // GPU:435-438
}

} // namespace gem
