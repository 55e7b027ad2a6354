/*
 * gem_hip_debug.h -- tuning knobs and profiling aids of libgem_hip.so.  NOT part of the drop-in boundary
 * (include/gem_hip.h): nothing the reference's callers need is declared here.  The knobs choose between code
 * paths that all produce the same map; the parity tests use them to drive every path with small inputs.
 */
#ifndef GEM_HIP_DEBUG_H
#define GEM_HIP_DEBUG_H

#include "gem_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* keys: "fuse_variant" (10..12: k_fuse_list geometry on 32x32 tiles), "tile_shift" (0 = per pass, 4, 5),
 *       "defer" (0/1: one launch per frame for streams of single sweeps), "dense_min" (records of one sweep in one
 *       16x16 tile above which the tile is counting-sorted), "dbg_sweep", "overlap" (0/1: binning of big passes on a
 *       second stream), "overlap_min_points", "sort_path" (0/1: the sorted pipeline for big passes), "sort_min_points",
 *       "sort_passes" (0 = by map size and form, 1..3: counting-sort passes), "sort_form" (0 = by pass: batches of sweeps block-sorted
 *       (k_fuse_block), single clouds cell-sorted (k_fuse_walk); 1 / 2 force cell / block), "fast_laser" (0/1: the
 *       zero-rotation-variance form of the laser variance for frames that qualify), "rank_by_ballot" (0/1: k_sort_scatter matches equal bins by ballot in every pass instead
 *       of through the LDS), "lane_sort" (0/1: k_fuse_block hands the
 *       cells to the threads by record count), "blk_batch" (0 = by the pass's mean block load, 512 / 2048: records k_fuse_block
 *       stages per round), "few_bins" (0 = by pass; n > 0: ballots per wave instruction before k_sort_scatter's coherent ranking takes the
 *       LDS way; < 0: one ballot per digit bit), "ray_lanes" (1, 4, 8, 16: lanes of a wave that share one walk of gem_raytracing) and
 *       "ray_depth" (4, 8: steps a lane walks ahead of the loads it waits for), "walk_permute" (0/1),
 *       "sort_streams" (1, 2: binning streams consecutive overlapped passes of the sorted pipeline alternate between), "sort_ring" (2..4: the buffer sets they rotate through),
 *       "trace" (0/1: one line on stderr per pass of the sorted pipeline), "stream_roles" (a permutation of 0123 as a decimal
 *       number: which of the handle's current own / bin / bin2 / upload streams takes each role; tools/dbg/roles.py).  Returns GEM_ERR_INVALID for an unknown key or a value out of range. */
int gem_debug_set(gem_handle* h, const char* key, long long value);

/* read-outs: "arena_allocations" (device allocations the handle's arenas have made so far: none may follow gem_reserve) */
int gem_debug_get(gem_handle* h, const char* key, long long* out);

/* per-tile cycle stamps of the last fuse launch ([tile][16] 64-bit counters); enable != 0 turns the stamps on for the
 * following passes; with out != NULL copies up to max_tiles rows and returns their number */
int gem_debug_fuse_stamps(gem_handle* h, int enable, unsigned long long* out, int max_tiles);

#ifdef __cplusplus
}
#endif
#endif
