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
 *       16x16 tile above which the tile is counting-sorted), "dbg_sweep", "dbg_frame" (0/1: with gem_debug_fuse_stamps on, a stream of
 *       single sweeps still runs as k_frame and its binning blocks are stamped too: rows [T, T + blocks)), "overlap" (0/1: binning of big passes on a
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
 *       "plain_loop" (0/1: the walks' plain chain loop for blocks / passes whose values are in range; 0 = the guarded loop everywhere),
 *       "light_fast" (0/1: k_fuse_block's rounds of 512 by arrival slots + a per-cell sorting network; 0 = the general rounds),
 *       "cache_tables" (0/1: a batched call whose frames / offsets / increments / map pose equal what a buffer set's device tables were
 *       built from skips building and uploading them), "walk_prio" (0 = off; blocks of at least that many records raise their waves'
 *       issue priority), "sort_chunk" (0 = by the pass's size, 1024 / 4096: records per counting-sort chunk of the sorted pipelines), "defer_walk" (0/1: an overlapped sorted pass of caller-owned device input leaves its walk to the next call, which launches it without a stream wait when its sort has completed), "walk_always_wait" (0/1: that launch waits for the sort's event on the device even then -- the cross-stream ordering stated rather than assumed), "fuse_count" (0 / 1 / 2: the second pass's counts come from the first pass's scatter never / for passes of up to 600 k points / always), "walk_lds_pad" (bytes of unused LDS per k_fuse_block workgroup: fewer workgroups per CU; an experiment's knob),
 *       "ride_events" (0/1: the sort's last dispatch carries the event the walk waits for, instead of a marker recorded behind it),
 *       "copy_threads" (0 .. 16, default 4: caller-owned HOST arrays travel through the handle's pinned staging buffer, this many threads
 *       -- the caller's and process-wide workers on its CCD -- copying between it and the arrays (csrc/gem_hostcopy.hpp); 0 = the arrays are
 *       handed to the runtime as they are), "download_groups" (1 .. 14, default 8: pieces a download of caller-owned host arrays is cut into -- the device
 *       writes piece g + 1 into the staging buffer while the copy threads move piece g on; profiles/r06_download_groups.txt),
 *       "roctx" (0/1: roctx ranges named after the entry points around every call that enqueues or transfers -- the marker library is
 *       loaded at run time, GEM_ERR_INVALID if there is none; `rocprofv3 --kernel-trace --marker-trace` shows them beside the kernels),
 *       "trace" (0/1: one line on stderr per pass of the sorted pipeline), "stream_roles" (a permutation of 0123 as a decimal
 *       number: which of the handle's current own / bin / bin2 / upload streams takes each role; tools/dbg/roles.py).  Returns GEM_ERR_INVALID for an unknown key or a value out of range. */
int gem_debug_set(gem_handle* h, const char* key, long long value);

/* read-outs: "arena_allocations" (device allocations the handle's arenas have made so far: none may follow gem_reserve),
 *            "hstage_allocations" (allocations of the pinned staging buffer: none may follow gem_reserve for clouds / maps within 64 MB),
 *            "xfer_upload_memcpy_ns", "xfer_upload_enqueue_ns", "xfer_download_enqueue_ns", "xfer_download_wait_ns", "xfer_download_memcpy_ns"
 *            (host time spent so far moving caller-owned host arrays: bench.py's node_host_arrays),
 *            "sort_fallbacks" (passes whose forced sorted form / pass count did not fit the map and took the other form),
 *            "walks_left" (walks of the sorted pipeline left to the next call) and "walks_unwaited" (those of them launched without a stream wait: their sort had completed),
 *            "step_pending" (1: the second half of a gem_add_sharded_device step is still to come),
 *            "step_exchange_ns", "step_walk_ns", "step_publish_ns", "step_gather_ns", "step_exchange_to_walk_ns": device time stamps of
 *            the last finished multi-rank step (recorded while gem_set_timing is on; read after gem_synchronize; -2^62 = not recorded),
 *            "step_exchange_bytes_out", "step_exchange_bytes_in" (bytes of sorted records + block ranges this rank sent / received in its last
 *            sharded step), "gather_bytes_out", "gather_bytes_in" (... in its last all-gather of the layers): bench.py --gpus N prices them against the links */
int gem_debug_get(gem_handle* h, const char* key, long long* out);

/* LOOPBACK communicator: nranks handles of THIS process, on ONE device, join the world `world_id` (any number the caller picks,
 * unique per group of handles) as ranks 0 .. nranks - 1, with row strips (tile_strips = 0, like gem_comm_init) or strips of whole
 * tile rows (1, like gem_comm_init_tiles).  From then on gem_add_sharded_device / gem_allgather_layers run the very code of the
 * multi-GPU path -- boundaries, counts, offsets, buffer rotation, stream order -- with the RCCL calls replaced by device-to-device
 * copies ordered by events (csrc/gem_transport.hpp).  Every handle must be driven by a host thread of its own (a collective waits
 * for all ranks, as on real ranks); a send that meets no receive of the same size is an error instead of a hang.  Test
 * infrastructure for one-GPU boxes: it is how the W > 1 code is covered where only one device exists. */
int gem_comm_init_loopback(gem_handle* h, long long world_id, int nranks, int rank, int tile_strips);

/* per-tile cycle stamps of the last fuse launch ([tile][16] 64-bit counters); enable != 0 turns the stamps on for the
 * following passes; with out != NULL copies up to max_tiles rows and returns their number */
int gem_debug_fuse_stamps(gem_handle* h, int enable, unsigned long long* out, int max_tiles);

#ifdef __cplusplus
}
#endif
#endif
