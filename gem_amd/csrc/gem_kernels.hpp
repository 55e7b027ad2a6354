// gem_kernels.hpp -- argument blocks and host launchers of the gfx950 kernels (internal header).
#pragma once

#include "gem_device.hpp"

namespace gem {

constexpr int kMaxPending = 4;  // queued Mapvar_update increments folded into the next fuse
constexpr int kMaxRanks = 8;    // devices of one node the tiling spans

// (sweep, tile, unit) descriptor word, 16 bits:  start << 7 | count  (a unit holds 64 records; 0 = empty)
constexpr int      kSegCountBits  = 7;
constexpr uint32_t kSegCountMask  = (1u << kSegCountBits) - 1u;
constexpr uint32_t kSegStartMask  = 63u;
constexpr uint32_t kFlagEpochMax  = 0xfffffff0u;      // touched-flag stamps of a pass; the flag table is cleared when this is reached

struct LayerPtrs {
    float *elevation, *variance, *intensity, *traver, *lowest;
    int   *colorR, *colorG, *colorB;
    float *rough, *slope;           // outputs of the traversability stage (not touched by init / clear, like the reference's scratch arrays)
};

// A "unit" is 64 consecutive points of one sweep, binned by one wave.  Unit u owns record slots
// [64 u, 64 (u + 1)) of the arena and one word per tile in its sweep's block of the descriptor table.
struct BinArgs {
    FrameConst frame0;                 // single-sweep call: the frame, by value (SGPRs)
    // batched call (n_sweeps > 1): per-sweep tables in device memory
    const FrameConst* frames;          // [n_sweeps]
    const int*        sweep_unit0;     // [n_sweeps+1] first unit of each sweep
    const long long*  sweep_first;     // [n_sweeps+1] first point of each sweep in the concatenated cloud
    const int*        sweep_orig0;     // [n_sweeps] original index of the sweep's first point (a big cloud cut into sweeps), or NULL
    int               n_sweeps;
    long long         n;               // total points
    // input, SRC 0: interleaved XYZI (+ optional packed rgb, original pixel index)
    const float4*   xyzi;
    const uint32_t* rgb;
    const int*      orig;
    // input, SRC 1: Fuse()'s arrays (GPU:1154)
    const int*   f_index; const float* f_height; const float* f_var;
    const int*   f_R; const int* f_G; const int* f_B; const float* f_I;
    // tiling
    int T;                             // number of tiles
    int tile_bits;                     // ceil(log2(T))
    uint32_t epoch;                    // stamps the touched flags of this pass
    int tiles_per_row;
    int B;                             // number of units (grid size)
    int Bpad;                          // units of the longest sweep, multiple of 32: row length of the descriptor table
    // outputs
    uint4*    rec;                     // [B * U] records of rec_words words (gem_kernels.hip: rec_load)
    int       rec_words;               // 3 = {h, var, cell}; 4 = ... + the point's index (colours are fused)
    uint16_t* seg;                     // [n_sweeps][T][Bpad]  descriptor words (all-zero between passes)
    uint32_t* flag;                    // [T][n_sweeps]  == epoch when the sweep put a record into the tile
    uint32_t* gflag;                   // [n_sweeps][T][Bpad/32]  == epoch when that group of 32 units did
    unsigned long long* counters;      // optional: [0] += binned points
    uint32_t*         srt_top;         // bump pointer of the sorted arena (reset here, used by the fuse of the same pass)
    int               keep_sentinel;   // keep records with h == -1 (GPU:482) for the LOWEST fuse variants
    unsigned long long* dbg;           // optional: [blocks][16] cycle stamps of the binning blocks (word 0 start, 1 end, 15 = block index + 1; profiling aid)
};

struct FuseArgs {
    const uint4*    rec;
    uint16_t* seg;                     // [n_sweeps][T][Bpad]  consumed words are zeroed
    const uint32_t* flag;              // [T][n_sweeps]
    const uint32_t* gflag;             // [n_sweeps][T][Bpad/32]
    uint32_t epoch;
    int   B_total;                     // all units of the pass
    int   U;                           // records per unit slot (64)
    int   n_sweeps;
    const int* sweep_unit0;            // [n_sweeps+1] (NULL when n_sweeps == 1: units [0, B_total))
    int   Bpad;                        // units of the longest sweep: row length of seg
    int   T, tiles_per_row, L;
    int   center_tr, center_tc;        // tile holding the map centre (storage coordinates): blocks are mapped to tiles centre-first
    int   row0, row1;                  // owned storage rows
    float mahal, var_floor;
    int   dense;                       // 1: visit every tile (pending variance increments / floor not yet established)
    int   n_pending; float pending[kMaxPending];   // applied before sweep 0 (GPU:540-547)
    const float* var_updates;          // [n_sweeps] applied before each sweep (batched call), or NULL
    float *elevation, *variance;
    // attributes (ATTR 1: xyzi.w + packed rgb, ATTR 2: Fuse() arrays)
    float* intensity; int *colorR, *colorG, *colorB;
    const float4* xyzi; const uint32_t* rgb;
    const int* f_R; const int* f_G; const int* f_B; const float* f_I;
    unsigned long long* counters;      // optional: [1] += distinct touched cells per sweep (per pass when count_per_pass)
    uint4*            srt;             // sorted arena (as many records as `rec`): dense tiles counting-sort their records by cell into it
    uint32_t*         srt_top;         // its bump pointer (records), zeroed by the k_bin of the pass
    float*            lowest;          // map_lowest layer (GEOGRAPHIC cell order), maintained by the LOWEST kernel variants
    int               start0, start1;  // circular-buffer start (storage -> geographic cell)
    int               dbg_sweep;       // debug stamps of the dense path: which sweep
    uint32_t          dense_min;       // a (tile, sweep) with more records than this takes the dense path (16x16 tiles only)
    int   count_per_pass;              // the sweeps are one cloud cut into pieces: count a cell once
    unsigned long long* dbg;           // optional: [T][16] cycle-counter stamps of thread 0 (profiling aid)
};

// ---- the sorted pipelines of big passes (gem_sort.hip) -----------------------------------------------------------------------
// Record key, 32 bits:  id | sweep << id_bits,  id = tile << 10 | cell in its 32x32 tile  (id_bits = 10 + bits of the tile index).
// A BLOCK is a quarter of a tile: eight rows of 32 cells, block id = id >> 8, cell in its block = id & 255.
// Two forms (gem_capi.cpp picks one per pass, both give the same map):
//   CELL-sorted : stable counting-sort passes over digits that cover the whole id, lowest digit first; every cell's records end
//                 up as one contiguous run, k_fuse_walk streams the runs (single dense clouds: a depth image's cells hold hundreds
//                 of points each, the whole chip has to take part in sorting them);
//   BLOCK-sorted: the digits cover the block id only (ONE pass for maps of up to kOnePassMaxBins blocks); inside a block the
//                 records stay in input order and k_fuse_block orders them by cell in LDS, a batch at a time (batches of sweeps,
//                 aggregated clouds: a few records per cell and sweep).
struct SortArgs {
    FrameConst frame0;                 // single-sweep call: the frame, by value
    const FrameConst* frames;          // [n_sweeps]    (batched call)
    const int*        sweep_chunk0;    // [n_sweeps+1]  first pass-1 chunk of each sweep (chunks never span sweeps); NULL = one sweep
    const long long*  sweep_first;     // [n_sweeps+1]  first point of each sweep in the concatenated cloud
    const int*        sweep_orig0;     // [n_sweeps]    original index of the sweep's first point, or NULL
    int               orig0_single;    // the same for a single-sweep call (no tables)
    int               n_sweeps;
    int               sweep_id0;       // id of the first sweep in the record keys (0; the first GLOBAL sweep of a multi-GPU shard)
    long long         n;               // total points
    const float4*   xyzi; const uint32_t* rgb; const int* orig;                       // SRC 0
    const int* f_index; const float* f_height; const float* f_var;                    // SRC 1: Fuse()'s arrays (GPU:1154)
    const int* f_R; const int* f_G; const int* f_B; const float* f_I;
    int keep_sentinel;                 // keep records with h == -1 (GPU:482) for the LOWEST walk
    int few_bins;                      // 0 = kFewBins (debug knob)
    int rank_by_ballot;                // k_sort_scatter matches equal bins by ballot in every pass (what coarse digits take anyway; debug knob:
                                       // the two ways of ranking -- through the LDS and by ballot -- are checked against each other by the tests)
    CameraConst cam;                   // SRC 3 (input colourisation): id = the pixel a point samples, record = {point index, 0}
    int tiles_per_row, T;              // 32x32-cell tiles
    int id_bits;
    int n_passes;                      // counting-sort passes over the id: two digits, three for maps with more than 2^20 cells
    int dshift[3], dbits[3], dbins[3]; // digit i = (id >> dshift[i]) & ((1 << dbits[i]) - 1), dbins[i] values
    int n_chunks1;                     // chunks of pass 1
    uint32_t *cnt[3], *segtot[3];      // per pass: [chunks][bins] per-chunk counts -> prefixes over the chunks of a segment; [4][bins] segment sums
    uint32_t *seg_cnt;                 // [chunks of pass 1][4] records kept by each of k_sort_project's four waves: they are stored COMPACTED at the
                                       // head of the wave's 1024-slot segment of the chunk (the rejected points leave no record)
    uint32_t *total;                   // [0] records kept by pass 1 (in the map, in the strip, accepted)
    uint32_t *blk_cnt;                 // [4 T] or NULL: k_sort_project adds the records of every block (id >> 8); launch_block_prefix turns them into ranges
    uint32_t *bin_base;                // [bins of the last pass + 1] first record of every highest-digit bin in the final order (what k_fuse_walk searches in)
    uint2 *hv_a, *hv_b;                // {h, var}: a = input order, then after the even passes; b = after the odd passes (the passes ping-pong)
    uint32_t *key_a, *key_b, *src_a, *src_b;      // keys; source point | colour flag << 31 (src only when colours are fused)
    unsigned long long* counters;      // optional: [0] += records
    uint32_t* odd_flag; uint32_t epoch; // k_sort_project stores `epoch` here when a record's h or v lies outside the plain range of the walks' chain loops
    int chunk;                         // records per counting-sort chunk of this pass: kSortChunkRecords, or kSortChunkSmall for passes that would not fill the chip otherwise (sort_chunk_for)
    int fuse_count;                    // 1 (n_passes >= 2, small passes): pass 1's scatter counts pass 2's digit on the way out -- no k_sort_count launch for pass 2;
                                       // k_sort_project clears the count rows pass 2 can have (gem_capi.cpp: kFuseCountMaxPoints)
};

// one counting-sort pass over a digit of the key
struct PassArgs {
    const uint2* hv_in; const uint32_t* key_in; const uint32_t* src_in;
    uint2* hv_out; uint32_t* key_out; uint32_t* src_out;
    uint32_t* cnt; const uint32_t* segtot;       // [chunks][bins] counts -> prefixes over the chunks of a segment; [4][bins] segment sums (k_sort_scan)
    int n_chunks;                                // pass 1: chunks of the pass (pass 2 derives them from *n_dev)
    int bins, shift, digit_bits; uint32_t mask;
    const uint32_t* n_dev; long long n_host;     // number of items: on the device (pass 2) or known to the host (pass 1)
    const uint32_t* seg_cnt;                     // pass 1: [chunks][4] records at the head of each 1024-slot segment (k_sort_project)
    const int* sweep_chunk0; const long long* sweep_first; int n_sweeps;   // pass 1 of a batched call: sweep-aligned chunks
    uint32_t* bin_base;                          // last pass: [bins + 1] published by workgroup 0
    unsigned long long* counters;
    int few_bins;                                // COHERENT ranking: ballots per wave instruction before the LDS way takes over (kFewBins)
    int coherent;                                // coarse digits (block-sorted form): consecutive records mostly share their bin
    int chunk;                                   // records per chunk (SortArgs::chunk)
    uint32_t* next_cnt; int next_bins, next_shift; uint32_t next_mask;   // SortArgs::fuse_count: the NEXT pass's count table and digit (NULL: that pass counts for itself)
};

struct WalkArgs {
    const uint2* hv; const uint32_t* key; const uint32_t* src; const uint32_t* bin_base;
    int   T, tiles_per_row, L, row0, row1, id_bits, bin_shift, n_sweeps;
    int   walk_order;                  // 1: blocks take the tile rows centre-first (center_tr = the tile row holding the map centre)
    int   center_tr;
    float mahal, var_floor;
    int   dense;                       // 1: visit every cell (pending variance increments / floor not yet established)
    int   n_pending; float pending[kMaxPending];
    const float* var_updates;          // [n_sweeps] applied before each sweep, or NULL
    float *elevation, *variance, *lowest;
    int   start0, start1;
    float* intensity; int *colorR, *colorG, *colorB;
    const float4* xyzi; const uint32_t* rgb;
    const int* f_R; const int* f_G; const int* f_B; const float* f_I;
    unsigned long long* counters;      // optional: [1] += distinct touched cells (per sweep unless count_per_pass)
    int   count_per_pass;
    // k_fuse_block only:
    unsigned long long* dbg;           // optional: [blocks][16] cycle stamps of thread 0 (profiling aid, gem_debug_fuse_stamps)
    const uint2* ranges;               // single source, optional: [4 T] every block's records {first, end} (k_block_prefix) instead of bin_base + search
    int   lane_sort;                   // 1: the block's cells are handed to the threads in descending order of their record count in the first batch
    int light_blocks;                  // k_fuse_block: rounds of 512 records instead of 2048 (blocks of a few hundred records)
    int   exact_bins;                  // 1: the last pass's bins ARE the blocks (one-pass sort): bin_base gives a block's records without a search
    const uint32_t* odd_flag; uint32_t epoch;   // k_fuse_walk: *odd_flag == epoch: some record of the pass is outside the plain range (k_sort_project)
    int   light_fast;                  // k_fuse_block, rounds of 512: arrival slots + a sorting network per cell instead of rank / scan / placement (0 = off, debug knob)
    int   lds_pad;                     // k_fuse_block: unused dynamic LDS on top of what the kernel needs (debug knob)
    int   prio_records;                // k_fuse_block: blocks of at least this many records raise their waves' issue priority (0 = off)
    int   plain_env;                   // 1: floor, threshold and every variance increment of the pass lie in the range the walks' plain chain loop assumes (walk_plain_env)
    // multi-GPU strip owner (gem_add_sharded_device): the block-sorted records received from every rank, taken in rank order
    int   n_src;                       // <= 1: the single source above (hv / key / src, searched through bin_base)
    const uint2* src_hv[kMaxRanks]; const uint32_t* src_key[kMaxRanks]; uint32_t src_n[kMaxRanks];
    // optional per source: its block ranges (k_block_prefix), entry 0 = block blk0 (the strip's first); positions count from src_base
    // in the source's own arrays, src_hv / src_key point at position src_base.  NULL: the block's records are found by search.
    const uint2* src_ranges[kMaxRanks]; uint32_t src_base[kMaxRanks]; uint32_t blk0;
};

// Are the pass's constants inside the range the plain chain loops assume (gem_sort.hip, k_fuse_block): floor in [2^-28, 2^28], a
// finite positive threshold, every increment in [0, 2^18] (at most 512 + 4 of them: their sum stays below 2^28)?
inline int walk_plain_env(float var_floor, float mahal, const float* pending, int n_pending, const float* var_updates, int n_sweeps)
{
    auto inc_ok = [](float u) { return u >= 0.0f && u <= 262144.0f; };
    bool ok = var_floor >= 3.7252902984619140625e-9f && var_floor <= 268435456.0f && mahal > 0.0f && mahal <= 268435456.0f && n_sweeps <= 512;
    for (int i = 0; i < n_pending; ++i) ok = ok && inc_ok(pending[i]);
    if (var_updates) for (int i = 0; i < n_sweeps && ok; ++i) ok = ok && inc_ok(var_updates[i]);
    return ok ? 1 : 0;
}

struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };   // optional dispatch time-stamps
struct SortShape { int nt, chunk; size_t lds; };
SortShape  sort_shape(int bins, bool attr, int chunk);   // workgroup shape of a pass with that many bins and that chunk
hipError_t launch_sort(hipStream_t st, const SortArgs& a, int src, bool attr, const LaunchEvents ev[9]);   // project, scan, scatter | count, scan, scatter | (count, scan, scatter)
hipError_t launch_walk(hipStream_t st, const WalkArgs& a, int flags, LaunchEvents ev);                    // cell-sorted records
hipError_t launch_block_prefix(hipStream_t st, uint32_t* blk_cnt, int n_blocks, uint2* ranges);   // ranges[b] = {first, end} of block b in the sorted records; leaves blk_cnt zero
hipError_t launch_block_walk(hipStream_t st, const WalkArgs& a, int flags, LaunchEvents ev);              // block-sorted records
constexpr int kOnePassMaxBins = 2048;  // block-sorted: maps of up to this many blocks are sorted by ONE counting-sort pass
hipError_t launch_strip_bounds(hipStream_t st, const uint32_t* keys, const uint32_t* n_records, int id_bits, const uint32_t* ids, uint32_t* out, int n);
constexpr int kSortChunkRecords = 4096, kSortSegsPerChunk = 4;   // records per counting-sort chunk; wave segments per pass-1 chunk (seg_cnt words: a quarter of a chunk each)
constexpr int kSortChunkSmall = 1024;  // ... of passes too small to fill the chip with chunks of 4096 (a 307 200-point depth image: 75 workgroups on 256 CUs)
// The chunk of a pass of n points: every kernel of the sort launches one workgroup per chunk, and below ~2 chunks per CU the chip idles
// (round 4, C3: k_sort_project / count / scatter at 75 workgroups, 51 us for 8.6 MB).
inline int sort_chunk_for(long long n, int forced = 0)
{
    if (forced == kSortChunkRecords || forced == kSortChunkSmall) return forced;
    return n / kSortChunkRecords < 2 * 256 ? kSortChunkSmall : kSortChunkRecords;
}
constexpr int kSortMaxBins = 8000;     // bins per pass the sorted pipeline handles (LDS of k_sort_scatter)

hipError_t launch_project(hipStream_t st, const FrameConst& fc, int first, int n, float* x, float* y, float* z, const int* orig,
                          int write_back, int* map_idx, float* var, float* xt, float* yt, float* zt);

hipError_t launch_bin(hipStream_t st, const BinArgs& a, int src, int ts, LaunchEvents ev);
hipError_t launch_fuse(hipStream_t st, const FuseArgs& a, int ts, int attr, int variant, LaunchEvents ev);
hipError_t launch_frame(hipStream_t st, const FuseArgs& fuse_prev, const BinArgs& bin_this, int attr, LaunchEvents ev);   // attr: 0, or 4 = lowest tracking
size_t     fuse_lds_bytes(int ts, int variant, int attr);
hipError_t launch_init(hipStream_t st, const LayerPtrs& m, int cells, int clear_lowest);
hipError_t launch_unpack_aos(hipStream_t st, const void* src, int n, int step, int ox, int oy, int oz, int oi, int orgb, float4* xyzi, uint32_t* rgb);
hipError_t launch_raytracing(hipStream_t st, const LayerPtrs& m, int L, int start0, int start1, float sensor_z, float obstacle_threshold,
                             int row0, int row1, uint32_t* list, uint32_t* counts, int parity, float* lowest_snapshot, int depth = 4, int lanes = 16);
                             // depth: loads in flight per lane (4, 8); lanes per ray (1, 4, 8, 16); list: L * L words; counts: two words (even / odd calls),
                             // this call's zero on entry; lowest_snapshot: L * L floats (the walks read it, the layer itself is reset to 10)
hipError_t launch_clear_strip(hipStream_t st, const LayerPtrs& m, int L, int start, int count, int is_row);
hipError_t launch_dense_variance(hipStream_t st, float* variance, int cells, int n_pending, const float* pending,
                                 int apply_floor, float var_floor);
hipError_t launch_update_height(hipStream_t st, float* elevation, int cells, float dz);
hipError_t launch_map_feature(hipStream_t st, const float* elevation, float* traver, float* rough, float* slope,
                              int L, float res, int sx, int sy, int row0, int row1);
hipError_t launch_show(hipStream_t st, const LayerPtrs& m, int L, int sx, int sy, double map_length, double resolution, double px, double py,
                       uint32_t* block_count, float* visual, float* xyz, unsigned char* rgb, unsigned char* image, uint32_t* total);

// input colourisation (EMg.cpp:349-381) on the records sorted by pixel
struct ColorArgs {
    const uint32_t* key; const uint2* hv; const uint32_t* total;       // records sorted by pixel (stable: ascending point index inside a pixel)
    uint32_t* first;                   // [width * height] first record of every pixel, ~0 = none   (pre-set to ~0)
    uint32_t* pix;                     // [n] the pixel of every point, ~0 = not in the image          (pre-set to ~0)
    uint32_t* link;                    // [n] the earlier point whose circle this point samples, itself = the image
    int n, width, height;
    const unsigned char* image; size_t stride;     // BGR8 rows
    float4* xyzi; uint32_t* rgb;       // intensity zeroed / 0x00RRGGBB written
};
hipError_t launch_colorize(hipStream_t st, const ColorArgs& a);
// device -> pinned host staging, several pieces per launch (k_copy_list)
constexpr int kCopyListMax = 12;
struct CopyPiece { void* dst; const void* src; size_t bytes; };
struct CopyList { CopyPiece piece[kCopyListMax]; int n; };
hipError_t launch_copy_list(hipStream_t st, const CopyList& l);
hipError_t launch_export_gridmap(hipStream_t st, const void* src, const float* elevation, float* dst, int L, int is_int);

} // namespace gem
