// kernels.h — launch interface between the engine (engine.hip) and the gfx950
// kernels (count.hip, scan.hip, radix.hip, stats.hip).  Plain structs of device
// pointers; every launcher is asynchronous on the given stream.
#pragma once
#include <hip/hip_runtime.h>
#include "wg_barrier.h"

#include <cstdint>

namespace irf {
namespace dev {

// ---- layout of the packed junction key -----------------------------------
// key = (global_left << 29) | (span << 1) | dir      (63 bits used)
//   global_left = chr_base[refID] + left   (< 2^34)
//   span        = right - left             (< 2^28, the BAM cigar length limit)
constexpr int kJuncSpanBits = 28;
constexpr int kJuncLeftShift = kJuncSpanBits + 1;
constexpr uint64_t kJuncMaxGlobal = 1ull << 34;
constexpr int kSegShift = 8;        // 256-position segments of the planes: one sum / depth carry each (kDepthSegment)
constexpr int kBucketShift = 9;     // 512-position buckets of the full coordinate space (point / junction indices, plane map)

// Plane map.  Only the buckets that hold measured intron bases are kept in the difference planes (about a third of
// GRCh38): pmap[b] = (index of the bucket among the kept ones << 1) | kept.  A position in a bucket that is not kept
// maps to the start of the next kept bucket — the prefix sum at every kept position is unchanged by that, because the
// map is monotone (an event still precedes exactly the kept positions it preceded before).
__host__ __device__ inline uint64_t plane_coord(const uint32_t* pmap, uint64_t g) {
    if (!pmap) return g;
    const uint32_t m = pmap[g >> kBucketShift];
    const uint64_t c = (uint64_t)(m >> 1) << kBucketShift;
    return (m & 1u) ? c + (g & ((1u << kBucketShift) - 1u)) : c;
}

enum ErrBits : uint32_t {
    ERR_CHR_RANGE = 1u,        // frag_chr >= n_chr
    ERR_BLOCK_RANGE = 2u,      // block end beyond the chromosome length
    ERR_JUNC_RANGE = 4u,       // junction span or coordinate does not fit the packed key
    ERR_JUNC_CAPACITY = 8u,    // junction event buffer full (host sizing bug)
    ERR_STATS_DEPTH = 16u,     // depth beyond what the rank-select levels cover
    ERR_CSR = 32u              // frag_blk_off of a device-resident batch: not non-decreasing, beyond n_blk, or not spanning [0, n_blk]
};

// ---- tables of the count kernel ------------------------------------------------
// Everything the count kernel asks about a POSITION comes from one 16-byte record of the 32-position cell the position
// lies in: the plane coordinate (for the difference events) and the rank of the position among the SpansPoint points (a
// block counts the points in [bs + sp_lo_add, be - sp_hi_sub]: the difference of two ranks, and the ranks are the indices
// of the points).  Two tables, one asked with block starts, one with block ends; their point bits are pre-shifted by the
// overhang, so that the cell of bs answers for bs + sp_lo_add and the cell of be for be - sp_hi_sub + 1:
//   x = points in front of the cell's first (shifted) position, y = bit i: a point at shifted position i of the cell,
//   z = (plane coordinate of the cell's first position >> 5) | kept << 31 — a cell of a bucket that is not kept maps, whole,
//       to the start of the next kept bucket (plane_coord above) —, w = the row in sp_cnt of point x (the points of one
//       chromosome have consecutive rows, so point x + t of the same block has row w + t).
constexpr int kCellShift = 5;
__host__ __device__ inline uint32_t junction_hash(uint32_t key_lo, uint32_t key_hi) {
    uint32_t x = key_lo * 0x9E3779B1u + key_hi * 0x85EBCA77u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 13;
    return x;
}

struct CountTables {
    uint32_t n_chr;
    uint32_t n_roi;
    uint32_t n_sp;
    uint32_t junc_min_flank;
    uint32_t sp_min_len;       // block is considered only if len > sp_min_len (= L + R)
    uint32_t n_roi_dev;
    uint32_t n_kj;             // rows of the known-junction table
    uint32_t n_known;          // entries of known_keys
    uint32_t jh_mask;          // slots of jhash - 1
    uint32_t pad0;
    uint64_t plane_stride;     // elements per strand plane of the difference array (compact coordinates)
    uint64_t junc_capacity;
    const uint4* chr_rec;      // [n_chr] x, y = offset of the chromosome in the full coordinate space, z = length, w = 0 not owned, 1 owned, 2 owned in part
    const uint2* chr_own;      // [n_chr] for w = 2: fragments that do not overlap [x, y) are ignored, FragmentsInChr counts the starts inside it
    const uint4* cell_s;       // [(full_size >> 5) + 2] asked with block starts (see above)
    const uint4* cell_e;       // ... with block ends
    // known junctions (ref-sj.ref) on owned chromosomes, open addressing (linear probing, load <= 1/32) on
    // junction_hash(key): x, y = key (global_left << 28 | span; all ones = empty), z = row of the known-junction table
    const uint4* jhash;
    // FragmentsInROI: per 512-position bucket the first ROI (by end) that ends at or after the bucket's start:
    // x / y = low words of its start / end (full coordinates, < 2^34), z = its index among the sorted ROIs (< 2^28) |
    // bits 32, 33 of the start << 28 | of the end << 30, w = its index in ref-ROI.bed (the row of roi_cnt)
    const uint4* roi_bkt;
    const uint64_t* roi_gstart;   // ROIs of owned chromosomes sorted by end, in full coordinates (the walk behind roi_bkt)
    const uint64_t* roi_gend;
    const uint32_t* roi_idx;      // file index of the sorted ROI
    uint32_t* roi_cnt;            // [2][n_roi]
    uint32_t* chr_cnt;            // [n_chr][2]
    uint32_t* sp_cnt;             // [2][n_sp]
    uint32_t* known_cnt;          // [2][n_kj]
    int32_t* diff;                // [2][plane_stride]  +1 at block start, -1 at block end
    uint32_t* seg_sum;            // [2][plane_stride >> kSegShift]  sum of every 256-position segment of diff
    // One byte per 64-byte line of the planes ([plane_stride / 16][2 strands]): set by whatever adds to the line (or may
    // have).  The planes of a sample are mostly untouched lines (introns at low depth), so clearing them for the next
    // sample (launch_sparse_clear) writes the marked lines only instead of all of them.
    uint8_t* dirty;
    uint64_t* junc_keys;          // novel-junction event buffer: packed keys ...
    uint32_t* junc_w;             // ... and their weights (a chunk's events on one junction are merged in LDS first)
    unsigned long long* junc_count;
    uint32_t* err_flags;
    // [fragments of the batch / count_chunk_fragments(), rounded up] where the count kernel puts the window of every chunk
    // (written by a kernel of launch_count in front of the count kernel); one more word behind them: the queue the
    // kernel's workgroups take their slices from, zeroed by the same kernel
    unsigned long long* chunk_origin;
    // known_junction_depth_kernel only: the plane map, the sorted keys (global_left << 28 | span), their rows and strands
    // (1 '-', 2 '+', 0/3 unknown) and the two vote sums of JunctionCount::Directional
    const uint32_t* pmap;         // full -> plane coordinates, one entry per 512-position bucket; nullptr = identity
    const uint64_t* known_keys;
    const uint32_t* known_kjidx;
    const uint8_t* known_dir;
    unsigned long long* dir_votes;
};

// cell tables (device-built at irf_chr_map_update): points = sorted full coordinates of the SpansPoint points on owned
// chromosomes; shift_s = sp_lo_add, shift_e = 1 - sp_hi_sub (see above)
// (rows[j] = row in sp_cnt of points[j])
void launch_build_cells(const uint64_t* points, const uint32_t* rows, uint32_t n_points, const uint32_t* pmap, uint64_t n_cells, int32_t shift_s,
                        int32_t shift_e, uint4* cell_s, uint4* cell_e, hipStream_t s);

struct BatchDev {
    uint32_t n_frag, n_blk;
    const uint32_t* frag_chr;
    const uint8_t* frag_flags;
    const uint32_t* frag_start;
    const uint32_t* frag_end;
    const uint32_t* frag_blk_off;
    const uint32_t* blk_start;
    const uint32_t* blk_len;
    const uint8_t* blk_read;
};

// The count kernel reads its tables through a pointer (the block is ~50 dwords: as a by-value argument it filled the scalar
// registers): d_T is a device copy of T that launch_count refreshes (a one-thread kernel on the stream) whenever T differs
// from *shadow, the host's record of what d_T holds.
// launch_count_prepare: the table block into device memory when it changed (shadow = the host's copy of what is there) and
// the chunk origins of the batch; launch_count: the count kernel alone (the engine's timing events bracket only that)
void launch_count_prepare(const CountTables& T, CountTables* d_T, CountTables* shadow, const BatchDev& B, hipStream_t s);
void launch_count(CountTables* d_T, const BatchDev& B, hipStream_t s);
uint32_t count_chunk_fragments();      // fragments per entry of CountTables::chunk_origin
// difference events of the reads on known junctions, from the junction counters: after the sample's last count kernel,
// before the carry scan (once per sample)
void launch_known_junction_depth(const CountTables& T, hipStream_t s);
// zeroes the lines of planes[2][plane_stride] that dirty[] marks, and the marks (dirty: dirty_map_bytes(plane_stride) bytes)
void launch_sparse_clear(int32_t* planes, uint8_t* dirty, uint64_t plane_stride, hipStream_t s);
inline size_t dirty_map_bytes(uint64_t plane_stride) { return (size_t)((plane_stride / 16 * 2 + 15) / 16 * 16); }

// ---- batches in any order (reorder.hip) -------------------------------------
constexpr int kReorderShift = 8;       // fragments are binned by (chromosome base + start) >> kReorderShift
// key of the radix passes = bin << kReorderKeyBinShift | min(block count, kReorderKeyNblkMax) << 32 | fragment index: the
// passes move 8 bytes per record and sort on the bin's bits only (bin < 2^26: up to 2^34 positions).  frag_recs: n_frag
// records of reorder_frag_rec_bytes() (reorder.hip).  launch_reorder_keys also checks the CSR offsets (ERR_CSR) and the
// packed block fields (ERR_BLOCK_RANGE).
constexpr int kReorderKeyBinShift = 38;
constexpr uint32_t kReorderKeyNblkMax = 63;
size_t reorder_frag_rec_bytes();
void launch_reorder_keys(const BatchDev& B, const uint64_t* chr_base, uint32_t n_chr, uint64_t* keys, void* frag_recs, uint32_t* err_flags, hipStream_t s);
// nblk: n_frag + 1 elements (the last one 0: its exclusive scan is the new offsets array, total included)
void launch_reorder_counts(uint32_t n_frag, const uint64_t* sorted_keys, const void* frag_recs, uint32_t* nblk, hipStream_t s);
void launch_reorder_gather(const BatchDev& B, const void* frag_recs, const uint64_t* sorted_keys, const uint32_t* new_off, uint32_t* chr, uint8_t* flags,
                           uint32_t* start, uint32_t* end, uint32_t* bs, uint32_t* bl, uint8_t* rd, hipStream_t s);

// ---- scans ----------------------------------------------------------------
constexpr uint32_t kScanTile = 4096;
// number of uint32 workspace elements a scan over n elements needs
uint64_t scan_workspace_elems(uint64_t n);
// Depth carries: sums of each kDepthSegment-element segment of the difference planes (then scanned exclusively in
// place with launch_exclusive_scan_u32): carry[k] = depth just before position k * kDepthSegment.  n % 4096 == 0.
constexpr uint32_t kDepthSegment = 256;
void launch_segment_sums(const int32_t* diff, uint64_t n, uint32_t* seg_sums, hipStream_t s);
// exclusive prefix sum, out may alias in
void launch_exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* workspace, hipStream_t s);

// ---- junction radix sort-reduce ---------------------------------------------
constexpr uint32_t kRadixTile = 4096;
uint64_t radix_workspace_elems(uint64_t n);     // uint32 elements
// OR / AND over all keys (to skip constant digits); out[0] = OR, out[1] = AND (pre-set to 0 / ~0)
void launch_key_bits(const uint64_t* keys, uint64_t n, unsigned long long* out, hipStream_t s);
// one stable counting pass on digit [shift, shift+8); the weights travel with their keys (win = wout = nullptr: keys only)
void launch_radix_pass(const uint64_t* in, uint64_t* out, const uint32_t* win, uint32_t* wout, uint64_t n, int shift,
                       uint32_t* workspace, hipStream_t s);
// sorted keys -> head flags (1 where key differs from predecessor)
void launch_head_flags(const uint64_t* keys, uint64_t n, uint32_t* flags, hipStream_t s);
// compaction: unique keys + run lengths.  uidx = exclusive scan of flags.
void launch_unique_runs(const uint64_t* keys, const uint32_t* flags, const uint32_t* uidx, uint64_t n,
                        uint64_t* ukeys, uint32_t* ustart, hipStream_t s);
// count of a unique key = sum of the weights of its run; wex = exclusive scan of the sorted weights w
void launch_run_sums(const uint32_t* ustart, uint64_t n_unique, uint64_t n, const uint32_t* wex, const uint32_t* w, uint32_t* ucount,
                     hipStream_t s);

// ---- junction-derived lookups ------------------------------------------------
struct JuncLookupTables {
    const uint64_t* ukeys;       // sorted unique (with dir bit)
    const uint32_t* ucount;
    uint64_t n_unique;
    const uint64_t* ustart_pos;  // sorted unique global intron starts
    const uint64_t* uend_pos;    // sorted unique global intron ends
    uint32_t n_ustart, n_uend;
    uint32_t* left_cnt;          // [2][n_ustart]   SpliceLeft by strand
    uint32_t* right_cnt;         // [2][n_uend]
    // known junctions: counted in place by the count kernel
    const uint64_t* known_keys;  // sorted (global_left << 28 | span) of ref-sj junctions on owned chromosomes
    const uint32_t* known_kjidx;
    const uint8_t* known_dir;    // per known_keys entry: 1 '-', 2 '+', 0/3 unknown
    const int32_t* known_us_idx; // per known_keys entry: index of its left end in ustart_pos / right end in uend_pos, -1: none
    const int32_t* known_ue_idx;
    uint32_t n_known, n_kj;
    const uint32_t* known_cnt;   // [2][n_kj]
    unsigned long long* dir_votes;   // [2]: same, opposite (JunctionCount::Directional); nullptr: the vote was taken elsewhere
};
// novel unique junctions -> SpliceLeft/Right counters
void launch_junction_lookups(const JuncLookupTables& J, hipStream_t s);
// known junctions -> SpliceLeft/Right counters + the Directional() vote
void launch_known_lookups(const JuncLookupTables& J, hipStream_t s);

// ---- multi-GPU collection (gather.hip) ------------------------------------------
// row code of an intron / table row: 0 = on a BAM reference another shard owns, 1 = on a reference this context owns,
// 2 = on no BAM reference at all (kept by the root of a gather); rows whose code is not in keep_bits are zeroed
constexpr uint32_t kRowForeign = 0, kRowMine = 1, kRowOrphan = 2;
void launch_mask_rows_f64(double* data, uint32_t n_cols, uint64_t n, const uint8_t* code, uint32_t keep_bits, hipStream_t s);
void launch_mask_rows_u32(uint32_t* data, uint32_t n_cols, uint64_t n, const uint8_t* code, uint32_t keep_bits, hipStream_t s);
void launch_add_into_f64(double* dst, const double* src, uint64_t n, hipStream_t s);
void launch_add_into_u32(uint32_t* dst, const uint32_t* src, uint64_t n, hipStream_t s);

// ---- per-intron statistics + IR table --------------------------------------------
// Everything a row needs besides depth, one 128-byte record per intron.  The kernel keeps a record one dword per lane
// (stats.hip: header pipeline), so the field order below is its ABI: dword index in the comment.
struct IntronRec {
    uint64_t g_start, g_end;       // 0-3   full coordinates of the intron
    uint64_t measured;             // 4-5   sum of the measured block lengths (N)
    uint32_t blk_off, n_blk;       // 6,7   measured blocks: blks[blk_off .. blk_off + n_blk)
    int32_t sp_s, sp_e;            // 8,9   index into sp_cnt of the point at Start / End (-1: none)
    int32_t us, ue;                // 10,11 index into left_cnt / right_cnt
    int32_t kj;                    // 12    row of the known-junction table for (Start,End), -1 if not in ref-sj
    uint32_t row;                  // 13    row of the IR table (= record index in ref-cover.bed)
    uint32_t flags;                // 14    bit0 valid (chromosome in the BAM header and owned), bit1 '+' strand, bits 8..15 qneed
    uint32_t prem;                 // 15    interpolation remainders of p25 | p50 << 8 | p75 << 16
    uint32_t qrank[8];             // 16-23 the rank queries (stats_queries below)
    uint32_t pad[8];
};
static_assert(sizeof(IntronRec) == 128, "IntronRec is 32 dwords");
struct BlkRec {                    // one measured block
    uint64_t gstart;               // full coordinates
    uint64_t cstart;               // plane (compact) coordinates
    uint32_t len;
    uint32_t pad;
};

// The rank queries every statistic of an intron with N measured bases reduces to (0-based ranks into the sorted
// depths).  q0, q1: "sum of the r smallest" at r = skip and r = N - skip for the trimmed mean (OQ-8:
// skip = floor(N (100 - centre) / 200)); q2..q7: the order statistics k-1 and k of p25, p50, p75 (OQ-7:
// (N + 1) p / 100 = k + rem / 100).  need bit q: the query has to be answered.
inline void stats_queries(uint64_t N, uint32_t trim_centre_pct, bool interpolate, uint32_t qrank[8], uint32_t* need, uint32_t* prem) {
    const uint64_t skip = N * (uint64_t)(100u - trim_centre_pct) / 200ull;
    const uint64_t lo = skip, hi = N - skip;
    *need = 0; *prem = 0;
    qrank[0] = (uint32_t)lo; if (N > 0 && lo > 0 && lo < N) *need |= 1u;
    qrank[1] = (uint32_t)hi; if (N > 0 && hi > 0 && hi < N) *need |= 2u;
    const uint32_t pcts[3] = {25u, 50u, 75u};
    for (int p = 0; p < 3; p++) {
        const uint64_t num = (N + 1ull) * pcts[p];
        const uint64_t k = num / 100ull;
        const uint32_t rem = (uint32_t)(num % 100ull);
        uint64_t ra = 0, rb = 0;
        bool nb = false;
        if (N > 0) {
            if (k < 1) ra = 0;
            else if (k >= N) ra = N - 1;
            else { ra = k - 1; rb = k; nb = interpolate && rem != 0; }
        }
        qrank[2 + 2 * p] = (uint32_t)ra; if (N > 0) *need |= 1u << (2 + 2 * p);
        qrank[3 + 2 * p] = (uint32_t)rb; if (nb) *need |= 1u << (3 + 2 * p);
        *prem |= rem << (8 * p);
    }
}

// recs is laid out by size class (by measured bases) and, inside a class, by workgroup: the host deals the introns of a
// class, longest first, to the least loaded workgroup.  Class 0: 256-thread workgroups, class 1: small introns,
// 64-thread workgroups.  wg_off[c] is a device array of n_wg[c] + 1 record indices.
struct StatsClasses { const uint32_t* wg_off[2]; uint32_t n_wg[2]; };
constexpr uint32_t kStatsMaxFlank = 512;       // flank_len limit (LDS window buffers)
constexpr uint32_t kStatsSmallFlank = 64;      // ... of the 64-thread class
constexpr uint64_t kStatsSmallMax = 8192;      // class 1: measured <= this

struct IntronTables {
    uint32_t n_intron;
    uint32_t n_sp;
    uint64_t plane_stride;
    const int32_t* diff;           // [2][plane_stride] difference events (NOT scanned)
    const uint32_t* carry;         // [2][plane_stride/256] depth just before each 256-position segment
    const IntronRec* recs;         // [n_intron] in launch order
    const BlkRec* blks;
    const uint32_t* pmap;          // full -> plane map (nullptr = identity), for raw-coordinate flank windows
    const uint32_t* sp_cnt;        // [2][n_sp]
    const uint32_t* left_cnt;      // [2][n_ustart]
    const uint32_t* right_cnt;     // [2][n_uend]
    uint32_t n_ustart, n_uend;
    const uint64_t* ukeys;         // novel unique junctions
    const uint32_t* ucount;
    uint64_t n_unique;
    const uint32_t* known_cnt;     // [2][n_kj]
    uint32_t n_kj;
    uint32_t ablate;               // diagnostic only (IRF_ABLATE_STATS): bit0 no depth pass, bit1 no row tail
    // parameters
    uint32_t trim_centre_pct, pct_interpolate, flank_offset, flank_len, flank_use_measured, warn_nonuniform_and;
    int32_t directionality;        // 0: non-directional table (both planes)
    // vote != nullptr: the statistics kernels take the strandedness decision themselves from the two sums of the vote
    // (JunctionCount::Directional(): same > opp * dir_ratio -> 1, opp > same * dir_ratio -> -1, else 0) instead of
    // `directionality`, and a launch whose shape does not fit the decision returns at once: the single-table launch when
    // the sample is stranded, the dual one when it is not.  The host queues both and reads the vote while they run.
    const unsigned long long* vote;
    uint32_t dir_ratio;
    double warn_lowcover_min, warn_lowsplice_min, warn_minor_ratio, warn_nonuniform_rel, warn_nonuniform_abs;
    // outputs
    double* out_f64;               // [8][n_intron]
    uint32_t* out_u32;             // [6][n_intron]  (5 counts + warning)
    // dual = 1: ONE pass computes the non-directional table into out_* and the directional one (under `directionality`,
    // which must be non-zero) into out_*_b; dual = 0: one table, non-directional if directionality == 0
    uint32_t dual;
    double* out_f64_b;
    uint32_t* out_u32_b;
    uint32_t* err_flags;
};
// d_T: device memory for one IntronTables (the kernels read their table block through it: as a by-value argument its
// ~60 dwords stayed live in scalar registers through the whole kernel, 140-240 of them spilled into vector lanes); the block
// is stored there by a one-thread kernel in front of the launch, in stream order
void launch_intron_stats(const IntronTables& T, IntronTables* d_T, const StatsClasses& K, hipStream_t s);
// The junction-dependent columns of the rows (SpliceLeft / SpliceRight / SpliceExact, IRratio, the warning) once more,
// from the depth columns launch_intron_stats has written: irf_finalize starts the statistics kernels before the
// novel-junction table is finished (T.n_unique = 0 there) and runs this afterwards with the finished tables.
void launch_ir_rows(const IntronTables& T, hipStream_t s);

}   // namespace dev
}   // namespace irf
