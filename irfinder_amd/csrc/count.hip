// count.hip — the fused per-batch counting kernel (gfx950).
//
// One pass over a structure-of-arrays fragment batch replaces, for every
// fragment, the ProcessBlocks calls of ⟨upstream⟩ ReadBlockProcessor.cpp
// FragmentsInChr / FragmentsInROI / JunctionCount / SpansPoint and of
// ReadBlockProcessor_CoverageBlocks.cpp CoverageBlocks (+ FragmentsMap):
//   * FragmentsInChr : wave-aggregated LDS histogram per workgroup, flushed with global atomics
//   * FragmentsInROI : one bucket-table record (the candidate ROI inline) per fragment
//   * CoverageBlocks : +1 / -1 into the per-strand difference array (depth comes
//                      from the prefix scan at finalize)
//   * SpansPoint     : the difference of two point ranks, both from the cell records the
//                      difference events need anyway (kernels.h: cell tables)
//   * JunctionCount  : one probe of a sparse hash table of the known junctions (ref-sj.ref);
//                      a hit bumps that junction's counter (LDS-cached), novel junctions are
//                      merged per slice in an LDS table, appended as packed
//                      (chr,left,span,strand) records and sort-reduced at finalize
// All outputs are commutative integer sums, so any batch order gives identical
// results.  Integer work bound by memory latency and by the vector-memory instruction rate: no MFMA.
//
// Shape of the kernel (round 3: the block phase no longer searches).  The input is CSR (fragments -> blocks).  A
// workgroup takes SLICES of the (coordinate-sorted) batch, kSlice chunks of 256 consecutive fragments each, and keeps its
// LDS accumulators across the chunks of a slice:
//   F. fragment phase, one fragment per lane: the chromosome record, FragmentsInChr, FragmentsInROI; the fragment's
//      facts (coordinate base, strand, chromosome length) go to LDS and the lane writes its id over the OWNER entries of
//      its blocks (with a first / last flag), so that
//   B. the block phase, one block per lane in strides of 63 (lane 63 only looks ahead), needs no search for the owning
//      fragment: one LDS read gives the owner, one its facts.  Every table question about a position is one 16-byte
//      cell record: the cell of the block start and the cell of the block end give both plane coordinates and both
//      SpansPoint ranks; the junction to the right of the block is one probe of the hash table.  Three table loads per
//      block, all independent of each other.  The successor block's columns, validity and start cell come from the next
//      lane by DPP.  A block owns the junction to its right and the two difference events next to it (its own end, its
//      successor's start): a read on a known junction costs one LDS increment of that junction's counter, which also
//      stands for both events (known_junction_depth_kernel applies them once per junction at finalize).
//   The difference window is circular (cell = plane position mod kWin) and SCROLLS with the chunks: what a chunk's first
//   fragment leaves behind is flushed (one global atomic per non-zero cell, whole 64-byte lines), the rest stays and keeps
//   merging events of later chunks; caches of far lines, junction counters, point counters and novel junctions are flushed
//   every kFlushEvery chunks.
#include "kernels.h"

#include <cstdio>

namespace irf {
namespace dev {

namespace {

#ifdef COUNT_PROF
// diagnostic build only (tools/build_variant.sh -DCOUNT_PROF): wall-clock ticks (10 ns) of waves 0 and 3 per phase and workgroup
__device__ uint32_t g_cprof[2048][2][8];
#define CPROF(k) do { if (lane == 0u && (wave == 0u || wave == 3u)) { const unsigned long long now_ = wall_clock64(); prof_acc[k] += (uint32_t)(now_ - prof_t); prof_t = now_; } } while (0)
#else
#define CPROF(k) do { } while (0)
#endif

#ifndef K_WIN
#define K_WIN 2048
#endif
#ifndef K_LINES
#define K_LINES 128
#endif
#ifndef K_JCACHE
#define K_JCACHE 256
#endif
#ifndef K_PCACHE
#define K_PCACHE 128
#endif
#ifndef K_TILE
#define K_TILE 1024
#endif
#ifndef K_SLICE
#define K_SLICE 16
#endif
#ifndef K_FLUSH
#define K_FLUSH 4
#endif
#ifndef K_WGS_PER_CU
#define K_WGS_PER_CU 5
#endif
#ifndef K_BU
#define K_BU 2
#endif
constexpr int kThreads = 256;
constexpr uint32_t kWaves = kThreads / 64;
constexpr uint32_t kChunk = kThreads;               // fragments per chunk: one per lane
constexpr uint32_t kTile = K_TILE;                  // owner entries: blocks of a chunk are processed in tiles of this many
constexpr uint32_t kSlice = K_SLICE;                // chunks per slice (a workgroup's unit of work; accumulators live this long)
constexpr uint32_t kFlushEvery = K_FLUSH;           // chunks between two flushes of the caches
constexpr int kBU = K_BU;                           // strides a wave works on at once (their table loads are in flight together)
constexpr uint32_t kStride = 63;                    // blocks per wave and step (lane 63 is the look-ahead)
constexpr uint32_t kChrBins = 128;                  // LDS histogram covers refIDs < 64; larger ones go straight to global
constexpr uint32_t kJCache = K_JCACHE;              // LDS cache of known-junction counters (direct-mapped on the junction's row)
constexpr uint32_t kEmpty = 0xffffffffu;
constexpr uint32_t kLines = K_LINES;                // LDS cache of 16-position lines of the difference planes outside the window
constexpr uint32_t kWin = K_WIN;                    // positions of the LDS difference-array window (both strands in one int32)
constexpr uint32_t kPCache = K_PCACHE;              // LDS cache of SpansPoint counters (direct-mapped on the point index)
constexpr uint32_t kNovelQ = 64;                    // LDS queue of novel-junction events (those the table below has no room for)
constexpr uint32_t kNovelH = 128;                   // LDS table that merges novel-junction events by key (open addressing, 4 probes)
constexpr unsigned long long kEmpty64 = ~0ull;      // (no event has bit 63 set)
// The window and the line cache keep both strands of a position as 16-bit halves of one word.  A chunk with more blocks than
// kWinMaxBlocks bypasses them; everything is flushed before the events added since the last complete flush can exceed 2^15.
constexpr uint32_t kWinMaxBlocks = 4000;
constexpr uint32_t kEvFlush = 22000;                // + 2 * kWinMaxBlocks < 32768
constexpr unsigned long long kNoWin = ~0ull;
constexpr uint32_t kFragValid = 1u << 31, kFragDir = 1u << 30, kFragBaseMask = (1u << 28) - 1u;   // base >> 6 (base is a multiple of 64, < 2^34)
constexpr uint32_t kOwnFirst = 0x4000u, kOwnLast = 0x8000u;
static_assert(kWin % 64u == 0u && (kWin & (kWin - 1u)) == 0u, "K_WIN");
static_assert((kLines * 16u) % kThreads == 0u, "the line flush runs with whole waves (DPP sums)");
static_assert(kEvFlush + 2u * kWinMaxBlocks < 32768u, "packed 16-bit halves");

struct Lds {
    uint2 facts[kChunk];               // x = base >> 6 | strand << 30 | counted << 31, y = chromosome length
    uint16_t owner[kTile + 2];         // per block of the tile: lane of the owning fragment | first << 14 | last << 15
    int32_t tile[kWin + 64];           // difference window: low half strand 0, high half strand 1 (+ one idle cell per lane)
    uint32_t ltag[kLines];             // far-event line cache: plane position >> 4
    int32_t lcnt[kLines + 1][16];      // both strands of a position in one word, as in the window; line kLines: idle
    uint32_t jtag[kJCache];            // known-junction cache: row of the known-junction table
    uint32_t jcnt[2][kJCache];
    uint32_t ptag[kPCache];            // SpansPoint cache: index of the point among the points on owned chromosomes
    uint32_t pcnt[2][kPCache];
    unsigned long long nq[kNovelQ];
    unsigned long long nh_key[kNovelH];
    uint32_t nh_cnt[kNovelH];
    uint32_t chr[kChrBins];
    uint32_t wmax[2];                  // highest window offset + 1 a chunk touched
    uint32_t nq_n, nh_n;
};

// Inclusive sums with DPP row shifts / broadcasts: lane 15 of every 16-lane row ends up with the row's total
// (row_sum), lane 63 with the wave's (wave_sum).
__device__ __forceinline__ int row_sum_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);    // row_shr:8
    return v;
}
__device__ __forceinline__ int wave_sum_dpp(int v) {
    v = row_sum_dpp(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v) {          // lane 63 ends up with the wave's maximum
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return v;
}
// Workgroup barrier with the wave's own LDS operations (the block phase's atomics among them) completed first.  The wait is
// spelled out: in front of a barrier at a loop header the compiler was seen to emit none (ROCm 7.2, gfx950: s_barrier right
// behind the back edge's ds_max_u32), and a read behind the barrier then missed an update issued in front of it.
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
}
// lane i gets lane i + 1's value (lane 63: 0); every lane of the wave must be active
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v) {
#ifdef K_NO_WAVE_SHL
    return (uint32_t)__shfl_down((int)v, 1);
#else
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);     // wave_shl:1
#endif
}

// The sum of every 256-position segment of the planes is kept next to them (seg_sum; its exclusive scan is the depth
// carry of stats.hip), so that finalize does not have to read the planes once more just to add them up.
__device__ __forceinline__ void seg_add(const CountTables& T, uint32_t dir, uint64_t c, int v) {
    __hip_atomic_fetch_add(T.seg_sum + (uint64_t)dir * (T.plane_stride >> kSegShift) + (c >> kSegShift), (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (CountTables.dirty: the line of plane position c of strand dir now holds something)
__device__ __forceinline__ void mark_dirty(const CountTables& T, uint32_t dir, uint64_t c) { T.dirty[((c >> 4) << 1) + dir] = 1; }

// A difference event outside the window, at plane position (p5 << 5) + sub: the line cache, else (line slot taken by
// another line, or the chunk bypasses the packed accumulators) a global atomic.
__device__ __forceinline__ void far_event(const CountTables& T, Lds& S, bool use_win, uint32_t p5, uint32_t sub, uint32_t dir, int v) {
    // (a line holds both strands, packed like the window: unstranded reads put both on the same positions; one
    //  compare-and-swap claims an empty slot or returns the line that holds it; the counter update is predicated like the
    //  window's — line kLines is nobody's — so only a conflict is a branch)
    const uint32_t want = (p5 << 1) | (sub >> 4);
    const uint32_t slot = want & (kLines - 1u);
    uint32_t old = want + 1u;
    if (use_win) old = atomicCAS(&S.ltag[slot], kEmpty, want);
    const bool mine = use_win && (old == kEmpty || old == want);
    atomicAdd(&S.lcnt[mine ? slot : kLines][sub & 15u], mine ? (dir ? v * 65536 : v) : 0);
    if (!mine) {
        const uint64_t c = ((uint64_t)p5 << 5) + sub;
        __hip_atomic_fetch_add(T.diff + (uint64_t)dir * T.plane_stride + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seg_add(T, dir, c, v);
        mark_dirty(T, dir, c);
    }
}

// One difference event with a predicate instead of a branch around the call.  w = the plane word of the position's cell
// (kernels.h: cell tables), sub = position & 31.  Inside the window [win5 << 5, + kWin): one LDS add to the circular
// window; the update is executed by every lane (lanes without an event, or with a far one, add 0 to a cell of their own
// behind the window), so that only the far path is a branch.
__device__ __forceinline__ void depth_event(const CountTables& T, Lds& S, uint32_t win5, bool use_win, uint32_t w, uint32_t sub, uint32_t dir, int v,
                                            bool pred, uint32_t lane, uint32_t& wmax) {
    const uint32_t p5 = w & 0x7fffffffu;
    const uint32_t subk = sub & (uint32_t)((int32_t)w >> 31);          // a cell that is not kept maps, whole, to one position
    const uint32_t rel5 = p5 - win5;
    const bool near = pred && use_win && rel5 < (kWin >> 5);
    const uint32_t idx = near ? (((p5 << 5) & (kWin - 1u)) | subk) : kWin + lane;
    atomicAdd(&S.tile[idx], near ? (dir ? v * 65536 : v) : 0);
    const uint32_t top = near ? ((rel5 << 5) | subk) + 1u : 0u;
    wmax = top > wmax ? top : wmax;
    if (pred && !near) far_event(T, S, use_win, p5, subk, dir, v);
}

// One read of strand dir on the known junction of row `row`: the LDS counter, or (slot taken by another junction) straight
// to the table.  The counter also stands for the read's two difference events next to the junction: they are applied once
// per junction and sample by known_junction_depth_kernel at finalize.
__device__ __forceinline__ void junction_add(const CountTables& T, Lds& S, uint32_t row, uint32_t dir) {
    const uint32_t slot = row & (kJCache - 1u);
    uint32_t tag = S.jtag[slot];
    if (tag == kEmpty) {
        const uint32_t old = atomicCAS(&S.jtag[slot], kEmpty, row);
        tag = old == kEmpty ? row : old;
    }
    if (tag == row) atomicAdd(&S.jcnt[dir][slot], 1u);
    else atomicAdd(&T.known_cnt[(uint64_t)dir * T.n_kj + row], 1u);
}

__device__ __forceinline__ void point_add(const CountTables& T, Lds& S, uint32_t pi, uint32_t dir) {
    const uint32_t slot = pi & (kPCache - 1u);
    uint32_t tag = S.ptag[slot];
    if (tag == kEmpty) { const uint32_t old = atomicCAS(&S.ptag[slot], kEmpty, pi); tag = old == kEmpty ? pi : old; }
    if (tag == pi) atomicAdd(&S.pcnt[dir][slot], 1u);
    else atomicAdd(&T.sp_cnt[(uint64_t)dir * T.n_sp + T.sp_row[pi]], 1u);
}

// A read on a novel junction: a packed event for the sort-reduce at finalize.  A slice sits on one locus, so its novel events
// repeat a handful of keys: they are merged in an LDS table and leave as (key, weight).
__device__ __forceinline__ void novel_add(const CountTables& T, Lds& S, unsigned long long ev, uint32_t lane, uint32_t& err) {
    const uint32_t h = (uint32_t)((ev * 0x9E3779B97F4A7C15ull) >> 57);
    bool merged = false;
#pragma unroll
    for (uint32_t pr = 0; pr < 4u; pr++) {
        if (merged) continue;
        const uint32_t sl = (h + pr) & (kNovelH - 1u);
        unsigned long long cur = S.nh_key[sl];
        if (cur == kEmpty64) {
            const unsigned long long old = atomicCAS(&S.nh_key[sl], kEmpty64, ev);
            cur = old == kEmpty64 ? ev : old;
            if (old == kEmpty64) atomicAdd(&S.nh_n, 1u);
        }
        if (cur == ev) { atomicAdd(&S.nh_cnt[sl], 1u); merged = true; }
    }
    if (!merged) {
        const uint32_t q = atomicAdd(&S.nq_n, 1u);
        if (q < kNovelQ) {
            S.nq[q] = ev;
        } else {
            // table and queue full (a run without ref-sj.ref over a dense locus): the lanes that got here reserve slots together
            const uint64_t m = __ballot(1);
            const int lead = __ffsll((long long)m) - 1;
            unsigned long long b = 0;
            if ((int)lane == lead) b = atomicAdd(T.junc_count, (unsigned long long)__popcll(m));
            b = __shfl(b, lead);
            const unsigned long long slot = b + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
            if (slot < T.junc_capacity) { T.junc_keys[slot] = ev; T.junc_w[slot] = 1u; } else err |= ERR_JUNC_CAPACITY;
        }
    }
}

// Window cells of the positions win + [0, to): one global atomic per non-zero cell and strand, consecutive lanes =
// consecutive positions (whole 64-byte lines for the memory-side atomic unit); the cells are left zero.
__device__ __forceinline__ void flush_window(const CountTables& T, Lds& S, unsigned long long win, uint32_t to, uint32_t lane, uint32_t wave) {
    for (uint32_t g = wave * 64u; g < to; g += kWaves * 64u) {
        const uint64_t p = win + g + lane;                              // (win and g are multiples of 64: one segment per wave)
        const uint32_t idx = (uint32_t)p & (kWin - 1u);
        const int32_t v = S.tile[idx];
#ifndef K_NO_SKIP
        if (__ballot(v != 0) == 0ull) continue;                         // (nothing in these 64 positions)
#endif
        const int32_t v0 = (int32_t)(int16_t)(uint16_t)(v & 0xffff);
        const int32_t v1 = (v - v0) >> 16;
        if (v) {
#ifndef K_DIAG_NO_FLUSH_ATOMICS
            int32_t* d = T.diff + p;
            if (v0) __hip_atomic_fetch_add(d, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v1) __hip_atomic_fetch_add(d + T.plane_stride, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            S.tile[idx] = 0;
        }
        // (the two strands' sums in one pass: the packed halves of a sum cannot overflow, see kEvFlush)
        const int32_t sp = wave_sum_dpp(v);
        const int32_t s0 = (int32_t)(int16_t)(uint16_t)(sp & 0xffff), s1 = (sp - s0) >> 16;
        if (lane == 63u) {
#ifndef K_DIAG_NO_FLUSH_ATOMICS
            if (s0) seg_add(T, 0u, p, s0);
            if (s1) seg_add(T, 1u, p, s1);
#endif
            // the wave's four lines, both strands: eight consecutive marks
            *reinterpret_cast<unsigned long long*>(T.dirty + ((p >> 6) << 3)) = 0x0101010101010101ull;
        }
    }
}

// Far-line cache, junction counters, point counters, novel junctions -> the tables; all left empty.
__device__ __forceinline__ void flush_caches(const CountTables& T, Lds& S, uint32_t tid, uint32_t lane, uint32_t wave, uint32_t& err) {
    for (uint32_t e = tid; e < kLines * 16u; e += kThreads) {
        const uint32_t slot = e >> 4, cell = e & 15u;               // a 16-lane row = one line = inside one segment
        const uint32_t tag = S.ltag[slot];
#ifndef K_NO_SKIP
        if (__ballot(tag != kEmpty) == 0ull) continue;              // (none of the wave's four lines is claimed)
#endif
        const int32_t v = tag != kEmpty ? S.lcnt[slot][cell] : 0;
        const int32_t v0 = (int32_t)(int16_t)(uint16_t)(v & 0xffff);
        const int32_t v1 = (v - v0) >> 16;
        const uint64_t c = ((uint64_t)tag << 4) + cell;
        if (v) {
#ifndef K_DIAG_NO_FLUSH_ATOMICS
            if (v0) __hip_atomic_fetch_add(T.diff + c, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v1) __hip_atomic_fetch_add(T.diff + T.plane_stride + c, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            S.lcnt[slot][cell] = 0;
        }
        const int32_t sp = row_sum_dpp(v);
        const int32_t s0 = (int32_t)(int16_t)(uint16_t)(sp & 0xffff), s1 = (sp - s0) >> 16;
        if (cell == 15u && tag != kEmpty) {
#ifndef K_DIAG_NO_FLUSH_ATOMICS
            if (s0) seg_add(T, 0u, c, s0);
            if (s1) seg_add(T, 1u, c, s1);
#endif
            // (a claimed line is marked for both strands whatever it holds: marking too much only costs the clear a line)
            *reinterpret_cast<uint16_t*>(T.dirty + ((uint64_t)tag << 1)) = 0x0101u;
            S.ltag[slot] = kEmpty;          // (the row's 16 lanes have read the tag: same wave, program order)
        }
    }
    for (uint32_t e = tid; e < kJCache; e += kThreads) {
        const uint32_t row = S.jtag[e];
        if (row != kEmpty) {
            const uint32_t c0 = S.jcnt[0][e], c1 = S.jcnt[1][e];
#ifndef K_DIAG_NO_FLUSH_ATOMICS
            if (c0) atomicAdd(&T.known_cnt[row], c0);
            if (c1) atomicAdd(&T.known_cnt[(uint64_t)T.n_kj + row], c1);
#endif
            S.jtag[e] = kEmpty; S.jcnt[0][e] = 0; S.jcnt[1][e] = 0;
        }
    }
    for (uint32_t e = tid; e < kPCache; e += kThreads) {
        const uint32_t pi = S.ptag[e];
        if (pi != kEmpty) {
            const uint32_t c0 = S.pcnt[0][e], c1 = S.pcnt[1][e];
            const uint32_t row = T.sp_row[pi];
            if (c0) atomicAdd(&T.sp_cnt[row], c0);
            if (c1) atomicAdd(&T.sp_cnt[(uint64_t)T.n_sp + row], c1);
            S.ptag[e] = kEmpty; S.pcnt[0][e] = 0; S.pcnt[1][e] = 0;
        }
    }
    if (wave == 0u) {
        // records of the slice so far: the merged table first (compacted with ballots), then the queue (weight 1)
        const uint32_t nq = S.nq_n < kNovelQ ? S.nq_n : kNovelQ, nh = S.nh_n;
        if (nq + nh) {
            unsigned long long qb = 0;
            if (lane == 0u) qb = atomicAdd(T.junc_count, (unsigned long long)(nq + nh));
            qb = __shfl(qb, 0);
            uint32_t at = 0;
#pragma unroll
            for (uint32_t e0 = 0; e0 < kNovelH; e0 += 64u) {
                const unsigned long long k = S.nh_key[e0 + lane];
                const uint32_t w = S.nh_cnt[e0 + lane];
                const uint64_t occ = __ballot(k != kEmpty64);
                if (k != kEmpty64) {
                    const unsigned long long slot = qb + at + (unsigned long long)__popcll(occ & ((1ull << lane) - 1ull));
                    if (slot < T.junc_capacity) { T.junc_keys[slot] = k; T.junc_w[slot] = w; } else err |= ERR_JUNC_CAPACITY;
                    S.nh_key[e0 + lane] = kEmpty64; S.nh_cnt[e0 + lane] = 0;
                }
                at += (uint32_t)__popcll(occ);
            }
            for (uint32_t e = lane; e < nq; e += 64u) {
                if (qb + nh + e < T.junc_capacity) { T.junc_keys[qb + nh + e] = S.nq[e]; T.junc_w[qb + nh + e] = 1u; } else err |= ERR_JUNC_CAPACITY;
            }
            if (lane == 0u) { S.nq_n = 0; S.nh_n = 0; }
        }
    }
}

// Where every chunk puts its window: the plane position, rounded down to 64, of the start of its first fragment; kNoWin: that
// fragment says nothing (not counted, outside its chromosome).  A kernel of its own in front of the count kernel, so that
// no wave of the count kernel walks this chain of three dependent loads while its workgroup waits.
__global__ __launch_bounds__(256) void chunk_origin_kernel(CountTables T, BatchDev B, uint32_t n_chunks) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n_chunks) return;
    const uint32_t f = c * kChunk;
    unsigned long long o = kNoWin;
    const uint32_t chr = B.frag_chr[f], fs = B.frag_start[f];
    if (chr < T.n_chr) {
        const uint4 rec = T.chr_rec[chr];
        if (rec.w && fs <= rec.z) {
            const uint64_t g = ((uint64_t)rec.y << 32 | rec.x) + fs;
            const uint32_t w = T.cell_s[g >> kCellShift].z;
            const uint64_t p = ((uint64_t)(w & 0x7fffffffu) << 5) + ((w >> 31) ? (uint32_t)g & 31u : 0u);
            o = p & ~63ull;
        }
    }
    T.chunk_origin[c] = o;
}

__global__ __launch_bounds__(kThreads, K_WGS_PER_CU) void count_kernel(CountTables T, BatchDev B) {
    __shared__ __align__(16) Lds S;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t k = tid; k < kChrBins; k += kThreads) S.chr[k] = 0;
    for (uint32_t k = tid; k < kWin + 64u; k += kThreads) S.tile[k] = 0;
    for (uint32_t k = tid; k < kLines; k += kThreads) S.ltag[k] = kEmpty;
    for (uint32_t k = tid; k < (kLines + 1u) * 16u; k += kThreads) S.lcnt[k >> 4][k & 15u] = 0;
    for (uint32_t k = tid; k < kJCache; k += kThreads) { S.jtag[k] = kEmpty; S.jcnt[0][k] = 0; S.jcnt[1][k] = 0; }
    for (uint32_t k = tid; k < kPCache; k += kThreads) { S.ptag[k] = kEmpty; S.pcnt[0][k] = 0; S.pcnt[1][k] = 0; }
    for (uint32_t k = tid; k < kNovelH; k += kThreads) { S.nh_key[k] = kEmpty64; S.nh_cnt[k] = 0; }
    if (tid == 0) { S.nq_n = 0; S.nh_n = 0; S.wmax[0] = 0; S.wmax[1] = 0; }

    uint32_t err = 0;
#ifdef COUNT_PROF
    unsigned long long prof_t = wall_clock64();
    uint32_t prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const uint32_t n_chunks = (B.n_frag + kChunk - 1u) / kChunk;
    const uint32_t n_slices = (n_chunks + kSlice - 1u) / kSlice;
    // the columns of a chunk's fragments and the chunk's block range: loaded one chunk ahead, under the block phase
    struct FragCols { uint32_t chr, fs, fe, o0, o1, flg, kb0, kb1; };
    auto load_frag = [&](uint32_t c) -> FragCols {
        FragCols r;
        const uint32_t f0 = c * kChunk, f = f0 + tid;
        const uint32_t fc = f < B.n_frag ? f : B.n_frag - 1u;
        r.chr = B.frag_chr[fc]; r.fs = B.frag_start[fc]; r.fe = B.frag_end[fc];
        r.o0 = B.frag_blk_off[fc]; r.o1 = B.frag_blk_off[fc + 1u]; r.flg = B.frag_flags[fc];
        const uint32_t f_end = f0 + kChunk < B.n_frag ? f0 + kChunk : B.n_frag;
        r.kb0 = B.frag_blk_off[f0]; r.kb1 = B.frag_blk_off[f_end];
        return r;
    };
    for (uint32_t slice = blockIdx.x; slice < n_slices; slice += gridDim.x) {
        const uint32_t c0 = slice * kSlice;
        const uint32_t nc = n_chunks - c0 < kSlice ? n_chunks - c0 : kSlice;
        unsigned long long win = 0;     // window = plane positions [win, win + kWin); uniform over the workgroup, like the next three
        uint32_t wtop = 0;              // cells of win + [0, wtop) may be non-zero
        uint32_t ev_since = 0;          // upper bound of the events added since everything was last flushed
        uint32_t since_flush = 0;       // chunks since the caches were last flushed
        FragCols fc_next = load_frag(c0);
#ifdef K_SLICE_BARRIER
        wg_barrier();
#endif
        for (uint32_t i = 0; i <= nc; i++) {
            CPROF(5);
            wg_barrier();            // ---- the block phase of chunk i - 1 is over (i = 0: the LDS is initialised / flushed) ----
            CPROF(0);
            wtop = umax(wtop, S.wmax[(i + 1u) & 1u]);
#ifdef K_DIAG_FULL_WTOP
            wtop = kWin;
#endif
            const bool last = i == nc;
            const unsigned long long nw = last ? kNoWin : T.chunk_origin[c0 + i];
            if (tid == 0) S.wmax[i & 1u] = 0;

            // ---------------- the window follows the chunks; the caches are flushed every few chunks ----------------
            {
                const bool ev_full = ev_since > kEvFlush;
                bool jump = false;
                uint32_t scroll = 0;
                if (nw != kNoWin) {
                    if (nw < win || nw - win >= kWin) jump = true; else scroll = (uint32_t)(nw - win);
                }
                if (last || ev_full || jump) {
                    flush_window(T, S, win, wtop, lane, wave);
                    wtop = 0;
                } else if (scroll) {
                    flush_window(T, S, win, scroll < wtop ? scroll : wtop, lane, wave);
                    wtop = wtop > scroll ? wtop - scroll : 0u;
                }
                if (nw != kNoWin) win = nw;
                since_flush++;
                if (last || ev_full || since_flush >= kFlushEvery) {
                    flush_caches(T, S, tid, lane, wave, err);
                    since_flush = 0;
                }
                if (last || ev_full) ev_since = 0;
            }
            CPROF(1);
            if (last) break;            // (the slice's accumulators are empty again)

            // ---------------- F. fragment phase of chunk i ----------------
            const uint32_t f0 = (c0 + i) * kChunk, f = f0 + tid;
            const bool in = f < B.n_frag;
            const uint32_t chr = fc_next.chr, fs = fc_next.fs, fe = fc_next.fe, o0 = fc_next.o0, o1 = fc_next.o1, flg = fc_next.flg;
            uint32_t kb0 = fc_next.kb0, kb1 = fc_next.kb1;
            // the CSR offsets (host batches were checked by irf_submit_batch; device-resident ones are checked here): inside
            // the array's bounds, non-decreasing, spanning [0, n_blk].  Clamped, they address nothing out of bounds.
            if (in && (o0 > B.n_blk || o1 > B.n_blk || o1 < o0 || (f == 0u && o0 != 0u) || (f == B.n_frag - 1u && o1 != B.n_blk))) err |= ERR_CSR;
            kb0 = kb0 < B.n_blk ? kb0 : B.n_blk;
            kb1 = kb1 < B.n_blk ? kb1 : B.n_blk;
            if (kb1 < kb0) kb1 = kb0;
            uint32_t o0c = o0 < B.n_blk ? o0 : B.n_blk, o1c = o1 < B.n_blk ? o1 : B.n_blk;
            if (!in) { o0c = 0; o1c = 0; }
            const bool chr_ok = in && chr < T.n_chr;
            if (in && !chr_ok) err |= ERR_CHR_RANGE;
            const uint4 rec = T.chr_rec[chr_ok ? chr : 0u];                // the table holds at least one entry
            const bool counted = chr_ok && rec.w != 0u;
            const uint64_t base = (uint64_t)rec.y << 32 | rec.x;
            const uint32_t clen = rec.z;
            const uint32_t dir = flg & 1u;
            S.facts[tid] = counted ? make_uint2((uint32_t)(base >> 6) | (dir ? kFragDir : 0u) | kFragValid, clen) : make_uint2(0u, 0u);
            // the owner entries of the fragment's blocks (first tile; the others below)
            auto scatter = [&](uint32_t t0) {
                for (uint32_t k = o0c > t0 ? o0c : t0; k < o1c && k - t0 <= kTile; k++)
                    S.owner[k - t0] = (uint16_t)(tid | (k == o0c ? kOwnFirst : 0u) | (k + 1u == o1c ? kOwnLast : 0u));
            };
            scatter(kb0);
            // FragmentsInROI: first ROI (by end) whose end >= fragment end; counted if it contains the fragment
            if (T.n_roi_dev) {
                const bool extent_ok = counted && fs <= fe && fe <= clen;
                const uint64_t gfs = base + fs, gfe = base + fe;
                const uint4 R = T.roi_bkt[extent_ok ? gfe >> kBucketShift : 0ull];
                if (extent_ok && R.x < T.n_roi_dev) {
                    uint32_t q = R.x;
                    uint64_t rgs = (uint64_t)(R.w & 0xffffu) << 32 | R.y;
                    const uint64_t rge = (uint64_t)(R.w >> 16) << 32 | R.z;
                    bool have = true;
                    if (rge < gfe) {              // several ROIs end inside the bucket before the fragment does
                        q++;
                        while (q < T.n_roi_dev && T.roi_gend[q] < gfe) q++;
                        have = q < T.n_roi_dev;
                        if (have) rgs = T.roi_gstart[q];
                    }
#ifndef K_DIAG_NO_ROI_ATOMICS
                    if (have && rgs <= gfs) atomicAdd(&T.roi_cnt[(uint64_t)dir * T.n_roi + T.roi_idx[q]], 1u);
#endif
                }
            }
            // FragmentsInChr: one LDS atomic per distinct (chromosome, strand) of the wave
            {
                const uint32_t k = counted ? chr * 2u + dir : kEmpty;
                uint64_t todo = __ballot(k != kEmpty);
                while (todo) {
                    const int lead = __ffsll((long long)todo) - 1;
                    const uint32_t lk = (uint32_t)__builtin_amdgcn_readlane((int)k, lead);
                    const uint64_t same = __ballot(k == lk);
                    if ((int)lane == lead) {
                        const uint32_t n = (uint32_t)__popcll(same);
                        if (lk < kChrBins) atomicAdd(&S.chr[lk], n);
                        else atomicAdd(&T.chr_cnt[lk], n);
                    }
                    todo &= ~same;
                }
            }
#ifndef K_NO_PREFETCH
            if (i + 1u < nc) fc_next = load_frag(c0 + i + 1u);      // (in flight under the block phase)
#ifdef K_PREFETCH_WAIT
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#endif
            CPROF(2);
            wg_barrier();            // ---- facts and owners of chunk i are in place, the flushed cells are zero ----
            CPROF(3);

            // ---------------- B. block phase: one block per lane and stride of 63, kBU strides at once ----------------
            const uint32_t nblk_c = kb1 - kb0;
            const bool use_win = nblk_c <= kWinMaxBlocks;
            if (use_win) ev_since += 2u * nblk_c;
            const uint32_t win5 = (uint32_t)(win >> 5);
            uint32_t wmax = 0;
            struct Cols { uint32_t bs, bl, rd; };
            auto load_cols = [&](uint32_t k) -> Cols {
                const uint32_t kc = k < B.n_blk ? k : B.n_blk - 1u;      // (n_blk > 0 wherever this runs)
                Cols c;
                c.bs = B.blk_start[kc]; c.bl = B.blk_len[kc]; c.rd = B.blk_read[kc];
                return c;
            };
            for (uint32_t t0 = kb0; t0 < kb1;) {
                const uint32_t tile_n = kb1 - t0 < kTile ? kb1 - t0 : kTile;
                const uint32_t n_str = (tile_n + kStride - 1u) / kStride;
                for (uint32_t s0 = wave; s0 < n_str; s0 += kWaves * kBU) {
                    // (a stride beyond the tile's last does nothing: `here` is false in every lane)
                    Cols cur[kBU];
                    uint32_t ti[kBU], ow[kBU], fb[kBU], clen_b[kBU], be[kBU], bdir[kBU];
                    bool here[kBU], own[kBU], ok[kBU];
                    uint4 rs[kBU], re[kBU], je[kBU];
#pragma unroll
                    for (int u = 0; u < kBU; u++) {
                        ti[u] = (s0 + u * kWaves) * kStride + lane;
                        cur[u] = load_cols(t0 + ti[u]);
                        here[u] = t0 + ti[u] < kb1 && ti[u] <= kTile;        // the block exists and has an owner entry in this tile
                        own[u] = here[u] && lane < kStride && ti[u] < tile_n;
                        ow[u] = S.owner[here[u] ? ti[u] : 0u];
                    }
#pragma unroll
                    for (int u = 0; u < kBU; u++) {
                        const uint2 fa = S.facts[ow[u] & 255u];
                        fb[u] = here[u] ? fa.x : 0u; clen_b[u] = fa.y;
                    }
#pragma unroll
                    for (int u = 0; u < kBU; u++) {
                        const uint32_t bs = cur[u].bs;
                        be[u] = bs + cur[u].bl;
                        const bool in_range = be[u] >= bs && be[u] <= clen_b[u];
                        const bool valid = (fb[u] & kFragValid) != 0u;
                        if (own[u] && valid && !in_range) err |= ERR_BLOCK_RANGE;
                        ok[u] = valid && in_range;
                        bdir[u] = (fb[u] & kFragDir) ? 1u : 0u;
                        const uint32_t cbase = (fb[u] & kFragBaseMask) << 1;      // base >> 5
                        // the two cell records: plane coordinates of start and end, point ranks of both range ends
                        rs[u] = T.cell_s[ok[u] ? cbase + (bs >> kCellShift) : 0u];
                        re[u] = T.cell_e[ok[u] ? cbase + (be[u] >> kCellShift) : 0u];
                    }
                    uint32_t bs1[kBU], span[kBU], klo[kBU], khi[kBU], slot[kBU];
                    uint64_t gbe[kBU];
                    bool has_next[kBU], jc[kBU];
#pragma unroll
                    for (int u = 0; u < kBU; u++) {
                        // the successor (next lane): junction to the right of the block — both flanks long enough, same read,
                        // same fragment
                        bs1[u] = from_next_lane(cur[u].bs);
                        const uint32_t bl1 = from_next_lane(cur[u].bl), rd1 = from_next_lane(cur[u].rd);
                        const bool ok1 = from_next_lane(ok[u] ? 1u : 0u) != 0u;
                        has_next[u] = own[u] && ok[u] && !(ow[u] & kOwnLast) && ok1;
                        jc[u] = has_next[u] && rd1 == cur[u].rd && cur[u].bl >= T.junc_min_flank && bl1 >= T.junc_min_flank;
                        span[u] = bs1[u] - be[u];
                        if (jc[u] && (bs1[u] < be[u] || span[u] >= (1u << kJuncSpanBits))) { err |= ERR_JUNC_RANGE; jc[u] = false; }
                        gbe[u] = ((uint64_t)(fb[u] & kFragBaseMask) << 6) + be[u];
                        klo[u] = ((uint32_t)gbe[u] << kJuncSpanBits) | span[u];
                        khi[u] = (uint32_t)(gbe[u] >> (32 - kJuncSpanBits));
                        slot[u] = junction_hash(klo[u], khi[u]) & T.jh_mask;
                        je[u] = T.jhash[jc[u] ? slot[u] : 0u];
                        asm volatile("" : "+v"(je[u].z));       // (the row comes with the key, not in a second load behind the compare)
                    }
#pragma unroll
                    for (int u = 0; u < kBU; u++) {
                        const uint32_t bs = cur[u].bs, bl = cur[u].bl;
                        const uint32_t rsz1 = from_next_lane(rs[u].z);
                        // known junction?  (the probe continues only past a slot another key holds: load <= 1/32)
                        bool hit = jc[u] && je[u].x == klo[u] && je[u].y == khi[u];
                        {
                            bool more = jc[u] && !hit && (je[u].x & je[u].y) != kEmpty;
                            while (more) {
                                slot[u] = (slot[u] + 1u) & T.jh_mask;
                                je[u] = T.jhash[slot[u]];
                                hit = je[u].x == klo[u] && je[u].y == khi[u];
                                more = !hit && (je[u].x & je[u].y) != kEmpty;
                            }
                        }
                        // JunctionCount: a known junction's counter also stands for the read's two difference events here
                        if (hit) junction_add(T, S, je[u].z, bdir[u]);
                        if (jc[u] && !hit) novel_add(T, S, (gbe[u] << kJuncLeftShift) | ((unsigned long long)span[u] << 1) | bdir[u], lane, err);
                        // CoverageBlocks / FragmentsMap: the block's end, its successor's start (unless the junction between
                        // them was counted on a known junction), and its own start if it opens the fragment
                        depth_event(T, S, win5, use_win, re[u].z, be[u] & 31u, bdir[u], -1, own[u] && ok[u] && !hit, lane, wmax);
                        depth_event(T, S, win5, use_win, rsz1, bs1[u] & 31u, bdir[u], 1, has_next[u] && !hit, lane, wmax);
                        depth_event(T, S, win5, use_win, rs[u].z, bs & 31u, bdir[u], 1, own[u] && ok[u] && (ow[u] & kOwnFirst) != 0u, lane, wmax);
                        // SpansPoint: the points in [bs + sp_lo_add, be - sp_hi_sub] = ranks n0 .. n1 - 1
                        if (own[u] && ok[u] && bl > T.sp_min_len) {
                            const uint32_t n0 = rs[u].x + (uint32_t)__popc(rs[u].y & ((1u << (bs & 31u)) - 1u));
                            const uint32_t n1 = re[u].x + (uint32_t)__popc(re[u].y & ((1u << (be[u] & 31u)) - 1u));
                            for (uint32_t p = n0; p < n1; p++) point_add(T, S, p, bdir[u]);
                        }
                    }
                }
                t0 += kTile;
                if (t0 < kb1) {         // a chunk with more blocks than one tile holds: the next tile's owners
                    wg_barrier();
                    scatter(t0);
                    wg_barrier();
                }
            }
            wmax = wave_max_dpp(wmax);
            if (lane == 63u && wmax) atomicMax(&S.wmax[i & 1u], wmax);
#ifdef K_NO_PREFETCH
            if (i + 1u < nc) fc_next = load_frag(c0 + i + 1u);
#endif
            CPROF(4);
        }
    }
#ifdef COUNT_PROF
    if (lane == 0u && (wave == 0u || wave == 3u) && blockIdx.x < 2048u) for (int k = 0; k < 8; k++) g_cprof[blockIdx.x][wave ? 1 : 0][k] = prof_acc[k];
#endif
    if (err) atomicOr(T.err_flags, err);
    wg_barrier();
    for (uint32_t k = tid; k < kChrBins; k += kThreads) {
        const uint32_t v = S.chr[k];
        if (v && (k >> 1) < T.n_chr) atomicAdd(&T.chr_cnt[k], v);
    }
}

// Every read on a known junction skipped two difference events in the count kernel (the end of the block before the
// junction, the start of the block after it): they are applied here, once per junction and strand, from the junction's
// counter.  Must run after the last count kernel of the sample and before the carries are scanned.
// The kernel also takes the strandedness vote (JunctionCount::Directional, OQ-5: known junctions only): irf_finalize
// needs it before it can start the statistics kernels, and this is the first pass over the junction counters.
__global__ __launch_bounds__(256) void known_junction_depth_kernel(CountTables T) {
    __shared__ unsigned long long s_votes[2];
    if (threadIdx.x < 2u) s_votes[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long same = 0, opp = 0;
    // (a fixed grid that walks the table: the vote leaves as two atomics per workgroup, not two per wave on one address)
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < T.n_known; i += gridDim.x * 256u) {
        const uint32_t row = T.known_kjidx[i];
        int cnt[2];
        cnt[0] = (int)T.known_cnt[row];
        cnt[1] = (int)T.known_cnt[(uint64_t)T.n_kj + row];
        const uint32_t kd = T.known_dir[i];
        if (kd == 1u) { same += (uint32_t)cnt[0]; opp += (uint32_t)cnt[1]; }
        else if (kd == 2u) { same += (uint32_t)cnt[1]; opp += (uint32_t)cnt[0]; }
        if (!(cnt[0] | cnt[1])) continue;
        const uint64_t key = T.known_keys[i];
        const uint64_t gl = key >> kJuncSpanBits, gr = gl + (key & ((1ull << kJuncSpanBits) - 1ull));
        const uint64_t cl = plane_coord(T.pmap, gl), cr = plane_coord(T.pmap, gr);
#pragma unroll
        for (uint32_t dir = 0; dir < 2u; dir++) {
            const int c = cnt[dir];
            if (!c) continue;
            __hip_atomic_fetch_add(T.diff + (uint64_t)dir * T.plane_stride + cl, -c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(T.diff + (uint64_t)dir * T.plane_stride + cr, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seg_add(T, dir, cl, -c);
            seg_add(T, dir, cr, c);
            mark_dirty(T, dir, cl);
            mark_dirty(T, dir, cr);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { same += __shfl_xor(same, d); opp += __shfl_xor(opp, d); }
    if ((threadIdx.x & 63u) == 0u) {
        if (same) atomicAdd(&s_votes[0], same);
        if (opp) atomicAdd(&s_votes[1], opp);
    }
    __syncthreads();
    if (threadIdx.x < 2u && s_votes[threadIdx.x]) atomicAdd(&T.dir_votes[threadIdx.x], s_votes[threadIdx.x]);
}

// Zeroes the marked lines of the planes and the marks.  A lane reads 16 marks (a wave: 1024 marks = 512 lines of both
// strands, one coalesced load); the groups that hold a mark are then taken one at a time by the whole wave, four lanes
// per mark (16 bytes each of the 64-byte line), so that the stores of a group cover whole lines next to each other.
__global__ __launch_bounds__(256) void sparse_clear_kernel(int32_t* __restrict__ planes, uint8_t* dirty, uint64_t plane_stride, uint64_t n_groups) {
    typedef uint32_t V4 __attribute__((ext_vector_type(4)));
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t q = lane >> 4, sh = ((lane >> 2) & 3u) * 8u;
    const uint64_t wave0 = ((uint64_t)blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (uint64_t)gridDim.x * 4u;
    for (uint64_t g0 = wave0 << 6; g0 < n_groups; g0 += n_waves << 6) {       // the wave's next 64 groups
    const uint64_t g = g0 + lane;
    uint4 m = make_uint4(0u, 0u, 0u, 0u);
    if (g < n_groups) m = reinterpret_cast<const uint4*>(dirty)[g];
    const bool any = (m.x | m.y | m.z | m.w) != 0u;
    uint64_t todo = __ballot(any);
    if (any) reinterpret_cast<uint4*>(dirty)[g] = make_uint4(0u, 0u, 0u, 0u);
    while (todo) {
        const int s = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const uint32_t mx = (uint32_t)__builtin_amdgcn_readlane((int)m.x, s), my = (uint32_t)__builtin_amdgcn_readlane((int)m.y, s),
                       mz = (uint32_t)__builtin_amdgcn_readlane((int)m.z, s), mw = (uint32_t)__builtin_amdgcn_readlane((int)m.w, s);
        const uint32_t w = q == 0u ? mx : (q == 1u ? my : (q == 2u ? mz : mw));
        if ((w >> sh) & 0xffu) {
            const uint64_t mark = (g0 + (uint64_t)s) * 16u + (lane >> 2);
            const uint64_t line = mark >> 1, dir = mark & 1u;
            *reinterpret_cast<V4*>(planes + dir * plane_stride + line * 16u + (lane & 3u) * 4u) = V4{0u, 0u, 0u, 0u};
        }
    }
    }
}

// The two cell tables (kernels.h).  One thread per cell: the rank of the cell's first shifted position among the sorted
// points (binary search), the points of its 32 shifted positions as bits, and the plane word.
__global__ __launch_bounds__(256) void build_cells_kernel(const uint64_t* __restrict__ points, uint32_t n_points, const uint32_t* __restrict__ pmap,
                                                          uint64_t n_cells, int32_t shift_s, int32_t shift_e, uint4* __restrict__ cell_s,
                                                          uint4* __restrict__ cell_e) {
    const uint64_t c = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (c >= n_cells) return;
    uint32_t word;
    if (pmap) {
        const uint32_t m = pmap[c >> (kBucketShift - kCellShift)];
        const uint64_t pc = (uint64_t)(m >> 1) << kBucketShift;
        const uint64_t plane = (m & 1u) ? pc + ((c & ((1u << (kBucketShift - kCellShift)) - 1u)) << kCellShift) : pc;
        word = (uint32_t)(plane >> kCellShift) | ((m & 1u) << 31);
    } else {
        word = (uint32_t)c | (1u << 31);
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const long long lo = (long long)(c << kCellShift) + (t == 0 ? shift_s : shift_e);
        uint32_t a = 0, b = n_points;                      // first point >= lo
        while (a < b) {
            const uint32_t mid = a + ((b - a) >> 1);
            if ((long long)points[mid] < lo) a = mid + 1u; else b = mid;
        }
        uint32_t bits = 0;
        for (uint32_t j = a; j < n_points && (long long)points[j] < lo + 32; j++) bits |= 1u << (uint32_t)((long long)points[j] - lo);
        (t == 0 ? cell_s : cell_e)[c] = make_uint4(a, bits, word, 0u);
    }
}

}   // namespace

void launch_build_cells(const uint64_t* points, uint32_t n_points, const uint32_t* pmap, uint64_t n_cells, int32_t shift_s, int32_t shift_e,
                        uint4* cell_s, uint4* cell_e, hipStream_t s) {
    if (!n_cells) return;
    hipLaunchKernelGGL(build_cells_kernel, dim3((uint32_t)((n_cells + 255u) / 256u)), dim3(256), 0, s, points, n_points, pmap, n_cells, shift_s, shift_e,
                       cell_s, cell_e);
}

void launch_known_junction_depth(const CountTables& T, hipStream_t s) {
    if (T.n_known == 0) return;
    const uint32_t wgs = (T.n_known + 255u) / 256u;
    hipLaunchKernelGGL(known_junction_depth_kernel, dim3(wgs < 256u ? wgs : 256u), dim3(256), 0, s, T);
}

void launch_sparse_clear(int32_t* planes, uint8_t* dirty, uint64_t plane_stride, hipStream_t s) {
    const uint64_t n_groups = dirty_map_bytes(plane_stride) / 16u;
    if (!n_groups) return;
    // (two workgroups per CU, looping: the clear is not on the critical path — the next sample takes the OTHER spare set —
    //  and must leave room for the copy kernels of the table export and for the next count kernel, which start next to it)
    const uint64_t wgs = (n_groups + 255u) / 256u;
    hipLaunchKernelGGL(sparse_clear_kernel, dim3((uint32_t)(wgs < 512u ? wgs : 512u)), dim3(256), 0, s, planes, dirty, plane_stride, n_groups);
}

void launch_count(const CountTables& T, const BatchDev& B, hipStream_t s) {
    if (B.n_frag == 0) return;
    const uint32_t n_chunks = (B.n_frag + kChunk - 1u) / kChunk;
    const uint32_t n_slices = (n_chunks + kSlice - 1u) / kSlice;
    const uint32_t most = 256u * K_WGS_PER_CU;
    hipLaunchKernelGGL(chunk_origin_kernel, dim3((n_chunks + 255u) / 256u), dim3(256), 0, s, T, B, n_chunks);
    hipLaunchKernelGGL(count_kernel, dim3(n_slices < most ? n_slices : most), dim3(kThreads), 0, s, T, B);
#ifdef COUNT_PROF
    {
        static uint32_t h[2048][2][8];
        const uint32_t blocks = n_slices < most ? n_slices : most;
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cprof), sizeof h);
        for (int w = 0; w < 2; w++) {
            double a[8] = {0};
            for (uint32_t b = 0; b < blocks && b < 2048u; b++) for (int k = 0; k < 8; k++) a[k] += h[b][w][k];
            const double n = blocks < 2048u ? blocks : 2048u;
            fprintf(stderr, "count_prof wave %d, 10 ns ticks per workgroup: wait at barrier 1 %.0f | flushes %.0f | fragment phase %.0f | wait at barrier 2 %.0f | block phase %.0f | loop overhead %.0f\n",
                    w ? 3 : 0, a[0] / n, a[1] / n, a[2] / n, a[3] / n, a[4] / n, a[5] / n);
        }
    }
#endif
}

}   // namespace dev
}   // namespace irf
