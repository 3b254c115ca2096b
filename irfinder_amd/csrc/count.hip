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
//                      merged in an LDS table, appended as packed (chr,left,span,strand)
//                      records and sort-reduced at finalize
// All outputs are commutative integer sums, so any batch order gives identical
// results.  Integer work bound by memory latency and by the vector-memory instruction rate: no MFMA.
//
// Shape of the kernel (round 3).  The input is CSR (fragments -> blocks).  A workgroup takes SLICES of the (coordinate-sorted)
// batch, kSlice chunks of kFPL x kThreads consecutive fragments each, and keeps its LDS accumulators across the chunks of a
// slice; its first slice by its index, every further one from a queue (slices differ in cost).  Per chunk, between two
// workgroup barriers each:
//   F. fragment phase, kFPL fragments per lane: the chromosome record (a copy of the table in LDS), FragmentsInChr,
//      FragmentsInROI; the fragment's facts (coordinate base, strand, chromosome length) go to LDS and the lane writes its
//      id over the OWNER entries of its blocks (with a first / last flag).  Then, with every load of the phase landed, the
//      waves share the FLUSH of what the previous chunks left in LDS: the part of the difference window the new chunk has
//      scrolled past, the overflow queue, and every few chunks the caches (far lines, junction / point / ROI counters,
//      novel junctions).
//   B. block phase, one block per lane in strides of 63 (lane 63 only looks ahead): one LDS read gives the owner, one its
//      facts — no search.  Every table question about a position is one 16-byte cell record: the cell of the block start
//      and the cell of the block end give both plane coordinates and both SpansPoint ranks; the junction to the right of
//      the block is one probe of the hash table.  Three table loads per block, all independent of each other; the block
//      columns of a pass are requested one pass ahead.  The successor block's columns, validity and start cell come from
//      the next lane by DPP.  A block owns the junction to its right and the two difference events next to it (its own
//      end, its successor's start): a read on a known junction costs one LDS increment of that junction's counter, which
//      also stands for both events (known_junction_depth_kernel applies them once per junction at finalize).
// Global atomics are issued in the flush only.  gfx9 retires a wave's vector-memory operations in order under one counter,
// and a memory-side atomic takes microseconds to be acknowledged, so a wave that has issued ONE global atomic waits that
// long for its next load (measured on configs[1]: the 1.1 M ROI atomics of an earlier version cost 0.3 of 2.95 ms).  In
// the phases that wait for loads nothing goes to the tables: what does not fit an LDS accumulator goes to an LDS queue
// (only when that is full, too, does a lane touch the tables itself); the flush comes when the fragment phase's loads
// have landed and is followed by a barrier, and chunks are large, so that the one wait behind it is paid once per ~3000 blocks.
// The difference window is circular (cell = plane position mod kWin) and SCROLLS with the chunks: what a chunk's first
// fragment leaves behind is flushed (one global atomic per non-zero cell, whole 64-byte lines), the rest stays and keeps
// merging events of later chunks.
#include "kernels.h"

#include <cstdio>
#include <cstring>

namespace irf {
namespace dev {

namespace {

#ifdef COUNT_PROF
// diagnostic build only (tools/build_variant.sh -DCOUNT_PROF): wall-clock ticks (10 ns) of the first and the last wave per
// phase and workgroup
__device__ uint32_t g_cprof[2048][3][8];
#define CPROF(k) do { if (lane == 0u && prof_on) { const unsigned long long now_ = wall_clock64(); prof_acc[k] += (uint32_t)(now_ - prof_t); prof_t = now_; } } while (0)
// ... and what the flusher did, summed over the launch: 0 window steps looked at, 1 of them not empty, 2 line steps not empty,
// 3 queue entries, 4 chunks whose window jumped, 5 cache flushes, 6 chunks
__device__ unsigned long long g_cstat[8];
#define CSTAT(k, n) do { if (lane == 0u) s_cstat_local[k] += (n); } while (0)
#else
#define CPROF(k) do { } while (0)
#define CSTAT(k, n) do { } while (0)
#endif

#ifndef K_WAVES
#define K_WAVES 8
#endif
#ifndef K_FPL
#define K_FPL 1
#endif
#ifndef K_WIN
#define K_WIN 4096
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
#define K_TILE 4096
#endif
#ifndef K_SLICE
#define K_SLICE 8
#endif
#ifndef K_FLUSH
#define K_FLUSH 2
#endif
#ifndef K_WGS_PER_CU
#define K_WGS_PER_CU 3
#endif
#ifndef K_BU
#define K_BU 1
#endif
constexpr uint32_t kWaves = K_WAVES;
constexpr int kThreads = K_WAVES * 64;
constexpr int kWavesPerSimd = (K_WGS_PER_CU * K_WAVES + 3) / 4;
constexpr int kFPL = K_FPL;                         // fragments per lane and chunk
constexpr uint32_t kChunk = kFPL * kThreads;        // fragments per chunk
constexpr uint32_t kTile = K_TILE;                  // owner entries: blocks of a chunk are processed in tiles of this many
constexpr uint32_t kSlice = K_SLICE;                // chunks per slice (a workgroup's unit of work; accumulators live this long)
constexpr uint32_t kFlushEvery = K_FLUSH;           // chunks between two flushes of the caches
constexpr int kBU = K_BU;                           // strides a wave works on at once (their table loads are in flight together)
constexpr uint32_t kStride = 63;                    // blocks per wave and step (lane 63 is the look-ahead)
constexpr uint32_t kChrTab = 64;                    // references whose record is kept in LDS
constexpr uint32_t kChrBins = 128;                  // LDS histogram covers refIDs < 64; larger ones go through the queue
constexpr uint32_t kJCache = K_JCACHE;              // LDS cache of known-junction counters (direct-mapped on the junction's row)
constexpr uint32_t kEmpty = 0xffffffffu;
constexpr uint32_t kLines = K_LINES;                // LDS cache of 16-position lines of the difference planes outside the window
constexpr uint32_t kWin = K_WIN;                    // positions of the LDS difference-array window (both strands in one int32)
constexpr uint32_t kPCache = K_PCACHE;              // LDS cache of SpansPoint counters (direct-mapped on the point's row)
constexpr uint32_t kRCache = 64;                    // LDS cache of ROI counters (direct-mapped on the ROI's row)
constexpr uint32_t kNovelQ = 64;                    // LDS queue of novel-junction events (those the table below has no room for)
constexpr uint32_t kNovelH = 128;                   // LDS table that merges novel-junction events by key (open addressing, 4 probes)
constexpr uint32_t kOQ = 128;                       // LDS queues (two, taking turns) of updates that found their cache slot taken
constexpr unsigned long long kEmpty64 = ~0ull;      // (no event has bit 63 set)
// The window and the line cache keep both strands of a position as 16-bit halves of one word.  A chunk with more blocks than
// kWinMaxBlocks bypasses them; everything is flushed before the events added since the last complete flush can exceed 2^15.
constexpr uint32_t kWinMaxBlocks = 4000;
constexpr uint32_t kEvFlush = 22000;                // + 2 * kWinMaxBlocks < 32768
constexpr unsigned long long kNoWin = ~0ull;
constexpr uint32_t kFragValid = 1u << 31, kFragDir = 1u << 30, kFragBaseMask = (1u << 28) - 1u;   // base >> 6 (base is a multiple of 64, < 2^34)
constexpr uint32_t kOwnFirst = 0x4000u, kOwnLast = 0x8000u, kOwnMask = 0x3fffu;
static_assert(kChunk <= 0x4000u, "owner ids are 14 bits");
static_assert(kWin % 64u == 0u && (kWin & (kWin - 1u)) == 0u, "K_WIN");
static_assert(kEvFlush + 2u * kWinMaxBlocks < 32768u && 2u * kFlushEvery * kWinMaxBlocks <= 32768u, "packed 16-bit halves");

// entries of the overflow queue: kind << 62 | strand << 61 | ...
constexpr unsigned long long kOqPlane = 0ull << 62;     // | minus << 60 | plane position: one difference event
constexpr unsigned long long kOqKnown = 1ull << 62;     // | count << 32 | row of known_cnt
constexpr unsigned long long kOqPoint = 2ull << 62;     // | count << 32 | row of sp_cnt
constexpr unsigned long long kOqRoi = 3ull << 62;       // | count << 32 | row of roi_cnt;  with bit 60: index into chr_cnt

#define AS1 __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// Pointers read out of the table block are generic to the compiler (flat instructions: both wait counters); cast to the
// global address space where they are used.
template <class X> __device__ __forceinline__ AS1 X* gp(X* p) { return (AS1 X*)p; }

struct Lds {
    uint2 facts[kChunk];               // x = base >> 6 | strand << 30 | counted << 31, y = chromosome length
    uint16_t owner[kTile + 2];         // per block of the tile: lane of the owning fragment | first << 14 | last << 15
    int32_t tile[kWin + 64];           // difference window: low half strand 0, high half strand 1 (+ one idle cell per lane)
    uint32_t ltag[kLines];             // far-event line cache: plane position >> 4
    int32_t lcnt[kLines + 1][16];      // both strands of a position in one word, as in the window; line kLines: idle
    uint32_t jtag[kJCache];            // known-junction cache: row of the known-junction table
    uint32_t jcnt[2][kJCache];
    uint32_t ptag[kPCache];            // SpansPoint cache: row of sp_cnt
    uint32_t pcnt[2][kPCache];
    uint32_t rtag[kRCache];            // ROI cache: row of roi_cnt
    uint32_t rcnt[2][kRCache];
    unsigned long long nq[kNovelQ];
    unsigned long long nh_key[kNovelH];
    uint32_t nh_cnt[kNovelH];
    unsigned long long oq[2][kOQ];
    uint32_t chr[kChrBins];
    u32x4 chrtab[kChrTab];             // chr_rec of the first kChrTab references (the fragment phase then waits for no load)
    // the claimed slots of the four caches, in the order they were claimed (the flusher walks these, not the caches)
    uint16_t llist[kLines], jlist[kJCache], plist[kPCache], rlist[kRCache];
    uint32_t ll_n, jl_n, pl_n, rl_n;
    uint32_t wmax[2];                  // highest window offset + 1 a chunk touched
    uint32_t nq_n, nh_n, oq_n[2], oq_cur;   // oq_cur: the queue the block phase fills (the flush drains the other one)
    // the slice's chunks: window origin of each, first block of each (and of the chunk behind the last); read once per slice
    unsigned long long s_org[kSlice];
    uint32_t s_kb[kSlice + 1];
    uint32_t next_slice;               // the slice this workgroup takes next (from the queue)
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
// lane i gets lane i + 1's value (lane 63: 0); every lane of the wave must be active
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v) {
#ifdef K_NO_WAVE_SHL
    return (uint32_t)__shfl_down((int)v, 1);
#else
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true);     // wave_shl:1
#endif
}

// ---- the tables (global atomics: the flusher wave, and a worker whose queue is full) ----------------------------
// The sum of every 256-position segment of the planes is kept next to them (seg_sum; its exclusive scan is the depth
// carry of stats.hip), so that finalize does not have to read the planes once more just to add them up.
__device__ __forceinline__ void seg_add(const CountTables& T, uint32_t dir, uint64_t c, int v) {
#ifdef K_DIAG_NO_ATOMICS
    return;
#endif
    __hip_atomic_fetch_add(gp(T.seg_sum) + (uint64_t)dir * (T.plane_stride >> kSegShift) + (c >> kSegShift), (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (CountTables.dirty: the line of plane position c of strand dir now holds something)
__device__ __forceinline__ void mark_dirty(const CountTables& T, uint32_t dir, uint64_t c) {
#ifdef K_DIAG_NO_ATOMICS
    return;
#endif
    gp(T.dirty)[((c >> 4) << 1) + dir] = 1; }
__device__ __forceinline__ void add_u32(uint32_t* p, uint32_t v) {
#ifdef K_DIAG_NO_ATOMICS
    return;
#endif
    __hip_atomic_fetch_add(gp(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void oq_apply(const CountTables& T, unsigned long long e) {
    const uint32_t kind = (uint32_t)(e >> 62), dir = (uint32_t)(e >> 61) & 1u;
    const uint32_t idx = (uint32_t)e, n = (uint32_t)(e >> 32) & 0x0fffffffu;
    if (kind == 0u) {
        const uint64_t c = e & ((1ull << 36) - 1ull);
        const int v = (e >> 60) & 1ull ? -1 : 1;
#ifndef K_DIAG_NO_ATOMICS
        __hip_atomic_fetch_add(gp(T.diff) + (uint64_t)dir * T.plane_stride + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        seg_add(T, dir, c, v);
        mark_dirty(T, dir, c);
    } else if (kind == 1u) {
        add_u32(T.known_cnt + (uint64_t)dir * T.n_kj + idx, n);
    } else if (kind == 2u) {
        add_u32(T.sp_cnt + (uint64_t)dir * T.n_sp + idx, n);
    } else if ((e >> 60) & 1ull) {
        add_u32(T.chr_cnt + idx, n);
    } else {
        add_u32(T.roi_cnt + (uint64_t)dir * T.n_roi + idx, n);
    }
}

// ---- worker side: LDS accumulators, and the queue when a slot is taken --------------------------------------------
__device__ __forceinline__ void oq_push(const CountTables& T, Lds& S, unsigned long long e) {
    const uint32_t b = S.oq_cur;
    const uint32_t q = atomicAdd(&S.oq_n[b], 1u);
    if (q < kOQ) S.oq[b][q] = e;
    else oq_apply(T, e);            // (queue full: the worker pays for the atomic itself)
}

// A difference event outside the window, at plane position (p5 << 5) + sub: the line cache, else (line slot taken by
// another line, or the chunk bypasses the packed accumulators) the queue.
__device__ __forceinline__ void far_event(const CountTables& T, Lds& S, bool use_win, uint32_t p5, uint32_t sub, uint32_t dir, int v) {
    // (a line holds both strands, packed like the window: unstranded reads put both on the same positions; one
    //  compare-and-swap claims an empty slot or returns the line that holds it; the counter update is predicated like the
    //  window's — line kLines is nobody's — so only a conflict is a branch)
    const uint32_t want = (p5 << 1) | (sub >> 4);
    uint32_t slot = want & (kLines - 1u);
    uint32_t old = want + 1u;
    if (use_win) {
        old = atomicCAS(&S.ltag[slot], kEmpty, want);
        if (old != kEmpty && old != want) {         // taken by another line: the line's second slot
            slot = (slot ^ (kLines >> 1) ^ ((want >> 8) & (kLines / 2u - 1u))) & (kLines - 1u);
            old = atomicCAS(&S.ltag[slot], kEmpty, want);
        }
        if (old == kEmpty) S.llist[atomicAdd(&S.ll_n, 1u)] = (uint16_t)slot;      // (claimed: at most kLines claims between two flushes)
    }
    const bool mine = use_win && (old == kEmpty || old == want);
    atomicAdd(&S.lcnt[mine ? slot : kLines][sub & 15u], mine ? (dir ? v * 65536 : v) : 0);
    if (!mine) oq_push(T, S, kOqPlane | (unsigned long long)dir << 61 | (v < 0 ? 1ull << 60 : 0ull) | (((unsigned long long)p5 << 5) + sub));
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

// One count of strand dir for row `row` in a direct-mapped LDS cache (tags, two counter rows); slot taken: the queue.
template <uint32_t N>
__device__ __forceinline__ void cache_add(const CountTables& T, Lds& S, uint32_t* tags, uint32_t (*cnt)[N], uint16_t* list, uint32_t* list_n, uint32_t row,
                                          uint32_t dir, unsigned long long oq_kind) {
    const uint32_t slot = row & (N - 1u);
    uint32_t tag = tags[slot];
    if (tag == kEmpty) {
        const uint32_t old = atomicCAS(&tags[slot], kEmpty, row);
        tag = old == kEmpty ? row : old;
        if (old == kEmpty) list[atomicAdd(list_n, 1u)] = (uint16_t)slot;
    }
    if (tag == row) atomicAdd(&cnt[dir][slot], 1u);
    else oq_push(T, S, oq_kind | (unsigned long long)dir << 61 | 1ull << 32 | row);
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
            if ((int)lane == lead) b = __hip_atomic_fetch_add(gp(T.junc_count), (unsigned long long)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b = __shfl(b, lead);
            const unsigned long long slot = b + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
            if (slot < T.junc_capacity) { gp(T.junc_keys)[slot] = ev; gp(T.junc_w)[slot] = 1u; } else err |= ERR_JUNC_CAPACITY;
        }
    }
}

// ---- the flush: LDS accumulators -> the tables, shared by the waves of the workgroup ------------------------------
// Window cells of the positions win + [0, to): one global atomic per non-zero cell and strand, consecutive lanes =
// consecutive positions (whole 64-byte lines for the memory-side atomic unit); the cells are left zero.  A wave takes whole
// 256-position segments, so that the segment's sum and the marks of its lines leave once.
__device__ __forceinline__ void flush_window(const CountTables& T, Lds& S, unsigned long long win, uint32_t to, uint32_t lane, uint32_t wave, uint32_t* s_cstat_local) {
    if (!to) return;
    const uint64_t seg_last = (win + to - 1u) >> 8;
    for (uint64_t seg = (win >> 8) + ((wave + kWaves - (uint32_t)((win >> 8) % kWaves)) % kWaves); seg <= seg_last; seg += kWaves) {
        int32_t v4[4];
        bool in4[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t p0 = (seg << 8) + 64u * k;
            in4[k] = p0 >= win && p0 < win + to;
            v4[k] = in4[k] ? S.tile[(uint32_t)(p0 + lane) & (kWin - 1u)] : 0;
        }
        int32_t acc = 0;                    // packed sums of the segment (lane 63)
        uint32_t marks = 0;                 // bit k: step k of the segment held something
#pragma unroll
        for (int k = 0; k < 4; k++) {
            CSTAT(0, in4[k] ? 1u : 0u);
            const int32_t v = v4[k];
            if (__ballot(v != 0) == 0ull) continue;                     // (nothing in these 64 positions)
            CSTAT(1, 1u);
            const uint64_t p = (seg << 8) + 64u * k + lane;
            const int32_t v0 = (int32_t)(int16_t)(uint16_t)(v & 0xffff);
            const int32_t v1 = (v - v0) >> 16;
            if (v) {
#ifndef K_DIAG_NO_ATOMICS
                AS1 int32_t* d = gp(T.diff) + p;
                if (v0) __hip_atomic_fetch_add(d, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v1) __hip_atomic_fetch_add(d + T.plane_stride, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
                S.tile[(uint32_t)p & (kWin - 1u)] = 0;
            }
            // (the two strands' sums in one pass: the packed halves of a sum cannot overflow, see kEvFlush)
            acc += wave_sum_dpp(v);
            marks |= 1u << k;
        }
        if (marks) {
            const int32_t tot = __builtin_amdgcn_readlane(acc, 63);
            const int32_t t0 = (int32_t)(int16_t)(uint16_t)(tot & 0xffff), t1 = (tot - t0) >> 16;
            if (lane >= 62u) {              // lane 63 adds strand 0's sum, lane 62 strand 1's: one instruction
                const uint32_t dir = 63u - lane;
                const int32_t v = dir ? t1 : t0;
                if (v) seg_add(T, dir, seg << 8, v);
            }
#ifndef K_DIAG_NO_ATOMICS
            // the marks of the segment's 16 lines, both strands: 8 bytes per step that held something
            if (lane < 4u && ((marks >> lane) & 1u)) *reinterpret_cast<AS1 unsigned long long*>(gp(T.dirty) + (((seg << 2) + lane) << 3)) = 0x0101010101010101ull;
#endif
        }
    }
}

// counters of a direct-mapped cache -> a [2][n_rows] table (n: the claimed slots; the caller resets the count)
template <uint32_t N>
__device__ __forceinline__ void flush_cache(uint32_t* tags, uint32_t (*cnt)[N], const uint16_t* list, uint32_t n, uint32_t* table, uint64_t n_rows, uint32_t tid) {
    for (uint32_t i = tid; i < n; i += kThreads) {
        const uint32_t e = list[i];
        const uint32_t row = tags[e];
        const uint32_t c0 = cnt[0][e], c1 = cnt[1][e];
        if (c0) add_u32(table + row, c0);
        if (c1) add_u32(table + n_rows + row, c1);
        tags[e] = kEmpty; cnt[0][e] = 0; cnt[1][e] = 0;
    }
}

// the claimed lines of the far-event cache, four per wave and step
__device__ __forceinline__ void flush_lines(const CountTables& T, Lds& S, uint32_t n_lines, uint32_t lane, uint32_t wave, uint32_t* s_cstat_local) {
    for (uint32_t e0 = wave * 4u; e0 < n_lines; e0 += kWaves * 4u) {
        const uint32_t li = e0 + (lane >> 4), cell = lane & 15u;    // a 16-lane row = one line = inside one segment
        const uint32_t slot = li < n_lines ? S.llist[li] : kLines;
        const uint32_t tag = li < n_lines ? S.ltag[slot] : kEmpty;
        CSTAT(2, 1u);
        const int32_t v = tag != kEmpty ? S.lcnt[slot][cell] : 0;
        const int32_t v0 = (int32_t)(int16_t)(uint16_t)(v & 0xffff);
        const int32_t v1 = (v - v0) >> 16;
        const uint64_t c = ((uint64_t)tag << 4) + cell;
        if (v) {
#ifndef K_DIAG_NO_ATOMICS
            if (v0) __hip_atomic_fetch_add(gp(T.diff) + c, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v1) __hip_atomic_fetch_add(gp(T.diff) + T.plane_stride + c, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
            S.lcnt[slot][cell] = 0;
        }
        const int32_t sp = row_sum_dpp(v);
        // (cell 15 holds the row's sums: it adds strand 0's, cell 14 strand 1's — one instruction)
        const int32_t spn = (int32_t)__builtin_amdgcn_update_dpp(0, sp, 0x101, 0xf, 0xf, true);     // row_shl:1: cell 14 gets cell 15's sums
        if (cell >= 14u && tag != kEmpty) {
            const uint32_t dir = 15u - cell;
            const int32_t t = dir ? spn : sp;
            const int32_t t0 = (int32_t)(int16_t)(uint16_t)(t & 0xffff), t1 = (t - t0) >> 16;
            const int32_t val = dir ? t1 : t0;
            if (val) seg_add(T, dir, c, val);
        }
        if (cell == 15u && tag != kEmpty) {
            // (a claimed line is marked for both strands whatever it holds: marking too much only costs the clear a line)
#ifndef K_DIAG_NO_ATOMICS
            *reinterpret_cast<AS1 uint16_t*>(gp(T.dirty) + ((uint64_t)tag << 1)) = 0x0101u;
#endif
            S.ltag[slot] = kEmpty;          // (the row's 16 lanes have read the tag: same wave, program order)
        }
    }
}

// novel-junction records (one wave): the merged table first (compacted with ballots), then the queue (weight 1)
__device__ __forceinline__ void flush_novel(const CountTables& T, Lds& S, uint32_t lane, uint32_t& err) {
    const uint32_t nq = S.nq_n < kNovelQ ? S.nq_n : kNovelQ, nh = S.nh_n;
    if (!(nq + nh)) return;
    unsigned long long qb = 0;
    if (lane == 0u) qb = __hip_atomic_fetch_add(gp(T.junc_count), (unsigned long long)(nq + nh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    qb = __shfl(qb, 0);
    uint32_t at = 0;
#pragma unroll
    for (uint32_t e0 = 0; e0 < kNovelH; e0 += 64u) {
        const unsigned long long k = S.nh_key[e0 + lane];
        const uint32_t w = S.nh_cnt[e0 + lane];
        const uint64_t occ = __ballot(k != kEmpty64);
        if (k != kEmpty64) {
            const unsigned long long slot = qb + at + (unsigned long long)__popcll(occ & ((1ull << lane) - 1ull));
            if (slot < T.junc_capacity) { gp(T.junc_keys)[slot] = k; gp(T.junc_w)[slot] = w; } else err |= ERR_JUNC_CAPACITY;
            S.nh_key[e0 + lane] = kEmpty64; S.nh_cnt[e0 + lane] = 0;
        }
        at += (uint32_t)__popcll(occ);
    }
    for (uint32_t e = lane; e < nq; e += 64u) {
        if (qb + nh + e < T.junc_capacity) { gp(T.junc_keys)[qb + nh + e] = S.nq[e]; gp(T.junc_w)[qb + nh + e] = 1u; } else err |= ERR_JUNC_CAPACITY;
    }
    if (lane == 0u) { S.nq_n = 0; S.nh_n = 0; }
}

// Where every chunk puts its window: the plane position, rounded down to 64, of the start of its first fragment; kNoWin: that
// fragment says nothing (not counted, outside its chromosome).  A kernel of its own in front of the count kernel, so that
// no wave of the count kernel walks this chain of three dependent loads while its workgroup waits.
__global__ __launch_bounds__(256) void chunk_origin_kernel(CountTables T, BatchDev B, uint32_t n_chunks) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n_chunks) return;
    if (c == 0u) T.chunk_origin[n_chunks] = 0ull;            // the count kernel's slice queue (behind the last chunk's origin)
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

// The slices of a batch.  The queue hands them out in order and the launch lasts until the last one is done: full slices
// (kSlice chunks) over the first K_TAIL_FROM / 16 of the chunks, short ones (kTailSlice chunks) over the rest, so that the
// last workgroups finish close together.  (Three numbers on purpose: a map of four stages kept in registers for the whole
// kernel cost more than the tail it removed, 2.59 against 2.10 ms.)
#ifndef K_TAIL_FROM
#define K_TAIL_FROM 15
#endif
#ifndef K_TAIL_SLICE
#define K_TAIL_SLICE 2
#endif
constexpr uint32_t kTailSlice = K_TAIL_SLICE;
static_assert(kTailSlice >= 1u && kTailSlice <= kSlice && K_TAIL_FROM <= 16, "K_TAIL_SLICE / K_TAIL_FROM");
struct SliceMap { uint32_t n_full, c_tail0, n_slices; };
__host__ __device__ inline SliceMap slice_map(uint32_t n_chunks) {
    SliceMap m;
    m.n_full = (uint32_t)((uint64_t)n_chunks * K_TAIL_FROM / 16u) / kSlice;
    m.c_tail0 = m.n_full * kSlice;
    m.n_slices = m.n_full + (n_chunks - m.c_tail0 + kTailSlice - 1u) / kTailSlice;
    return m;
}

// the count kernel's table block in device memory (launch_count)
__global__ void store_tables_kernel(CountTables T, CountTables* dst) { *dst = T; }

__global__ __launch_bounds__(kThreads, kWavesPerSimd) void count_kernel(const CountTables* __restrict__ Tp, BatchDev B) {
    __shared__ __align__(16) Lds S;
    const CountTables& T = *Tp;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t k = tid; k < kChrBins; k += kThreads) S.chr[k] = 0;
    for (uint32_t k = tid; k < kWin + 64u; k += kThreads) S.tile[k] = 0;
    for (uint32_t k = tid; k < kLines; k += kThreads) S.ltag[k] = kEmpty;
    for (uint32_t k = tid; k < (kLines + 1u) * 16u; k += kThreads) S.lcnt[k >> 4][k & 15u] = 0;
    for (uint32_t k = tid; k < kJCache; k += kThreads) { S.jtag[k] = kEmpty; S.jcnt[0][k] = 0; S.jcnt[1][k] = 0; }
    for (uint32_t k = tid; k < kPCache; k += kThreads) { S.ptag[k] = kEmpty; S.pcnt[0][k] = 0; S.pcnt[1][k] = 0; }
    for (uint32_t k = tid; k < kRCache; k += kThreads) { S.rtag[k] = kEmpty; S.rcnt[0][k] = 0; S.rcnt[1][k] = 0; }
    for (uint32_t k = tid; k < kNovelH; k += kThreads) { S.nh_key[k] = kEmpty64; S.nh_cnt[k] = 0; }
    if (tid == 0) { S.nq_n = 0; S.nh_n = 0; S.oq_n[0] = 0; S.oq_n[1] = 0; S.oq_cur = 0; S.wmax[0] = 0; S.wmax[1] = 0; S.ll_n = 0; S.jl_n = 0; S.pl_n = 0; S.rl_n = 0; }
    for (uint32_t k = tid; k < kChrTab; k += kThreads) S.chrtab[k] = k < T.n_chr ? ((const AS1 u32x4*)T.chr_rec)[k] : u32x4{0u, 0u, 0u, 0u};

    uint32_t err = 0;
    uint32_t s_cstat_local[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // (used by the diagnostic build only)
#ifdef COUNT_PROF
    const bool prof_on = wave == 0u || wave == kWaves - 1u;
    unsigned long long prof_t = wall_clock64();
    uint32_t prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const uint32_t n_chunks = (B.n_frag + kChunk - 1u) / kChunk;
    const SliceMap smap = slice_map(n_chunks);
    const uint32_t n_slices = smap.n_slices;
    const AS1 unsigned long long* origin = gp((const unsigned long long*)T.chunk_origin);
    const AS1 u32x4* cell_s = (const AS1 u32x4*)T.cell_s;
    const AS1 u32x4* cell_e = (const AS1 u32x4*)T.cell_e;
    const AS1 u32x4* jhash = (const AS1 u32x4*)T.jhash;
    const AS1 u32x4* chr_rec = (const AS1 u32x4*)T.chr_rec;
    const AS1 u32x4* roi_bkt = (const AS1 u32x4*)T.roi_bkt;
    const uint32_t jh_mask = T.jh_mask, n_chr = T.n_chr, n_roi_dev = T.n_roi_dev, junc_min_flank = T.junc_min_flank, sp_min_len = T.sp_min_len;
    struct Cols { uint32_t bs, bl, rd; };
    auto load_cols = [&](uint32_t k) -> Cols {
        Cols c = {0u, 0u, 0u};
        if (B.n_blk == 0u) return c;            // (fragments without any block: the arrays may not exist)
        const uint32_t kc = k < B.n_blk ? k : B.n_blk - 1u;
        c.bs = B.blk_start[kc]; c.bl = B.blk_len[kc]; c.rd = B.blk_read[kc];
        return c;
    };

    // the columns of a chunk's fragments: requested one chunk ahead, under the block phase
    struct FragCols { uint32_t chr[kFPL], fs[kFPL], fe[kFPL], o0[kFPL], o1[kFPL], flg[kFPL]; };
    auto load_frag = [&](uint32_t c) -> FragCols {
        FragCols r;
#pragma unroll
        for (int j = 0; j < kFPL; j++) {
            const uint32_t f = c * kChunk + j * kThreads + tid;
            const uint32_t fc = f < B.n_frag ? f : B.n_frag - 1u;
            r.chr[j] = B.frag_chr[fc]; r.fs[j] = B.frag_start[fc]; r.fe[j] = B.frag_end[fc];
            r.o0[j] = B.frag_blk_off[fc]; r.o1[j] = B.frag_blk_off[fc + 1u]; r.flg[j] = B.frag_flags[fc];
        }
        return r;
    };
#ifdef K_STATIC_SLICES
    for (uint32_t slice = blockIdx.x; slice < n_slices; slice += gridDim.x) {
#else
    // Slices cost what their locus costs (blocks per fragment, events outside the window): a workgroup takes its first
    // slice by its index and every further one from a queue (one returning atomic per slice, issued when the slice begins
    // and looked at under its first chunk), so that no workgroup is left with the expensive ones at the end of the launch.
    AS1 uint32_t* const slice_queue = (AS1 uint32_t*)(gp(T.chunk_origin) + n_chunks);
    for (uint32_t slice = blockIdx.x; slice < n_slices; slice = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.next_slice)) {
        uint32_t queued = 0;
        if (tid == 0) queued = gridDim.x + __hip_atomic_fetch_add(slice_queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        const bool full = slice < smap.n_full;
        const uint32_t c0 = full ? slice * kSlice : smap.c_tail0 + (slice - smap.n_full) * kTailSlice;
        const uint32_t cap = full ? kSlice : kTailSlice;
        const uint32_t nc = n_chunks - c0 < cap ? n_chunks - c0 : cap;
        FragCols fc_next = load_frag(c0);
        // (two dependent round trips per chunk otherwise: the origin decides what the flush does, the block range where the
        //  first strides' columns are.  Nobody reads these words between the last barrier of a slice and the first of the next.)
        if (tid < kSlice) S.s_org[tid] = tid < nc ? origin[c0 + tid] : kNoWin;
        if (tid >= 64u && tid - 64u <= kSlice) {
            const unsigned long long f = (unsigned long long)(c0 + tid - 64u) * kChunk;
            S.s_kb[tid - 64u] = B.frag_blk_off[f < B.n_frag ? (uint32_t)f : B.n_frag];
        }
        unsigned long long win = 0;     // window = plane positions [win, win + kWin); uniform over the workgroup, like the next three
        uint32_t wtop = 0;              // cells of win + [0, wtop) may be non-zero
        uint32_t ev_since = 0;          // upper bound of the events added since everything was last flushed
        uint32_t since_flush = 0;       // chunks since the caches were last flushed
        for (uint32_t i = 0; i <= nc; i++) {
            CPROF(3);
            wg_barrier();               // ---- the block phase of chunk i - 1 is over (i = 0: the LDS is initialised / flushed) ----
            CPROF(0);
            const bool last = i == nc;
            // what the flush below will work on (read now: the words are reset behind it)
            wtop = umax(wtop, S.wmax[(i + 1u) & 1u]);      // (i = 0: may be the previous slice's; too large does no harm)
            // (the queue chunk i - 1 filled; a slice begins with both queues drained, whatever their counts say)
            const uint32_t n_oq = i == 0u ? 0u : (S.oq_n[(i + 1u) & 1u] < kOQ ? S.oq_n[(i + 1u) & 1u] : kOQ);
            const unsigned long long win_old = win;
            const unsigned long long nw = last ? kNoWin : S.s_org[last ? 0u : i];
            if (tid == 0) { S.wmax[i & 1u] = 0; S.oq_n[i & 1u] = 0; S.oq_cur = i & 1u; }
            const bool ev_full = ev_since > kEvFlush;
            bool jump = false;
            uint32_t scroll = 0;
            if (nw != kNoWin) {
                if (nw < win || nw - win >= kWin) jump = true; else scroll = (uint32_t)(nw - win);
            }
            uint32_t flush_to = 0;                          // the window cells win_old + [0, flush_to) leave
            if (last || ev_full || jump) { flush_to = wtop; wtop = 0; }
            else if (scroll) { flush_to = scroll < wtop ? scroll : wtop; wtop = wtop > scroll ? wtop - scroll : 0u; }
            if (nw != kNoWin) win = nw;
            since_flush++;
            const bool do_caches = last || ev_full || since_flush >= kFlushEvery;
            if (do_caches) since_flush = 0;
            if (last || ev_full) ev_since = 0;

            // ---------------- F. fragment phase of chunk i ----------------
            uint32_t kb0 = 0, kb1 = 0;
            uint32_t roi_row[kFPL];                   // row of roi_cnt | strand << 31, or kEmpty
            Cols nxt[kBU];
            // the owner entries of the blocks [a, b) of the lane's fragment j that lie in tile t0
            auto scatter = [&](uint32_t t0, int j, uint32_t a, uint32_t b) {
                for (uint32_t k = a > t0 ? a : t0; k < b && k - t0 <= kTile; k++)
                    S.owner[k - t0] = (uint16_t)((j * kThreads + tid) | (k == a ? kOwnFirst : 0u) | (k + 1u == b ? kOwnLast : 0u));
            };
            if (!last) {
                const uint32_t f0 = (c0 + i) * kChunk;
                const uint32_t f_end = f0 + kChunk < B.n_frag ? f0 + kChunk : B.n_frag;
                kb0 = S.s_kb[i]; kb1 = S.s_kb[i + 1u];
                kb0 = kb0 < B.n_blk ? kb0 : B.n_blk;
                kb1 = kb1 < B.n_blk ? kb1 : B.n_blk;
                if (kb1 < kb0) kb1 = kb0;
                uint32_t chr[kFPL], fs[kFPL], fe[kFPL], o0[kFPL], o1[kFPL], flg[kFPL];
#pragma unroll
                for (int j = 0; j < kFPL; j++) {
                    chr[j] = fc_next.chr[j]; fs[j] = fc_next.fs[j]; fe[j] = fc_next.fe[j];
                    o0[j] = fc_next.o0[j]; o1[j] = fc_next.o1[j]; flg[j] = fc_next.flg[j];
                }
                // the block columns of the wave's first strides are requested now as well
#pragma unroll
                for (int u = 0; u < kBU; u++) nxt[u] = load_cols(kb0 + (wave + u * kWaves) * kStride + lane);
                u32x4 R[kFPL];
                bool extent_ok[kFPL];
                uint64_t gfs[kFPL], gfe[kFPL];
#pragma unroll
                for (int j = 0; j < kFPL; j++) {
                    const uint32_t f = f0 + j * kThreads + tid;
                    const bool in = f < B.n_frag;
                    // the CSR offsets (host batches were checked by irf_submit_batch; device-resident ones are checked here): inside
                    // the array's bounds, non-decreasing, spanning [0, n_blk].  Clamped, they address nothing out of bounds.
                    if (in && (o0[j] > B.n_blk || o1[j] > B.n_blk || o1[j] < o0[j] || (f == 0u && o0[j] != 0u) || (f == B.n_frag - 1u && o1[j] != B.n_blk))) err |= ERR_CSR;
                    uint32_t o0c = o0[j] < B.n_blk ? o0[j] : B.n_blk, o1c = o1[j] < B.n_blk ? o1[j] : B.n_blk;
                    if (!in) { o0c = 0; o1c = 0; }
                    scatter(kb0, j, o0c, o1c);
                    const bool chr_ok = in && chr[j] < n_chr;
                    if (in && !chr_ok) err |= ERR_CHR_RANGE;
                    u32x4 rec = S.chrtab[chr_ok && chr[j] < kChrTab ? chr[j] : 0u];
                    if (chr_ok && chr[j] >= kChrTab) rec = chr_rec[chr[j]];
                    bool counted = chr_ok && rec.w != 0u;
                    bool in_chr_count = counted;
                    if (counted && rec.w == 2u) {
                        // a reference this context owns in part (irf_chr_map_update_ranges): a fragment that does not reach
                        // into [x, y) is ignored; one that does is counted whole (it may add to owned features on this side
                        // of the cut), but FragmentsInChr and FragmentsInROI belong to the context that owns its start
                        const u32x2 own = ((const AS1 u32x2*)T.chr_own)[chr[j]];
                        counted = fs[j] < own.y && fe[j] > own.x;
                        in_chr_count = counted && fs[j] >= own.x;
                    }
                    const uint64_t base = (uint64_t)rec.y << 32 | rec.x;
                    const uint32_t clen = rec.z;
                    const uint32_t fdir = flg[j] & 1u;
                    roi_row[j] = fdir << 31;
                    S.facts[j * kThreads + tid] = counted ? make_uint2((uint32_t)(base >> 6) | (fdir ? kFragDir : 0u) | kFragValid, clen) : make_uint2(0u, 0u);
                    // FragmentsInROI: first ROI (by end) whose end >= fragment end; counted if it contains the fragment.  The
                    // record is requested here and looked at below
                    extent_ok[j] = n_roi_dev != 0u && in_chr_count && fs[j] <= fe[j] && fe[j] <= clen;       // (FragmentsInROI, like FragmentsInChr, belongs to the owner of the fragment's start)
                    gfs[j] = base + fs[j]; gfe[j] = base + fe[j];
                    R[j] = roi_bkt[extent_ok[j] ? gfe[j] >> kBucketShift : 0ull];
                    // FragmentsInChr: one LDS atomic per distinct (chromosome, strand) of the wave
                    const uint32_t k = in_chr_count ? chr[j] * 2u + fdir : kEmpty;
                    uint64_t todo = __ballot(k != kEmpty);
                    while (todo) {
                        const int lead = __ffsll((long long)todo) - 1;
                        const uint32_t lk = (uint32_t)__builtin_amdgcn_readlane((int)k, lead);
                        const uint64_t same = __ballot(k == lk);
                        if ((int)lane == lead) {
                            const uint32_t n = (uint32_t)__popcll(same);
                            if (lk < kChrBins) atomicAdd(&S.chr[lk], n);
                            else add_u32(T.chr_cnt + lk, n);        // (a reference beyond the LDS histogram: this wave pays for the atomic)
                        }
                        todo &= ~same;
                    }
                }
#pragma unroll
                for (int j = 0; j < kFPL; j++) {
                    const uint32_t fdir31 = roi_row[j];
                    roi_row[j] = kEmpty;
                    uint32_t q = R[j].z & 0x0fffffffu;
                    if (extent_ok[j] && q < n_roi_dev) {
                        uint64_t rgs = (uint64_t)((R[j].z >> 28) & 3u) << 32 | R[j].x;
                        const uint64_t rge = (uint64_t)(R[j].z >> 30) << 32 | R[j].y;
                        uint32_t row = R[j].w;
                        bool have = true;
                        if (rge < gfe[j]) {           // several ROIs end inside the bucket before the fragment does
                            q++;
                            while (q < n_roi_dev && gp(T.roi_gend)[q] < gfe[j]) q++;
                            have = q < n_roi_dev;
                            if (have) { rgs = gp(T.roi_gstart)[q]; row = gp(T.roi_idx)[q]; }
                        }
                        if (have && rgs <= gfs[j]) roi_row[j] = row | fdir31;
                    }
                }
            }
            CPROF(1);
            // ---------------- the flush: every load above has landed, nothing is waited for until the barrier ----------------
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef K_STATIC_SLICES
            if (tid == 0 && i == 0u) S.next_slice = queued;         // (read when the slice is over: many barriers from here)
#endif
            flush_window(T, S, win_old, flush_to, lane, wave, s_cstat_local);
            for (uint32_t e = tid; e < n_oq; e += kThreads) oq_apply(T, S.oq[(i + 1u) & 1u][e]);
            CSTAT(3, wave == 0u ? n_oq : 0u); CSTAT(6, wave == 0u ? 1u : 0u);
            if (do_caches) {
                CSTAT(5, wave == 0u ? 1u : 0u);
                const uint32_t n_l = S.ll_n, n_j = S.jl_n, n_p = S.pl_n, n_r = S.rl_n;
                flush_lines(T, S, n_l, lane, wave, s_cstat_local);
                flush_cache<kJCache>(S.jtag, S.jcnt, S.jlist, n_j, T.known_cnt, T.n_kj, tid);
                flush_cache<kPCache>(S.ptag, S.pcnt, S.plist, n_p, T.sp_cnt, T.n_sp, tid);
                flush_cache<kRCache>(S.rtag, S.rcnt, S.rlist, n_r, T.roi_cnt, T.n_roi, tid);
                if (wave == kWaves - 1u) flush_novel(T, S, lane, err);
                wg_barrier();           // (every wave has read the counts of claimed slots)
                if (tid == 0) { S.ll_n = 0; S.jl_n = 0; S.pl_n = 0; S.rl_n = 0; }
            }
            CPROF(2);
            if (last) break;            // (the slice's accumulators are empty again)
#ifndef K_NO_PREFETCH
            if (i + 1u < nc) fc_next = load_frag(c0 + i + 1u);      // (in flight under the block phase)
#endif
            wg_barrier();               // ---- facts and owners of chunk i are in place, the flushed cells are zero ----
            CPROF(4);

            // ---------------- B. block phase: one block per lane and stride of 63, kBU strides at once ----------------
#pragma unroll
            for (int j = 0; j < kFPL; j++)
                if (roi_row[j] != kEmpty) cache_add<kRCache>(T, S, S.rtag, S.rcnt, S.rlist, &S.rl_n, roi_row[j] & 0x7fffffffu, roi_row[j] >> 31, kOqRoi);
            const bool use_win = kb1 - kb0 <= kWinMaxBlocks;
            if (use_win) ev_since += 2u * (kb1 - kb0);
            const uint32_t win5 = (uint32_t)(win >> 5);
            uint32_t wmax = 0;
            for (uint32_t t0 = kb0; t0 < kb1;) {
                const uint32_t tile_n = kb1 - t0 < kTile ? kb1 - t0 : kTile;
                const uint32_t n_str = (tile_n + kStride - 1u) / kStride;
                // (The first stride's columns were requested in the fragment phase and have long arrived; the compiler does not
                //  know — the wait behind the fragment phase is inline assembly — and, merging "still in flight" from the loop's
                //  entry with the back edge, it waited for the NEXT stride's columns at the top of every iteration, a full
                //  round trip in front of the table loads.  A use here makes it place its own wait in front of the loop.)
#pragma unroll
                for (int u = 0; u < kBU; u++) asm volatile("" : "+v"(nxt[u].bs), "+v"(nxt[u].bl), "+v"(nxt[u].rd));
                for (uint32_t s0 = wave; s0 < n_str; s0 += kWaves * kBU) {
                    // (a stride beyond the tile's last does nothing: `here` is false in every lane)
                    Cols cur[kBU];
                    uint32_t ti[kBU], ow[kBU], fb[kBU], clen_b[kBU], be[kBU], bdir[kBU];
                    bool here[kBU], own[kBU], ok[kBU];
                    u32x4 rs[kBU], re[kBU], je[kBU];
#pragma unroll
                    for (int u = 0; u < kBU; u++) {
                        ti[u] = (s0 + u * kWaves) * kStride + lane;
                        cur[u] = nxt[u];
                        nxt[u] = load_cols(t0 + ti[u] + kWaves * kBU * kStride);        // (the next pass, requested now)
                        here[u] = t0 + ti[u] < kb1 && ti[u] <= kTile;        // the block exists and has an owner entry in this tile
                        own[u] = here[u] && lane < kStride && ti[u] < tile_n;
                        ow[u] = S.owner[here[u] ? ti[u] : 0u];
                    }
#pragma unroll
                    for (int u = 0; u < kBU; u++) {
                        const uint2 fa = S.facts[ow[u] & kOwnMask];
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
                        rs[u] = cell_s[ok[u] ? cbase + (bs >> kCellShift) : 0u];
                        re[u] = cell_e[ok[u] ? cbase + (be[u] >> kCellShift) : 0u];
                    }
                    uint32_t bs1[kBU], span[kBU], klo[kBU], khi[kBU], slot[kBU], rsz1[kBU];
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
                        jc[u] = has_next[u] && rd1 == cur[u].rd && cur[u].bl >= junc_min_flank && bl1 >= junc_min_flank;
                        span[u] = bs1[u] - be[u];
                        if (jc[u] && (bs1[u] < be[u] || span[u] >= (1u << kJuncSpanBits))) { err |= ERR_JUNC_RANGE; jc[u] = false; }
                        gbe[u] = ((uint64_t)(fb[u] & kFragBaseMask) << 6) + be[u];
                        klo[u] = ((uint32_t)gbe[u] << kJuncSpanBits) | span[u];
                        khi[u] = (uint32_t)(gbe[u] >> (32 - kJuncSpanBits));
                        slot[u] = junction_hash(klo[u], khi[u]) & jh_mask;
                        je[u] = jhash[jc[u] ? slot[u] : 0u];
                        asm volatile("" : "+v"(je[u].z));       // (the row comes with the key, not in a second load behind the compare)
                    }
#pragma unroll
                    for (int u = 0; u < kBU; u++) rsz1[u] = from_next_lane(rs[u].z);
#pragma unroll
                    for (int u = 0; u < kBU; u++) {
                        const uint32_t bs = cur[u].bs, bl = cur[u].bl;
                        // known junction?  (the probe continues only past a slot another key holds: load <= 1/32)
                        bool hit = jc[u] && je[u].x == klo[u] && je[u].y == khi[u];
                        {
                            bool more = jc[u] && !hit && (je[u].x & je[u].y) != kEmpty;
                            while (more) {
                                slot[u] = (slot[u] + 1u) & jh_mask;
                                je[u] = jhash[slot[u]];
                                hit = je[u].x == klo[u] && je[u].y == khi[u];
                                more = !hit && (je[u].x & je[u].y) != kEmpty;
                            }
                        }
                        // JunctionCount: a known junction's counter also stands for the read's two difference events here
                        if (hit) cache_add<kJCache>(T, S, S.jtag, S.jcnt, S.jlist, &S.jl_n, je[u].z, bdir[u], kOqKnown);
                        if (jc[u] && !hit) novel_add(T, S, (gbe[u] << kJuncLeftShift) | ((unsigned long long)span[u] << 1) | bdir[u], lane, err);
                        // CoverageBlocks / FragmentsMap: the block's end, its successor's start (unless the junction between
                        // them was counted on a known junction), and its own start if it opens the fragment
                        depth_event(T, S, win5, use_win, re[u].z, be[u] & 31u, bdir[u], -1, own[u] && ok[u] && !hit, lane, wmax);
                        depth_event(T, S, win5, use_win, rsz1[u], bs1[u] & 31u, bdir[u], 1, has_next[u] && !hit, lane, wmax);
                        depth_event(T, S, win5, use_win, rs[u].z, bs & 31u, bdir[u], 1, own[u] && ok[u] && (ow[u] & kOwnFirst) != 0u, lane, wmax);
                        // SpansPoint: the points in [bs + sp_lo_add, be - sp_hi_sub] are n1 - n0 points with consecutive rows
                        if (own[u] && ok[u] && bl > sp_min_len) {
                            const uint32_t skip = (uint32_t)__popc(rs[u].y & ((1u << (bs & 31u)) - 1u));
                            const uint32_t n0 = rs[u].x + skip;
                            const uint32_t n1 = re[u].x + (uint32_t)__popc(re[u].y & ((1u << (be[u] & 31u)) - 1u));
                            const uint32_t row0 = rs[u].w + skip;
                            for (uint32_t p = n0; p < n1; p++) cache_add<kPCache>(T, S, S.ptag, S.pcnt, S.plist, &S.pl_n, row0 + (p - n0), bdir[u], kOqPoint);
                        }
                    }
                }
                t0 += kTile;
                if (t0 < kb1) {         // a chunk with more blocks than one tile holds: the next tile's owners
                    wg_barrier();
#pragma unroll
                    for (int j = 0; j < kFPL; j++) {        // (the offsets once more: not kept in registers for this rare case)
                        const uint32_t f = (c0 + i) * kChunk + j * kThreads + tid;
                        if (f < B.n_frag) {
                            const uint32_t a = B.frag_blk_off[f], b = B.frag_blk_off[f + 1u];
                            scatter(t0, j, a < B.n_blk ? a : B.n_blk, b < B.n_blk ? b : B.n_blk);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kBU; u++) nxt[u] = load_cols(t0 + (wave + u * kWaves) * kStride + lane);
                    wg_barrier();
                }
            }
            wmax = wave_max_dpp(wmax);
            if (lane == 63u && wmax) atomicMax(&S.wmax[i & 1u], wmax);
#ifdef K_NO_PREFETCH
            if (i + 1u < nc) fc_next = load_frag(c0 + i + 1u);
#endif
        }
    }
#ifdef COUNT_PROF
    if (lane == 0u && wave == 0u) for (int k = 0; k < 8; k++) atomicAdd(&g_cstat[k], (unsigned long long)s_cstat_local[k]);
    if (lane == 0u && prof_on && blockIdx.x < 2048u) for (int k = 0; k < 8; k++) g_cprof[blockIdx.x][wave == 0u ? 0 : 1][k] = prof_acc[k];
#endif
    if (err) __hip_atomic_fetch_or(gp(T.err_flags), err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wg_barrier();
    for (uint32_t k = tid; k < kChrBins; k += kThreads) {
        const uint32_t v = S.chr[k];
        if (v && (k >> 1) < T.n_chr) add_u32(T.chr_cnt + k, v);
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
    wg_barrier();
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
            __hip_atomic_fetch_add(T.seg_sum + (uint64_t)dir * (T.plane_stride >> kSegShift) + (cl >> kSegShift), (uint32_t)-c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(T.seg_sum + (uint64_t)dir * (T.plane_stride >> kSegShift) + (cr >> kSegShift), (uint32_t)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            T.dirty[((cl >> 4) << 1) + dir] = 1;
            T.dirty[((cr >> 4) << 1) + dir] = 1;
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { same += __shfl_xor(same, d); opp += __shfl_xor(opp, d); }
    if ((threadIdx.x & 63u) == 0u) {
        if (same) atomicAdd(&s_votes[0], same);
        if (opp) atomicAdd(&s_votes[1], opp);
    }
    wg_barrier();
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
// points (binary search), the points of its 32 shifted positions as bits, the plane word and the counter row of the first.
__global__ __launch_bounds__(256) void build_cells_kernel(const uint64_t* __restrict__ points, const uint32_t* __restrict__ rows, uint32_t n_points,
                                                          const uint32_t* __restrict__ pmap, uint64_t n_cells, int32_t shift_s, int32_t shift_e,
                                                          uint4* __restrict__ cell_s, uint4* __restrict__ cell_e) {
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
        (t == 0 ? cell_s : cell_e)[c] = make_uint4(a, bits, word, a < n_points ? rows[a] : 0u);
    }
}

}   // namespace

uint32_t count_chunk_fragments() { return kChunk; }

void launch_build_cells(const uint64_t* points, const uint32_t* rows, uint32_t n_points, const uint32_t* pmap, uint64_t n_cells, int32_t shift_s,
                        int32_t shift_e, uint4* cell_s, uint4* cell_e, hipStream_t s) {
    if (!n_cells) return;
    hipLaunchKernelGGL(build_cells_kernel, dim3((uint32_t)((n_cells + 255u) / 256u)), dim3(256), 0, s, points, rows, n_points, pmap, n_cells, shift_s,
                       shift_e, cell_s, cell_e);
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
    //  and must leave room for the copy kernels of the table export and for the next count kernel, which start next to it:
    //  with eight workgroups per CU it is over sooner, but the count kernel's workgroups then wait for its slots, 2.86 ms
    //  against 2.61 ms for the count kernel in the step)
    const uint64_t kClearWgs = 512;
    const uint64_t wgs = (n_groups + 255u) / 256u;
    hipLaunchKernelGGL(sparse_clear_kernel, dim3((uint32_t)(wgs < kClearWgs ? wgs : kClearWgs)), dim3(256), 0, s, planes, dirty, plane_stride, n_groups);
}

void launch_count_prepare(const CountTables& T, CountTables* d_T, CountTables* shadow, const BatchDev& B, hipStream_t s) {
    if (B.n_frag == 0) return;
    if (memcmp(&T, shadow, sizeof T) != 0) {
        hipLaunchKernelGGL(store_tables_kernel, dim3(1), dim3(1), 0, s, T, d_T);
        *shadow = T;
    }
    const uint32_t n_chunks = (B.n_frag + kChunk - 1u) / kChunk;
    hipLaunchKernelGGL(chunk_origin_kernel, dim3((n_chunks + 255u) / 256u), dim3(256), 0, s, T, B, n_chunks);
}

void launch_count(CountTables* d_T, const BatchDev& B, hipStream_t s) {
    if (B.n_frag == 0) return;
    const uint32_t n_chunks = (B.n_frag + kChunk - 1u) / kChunk;
    const uint32_t n_slices = slice_map(n_chunks).n_slices;
    const uint32_t most = 256u * K_WGS_PER_CU;
    const uint32_t blocks = n_slices < most ? n_slices : most;
    hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(kThreads), 0, s, d_T, B);
#ifdef COUNT_PROF
    {
        static uint32_t h[2048][3][8];
        unsigned long long st[8], zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cprof), sizeof h);
        (void)hipMemcpyFromSymbol(st, HIP_SYMBOL(g_cstat), sizeof st);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_cstat), zero, sizeof zero);
        fprintf(stderr, "count_stat: chunks %llu, window steps looked at %llu, not empty %llu, line steps not empty %llu, queue entries %llu, jumps %llu, cache flushes %llu\n",
                st[6], st[0], st[1], st[2], st[3], st[4], st[5]);
        const char* who[2] = {"wave 0", "last wave"};
        for (int w = 0; w < 2; w++) {
            double a[8] = {0};
            for (uint32_t b = 0; b < blocks && b < 2048u; b++) for (int k = 0; k < 8; k++) a[k] += h[b][w][k];
            const double n = blocks < 2048u ? blocks : 2048u;
            fprintf(stderr, "count_prof %s, 10 ns ticks per workgroup: wait at barrier 1 %.0f | fragment phase %.0f | flush %.0f | wait at barrier 2 %.0f | block phase %.0f\n",
                    who[w], a[0] / n, a[1] / n, a[2] / n, a[4] / n, a[3] / n);
        }
    }
#endif
}

}   // namespace dev
}   // namespace irf
