// count.hip — the fused per-batch counting kernel (gfx950).
//
// One pass over a structure-of-arrays fragment batch replaces, for every
// fragment, the ProcessBlocks calls of ⟨upstream⟩ ReadBlockProcessor.cpp
// FragmentsInChr / FragmentsInROI / JunctionCount / SpansPoint and of
// ReadBlockProcessor_CoverageBlocks.cpp CoverageBlocks (+ FragmentsMap):
//   * FragmentsInChr : wave-aggregated LDS histogram per workgroup, flushed with global atomics
//   * FragmentsInROI : bucket index over the end-sorted ROI table + short walk
//   * CoverageBlocks : +1 / -1 into the per-strand difference array (depth comes
//                      from the prefix scan at finalize)
//   * SpansPoint     : bucket table + short walk of the sorted points
//   * JunctionCount  : events on a known junction (ref-sj.ref) bump that junction's
//                      counter (bucket table + walk; LDS-cached per chunk); novel junctions
//                      are appended as packed (chr,left,span,strand) events to a
//                      buffer and sort-reduced at finalize
// All outputs are commutative integer sums, so any batch order gives identical
// results.  Integer work bound by memory latency and by the atomic rate: no MFMA.
//
// Shape of the kernel.  The input is CSR (fragments -> blocks).  A workgroup processes a chunk of kChunk consecutive
// fragments in bulk phases, so that every lane has many independent, wide loads in flight:
//   1. fragment-parallel, 4 consecutive fragments per lane (16-byte loads of the fragment columns): per-fragment
//      facts (chromosome base, strand, chromosome length, CSR offset) go to LDS; FragmentsInChr and
//      FragmentsInROI are done here;
//   2. block-parallel, 4 consecutive blocks per lane (16-byte loads of the block columns): the owning fragment comes
//      from a binary search of the LDS offsets; ONE 16-byte bucket-table entry per block end gives the plane map,
//      the SpansPoint range and the known-junction range.  A block owns the junction to its right, and with it
//      the two difference events next to that junction (its own end, its successor's start): a read on a known
//      junction costs one wave-aggregated LDS increment of that junction's counter, and the junction's two
//      difference events are emitted once per chunk from the counter;
//   3. the LDS accumulators are flushed with global atomics shaped for the memory-side atomic unit (consecutive
//      lanes = consecutive cells, whole 64-byte lines): the difference window (both strands packed in one int32
//      per position), the line-granular cache of far events, the junction cache, the novel-junction queue.
// Loads are unconditional on clamped indices: a conditional load compiles to a branch with a full vmcnt wait
// behind it, which serialises the round trips.
#include "kernels.h"

#include <cstdio>

namespace irf {
namespace dev {

namespace {

#ifdef COUNT_PROF
// diagnostic build only (tools/build_variant.sh -DCOUNT_PROF): wall-clock ticks (10 ns) of wave 0 per phase and workgroup
__device__ uint32_t g_cprof[1024][8];
#define CPROF(k) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); prof_acc[k] += (uint32_t)(now_ - prof_t); prof_t = now_; } } while (0)
#else
#define CPROF(k) do { } while (0)
#endif

#ifndef K_THREADS
#define K_THREADS 256
#endif
// Four workgroups per CU: a chunk is a chain of dependent phases (columns -> LDS facts -> bucket entries -> key / point
// tables -> LDS atomics -> flush), and what a CU gets done is the number of chunks it has in flight (2 workgroups: 5.5 ms,
// 3: 3.03 ms, 4: 2.75 ms).  That takes <= 40 KB of LDS (a 2048-position window instead of 4096: 39.8 KB) and <= 128
// registers (168 wanted: 27 spilled, almost all of them outside the block loop).
#ifndef K_WIN
#define K_WIN 2048
#endif
#ifndef K_LINES
#define K_LINES 128
#endif
#ifndef K_JCACHE
#define K_JCACHE 256
#endif
#ifndef K_NOVELQ
#define K_NOVELQ 256
#endif
#ifndef K_WGS_PER_CU
#define K_WGS_PER_CU 4
#endif
#ifndef K_MIN_WGS
#define K_MIN_WGS 4
#endif
constexpr int kThreads = K_THREADS;
#ifndef K_VEC
#define K_VEC 4
#endif
constexpr int kVec = K_VEC;                           // consecutive fragments / blocks per lane (4: 16-byte loads, 2: 8-byte loads)
static_assert(kVec == 4 || kVec == 2, "K_VEC");
// a lane's kVec consecutive elements of a uint32 / uint8 column in one load
template <int V> struct VecT;
template <> struct VecT<4> { typedef uint4 u32v; typedef uint32_t u8v; };
template <> struct VecT<2> { typedef uint2 u32v; typedef uint16_t u8v; };
typedef VecT<kVec>::u32v U32V;
typedef VecT<kVec>::u8v U8V;
struct __attribute__((aligned(4))) Quad { uint32_t v[4]; };      // four consecutive table entries at any dword
__device__ __forceinline__ void unpack(const uint4& v, uint32_t* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void unpack(const uint2& v, uint32_t* o) { o[0] = v.x; o[1] = v.y; }
constexpr int kVecLog2 = kVec == 4 ? 2 : 1;
constexpr int kChunk = kThreads * kVec;               // fragments per workgroup iteration
constexpr int kChunkLog2 = kChunk == 1024 ? 10 : kChunk == 2048 ? 11 : kChunk == 512 ? 9 : 0;
static_assert(kChunkLog2 > 0, "K_THREADS");
constexpr int kChrBins = 512;       // LDS histogram covers refIDs < 256; larger ones go straight to global
constexpr uint32_t kJCache = K_JCACHE;   // LDS cache of known-junction counters (direct-mapped on the known-key index)
constexpr uint32_t kEmpty = 0xffffffffu;
constexpr uint32_t kLines = K_LINES;     // LDS cache of 16-position lines of the difference planes outside the window
constexpr uint32_t kWin = K_WIN;         // positions of the LDS difference-array window (both strands in one int32)
#ifndef K_PCACHE
#define K_PCACHE 128
#endif
constexpr uint32_t kPCache = K_PCACHE;         // LDS cache of SpansPoint counters (direct-mapped on the point index)
constexpr uint32_t kNovelQ = K_NOVELQ;   // LDS queue of novel-junction events per chunk (those the table below has no room for)
constexpr uint32_t kNovelH = 128;        // LDS table that merges a chunk's novel-junction events by key (open addressing, 4 probes)
constexpr unsigned long long kEmpty64 = ~0ull;     // (no event has bit 63 set)
constexpr uint32_t kWinMaxBlocks = 16000;   // more blocks than this in one chunk: a packed 16-bit counter could overflow
constexpr uint32_t kFragValid = 1u << 31, kFragDir = 1u << 30, kFragBaseMask = (1u << 30) - 1u;   // base is a multiple of 64

struct Lds {
    uint32_t off[kChunk + 4];          // CSR offsets of the chunk's fragments (clamped to n_blk); [kChunk] = end
    uint32_t fb[kChunk];               // chr_base >> 6 | strand << 30 | counted << 31
    uint32_t fclen[kChunk];            // chromosome length
    int32_t tile[kWin + 64];           // difference window: low half strand 0, high half strand 1 (+ one idle cell per lane)
    uint32_t ltag[kLines];             // far-event line cache: plane position >> 4
    int32_t lcnt[kLines + 1][16];      // both strands of a position in one word, as in the window (low half strand 0); line kLines: idle
    uint32_t jtag[kJCache];            // known-junction cache: index into known_keys
    uint32_t jcnt[2][kJCache];
    uint32_t ptag[kPCache];            // SpansPoint cache: index into sp_pos32
    uint32_t pcnt[2][kPCache];
    unsigned long long nq[kNovelQ];
    unsigned long long nh_key[kNovelH];
    uint32_t nh_cnt[kNovelH];
    uint32_t chr[kChrBins];
    unsigned long long win, nq_base;
    uint32_t nq_n, nh_n, wmax;
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

// The sum of every 256-position segment of the planes is kept next to them (seg_sum; its exclusive scan is the depth
// carry of stats.hip), so that finalize does not have to read the planes once more just to add them up.
__device__ __forceinline__ void seg_add(const CountTables& T, uint32_t dir, uint64_t c, int v) {
    __hip_atomic_fetch_add(T.seg_sum + (uint64_t)dir * (T.plane_stride >> kSegShift) + (c >> kSegShift), (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (CountTables.dirty: the line of plane position c of strand dir now holds something)
__device__ __forceinline__ void mark_dirty(const CountTables& T, uint32_t dir, uint64_t c) { T.dirty[((c >> 4) << 1) + dir] = 1; }

__device__ __forceinline__ uint64_t plane_of(const CountTables& T, uint32_t m, uint64_t g) {
    if (!T.pmap) return g;
    const uint64_t c = (uint64_t)(m >> 1) << kBucketShift;
    return (m & 1u) ? c + (g & ((1u << kBucketShift) - 1u)) : c;
}

// A difference event outside the chunk's window: the line cache, else (line slot taken by another line) a global atomic.
__device__ __forceinline__ void far_event(const CountTables& T, Lds& S, bool use_win, uint64_t c, uint32_t dir, int v) {
#ifdef K_NOLINES
    __hip_atomic_fetch_add(T.diff + (uint64_t)dir * T.plane_stride + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    seg_add(T, dir, c, v);
    mark_dirty(T, dir, c);
    return;
#endif
    // (a line holds both strands, packed like the window: unstranded reads put both on the same positions, so this is
    //  twice the lines for the same LDS; the packed halves cannot overflow while use_win holds)
    // (one compare-and-swap claims an empty slot or returns the line that holds it: no read + branch in front of it; the
    //  counter update is predicated like the window's — line kLines is nobody's — so only a conflict is a branch)
    const uint32_t want = (uint32_t)(c >> 4);
    const uint32_t slot = want & (kLines - 1u);
    const uint32_t old = atomicCAS(&S.ltag[slot], kEmpty, want);
    const bool mine = use_win && (old == kEmpty || old == want);
    atomicAdd(&S.lcnt[mine ? slot : kLines][c & 15u], mine ? (dir ? v * 65536 : v) : 0);
    if (!mine) {
        __hip_atomic_fetch_add(T.diff + (uint64_t)dir * T.plane_stride + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        seg_add(T, dir, c, v);
        mark_dirty(T, dir, c);
    }
}

// One difference event at plane position c, with a predicate instead of a branch around the call: the chunk's LDS window if
// it is inside, else far_event.  The window update is executed by every lane (lanes
// without an event, or with a far one, add 0 to a cell of their own behind the window), so that only the far path is a
// branch — the kernel is bound by instruction issue, and the exec-mask bookkeeping of a branch costs as much as the update.
__device__ __forceinline__ void depth_event_p(const CountTables& T, Lds& S, unsigned long long win, bool use_win, uint64_t c,
                                              uint32_t dir, int v, bool pred, uint32_t& wmax) {
    const uint64_t o = c - win;
    const bool near = pred && use_win && o < kWin;
    const uint32_t idx = near ? (uint32_t)o : kWin + (threadIdx.x & 63u);
    atomicAdd(&S.tile[idx], near ? (dir ? v * 65536 : v) : 0);
    const uint32_t om = near ? (uint32_t)o : 0u;
    wmax = om > wmax ? om : wmax;
    if (pred && !near) far_event(T, S, use_win, c, dir, v);
}

// n reads of strand dir on known junction idx: the chunk's LDS counter, or (slot taken by another junction) straight to
// the table.  The counter also stands for the reads' two difference events next to the junction: they are applied once
// per junction and sample by known_junction_depth_kernel at finalize, not once per read (or per chunk) here.
__device__ __forceinline__ void junction_add(const CountTables& T, Lds& S, uint32_t idx, uint32_t dir, uint32_t n) {
    const uint32_t slot = idx & (kJCache - 1u);
    uint32_t tag = S.jtag[slot];
    if (tag == kEmpty) {
        const uint32_t old = atomicCAS(&S.jtag[slot], kEmpty, idx);
        tag = old == kEmpty ? idx : old;
    }
    if (tag == idx) atomicAdd(&S.jcnt[dir][slot], n);
    else atomicAdd(&T.known_cnt[(uint64_t)dir * T.n_kj + T.known_kjidx[idx]], n);
}

__global__ __launch_bounds__(kThreads, K_MIN_WGS) void count_kernel(CountTables T, BatchDev B) {
    __shared__ __align__(16) Lds S;
    const uint32_t tid = threadIdx.x;
    for (int k = tid; k < kChrBins; k += kThreads) S.chr[k] = 0;
    for (int k = tid; k < (int)kWin + 64; k += kThreads) S.tile[k] = 0;
    for (int k = tid; k < (int)kLines; k += kThreads) S.ltag[k] = kEmpty;
    for (int k = tid; k < (int)(kLines + 1) * 16; k += kThreads) S.lcnt[k >> 4][k & 15] = 0;
    for (int k = tid; k < (int)kJCache; k += kThreads) { S.jtag[k] = kEmpty; S.jcnt[0][k] = 0; S.jcnt[1][k] = 0; }
    for (int k = tid; k < (int)kPCache; k += kThreads) { S.ptag[k] = kEmpty; S.pcnt[0][k] = 0; S.pcnt[1][k] = 0; }
    for (int k = tid; k < (int)kNovelH; k += kThreads) { S.nh_key[k] = kEmpty64; S.nh_cnt[k] = 0; }
    if (tid == 0) { S.nq_n = 0; S.nh_n = 0; S.wmax = 0; }
    __syncthreads();

    const uint32_t lane = tid & 63u;
    uint32_t err = 0;
#ifdef COUNT_PROF
    unsigned long long prof_t = wall_clock64();
    uint32_t prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const uint32_t n_chunks = (B.n_frag + kChunk - 1) / kChunk;
    const U32V* __restrict__ chr4 = reinterpret_cast<const U32V*>(B.frag_chr);
    const U32V* __restrict__ fs4 = reinterpret_cast<const U32V*>(B.frag_start);
    const U32V* __restrict__ fe4 = reinterpret_cast<const U32V*>(B.frag_end);
    const U32V* __restrict__ off4 = reinterpret_cast<const U32V*>(B.frag_blk_off);
    const U8V* __restrict__ flg4 = reinterpret_cast<const U8V*>(B.frag_flags);
    const U32V* __restrict__ bs4 = reinterpret_cast<const U32V*>(B.blk_start);
    const U32V* __restrict__ bl4 = reinterpret_cast<const U32V*>(B.blk_len);
    const U8V* __restrict__ rd4 = reinterpret_cast<const U8V*>(B.blk_read);
    // XCD-aware chunk order: workgroup b runs on XCD b % 8 (round-robin dispatch); every XCD gets a contiguous
    // eighth of the (coordinate-sorted) batch, so that neighbouring chunks — same table lines, overlapping
    // difference windows — share one L2
    const uint32_t xcd = blockIdx.x & 7u, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const uint32_t per_xcd = (n_chunks + 7u) / 8u;
    const uint32_t c_end = (xcd + 1u) * per_xcd < n_chunks ? (xcd + 1u) * per_xcd : n_chunks;
#ifdef K_INTERLEAVE
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
#else
    for (uint32_t c = xcd * per_xcd + wg_in_xcd; c < c_end; c += wgs_per_xcd) {
#endif
        const uint32_t f0 = c * kChunk;
        // ---------------- 1. fragment-parallel: fragments f0 + 4 tid .. + 3 ----------------
        {
            // a 16-byte group that starts inside an array never leaves the array's last page: lanes past the end read
            // the last group again and discard it
            const uint32_t g_last = (B.n_frag - 1u) >> kVecLog2;
            const uint32_t g = (f0 >> kVecLog2) + tid < g_last ? (f0 >> kVecLog2) + tid : g_last;
            const U32V vchr = chr4[g], vfs = fs4[g], vfe = fe4[g], voff = off4[g];
            const uint32_t vflg = flg4[g];
            if (tid == 0) {
                const uint32_t last = f0 + kChunk < B.n_frag ? f0 + kChunk : B.n_frag;
                const uint32_t o = B.frag_blk_off[last];
                S.off[kChunk] = o < B.n_blk ? o : B.n_blk;
            }
            uint32_t chr[kVec], fs[kVec], fe[kVec], off[kVec];
            unpack(vchr, chr); unpack(vfs, fs); unpack(vfe, fe); unpack(voff, off);
            uint32_t key[kVec], clen[kVec];
            uint64_t base[kVec];
            bool in[kVec];
#pragma unroll
            for (int j = 0; j < kVec; j++) {
                const uint32_t i = f0 + tid * kVec + j;
                in[j] = i < B.n_frag;
                const bool ok = in[j] && chr[j] < T.n_chr;
                if (in[j] && !ok) err |= ERR_CHR_RANGE;
                // the CSR offsets (host batches were checked by irf_submit_batch; device-resident ones are checked here):
                // inside the array's bounds, non-decreasing inside the lane's group (across groups: after the barrier)
                if (in[j] && (off[j] > B.n_blk || (j > 0 && off[j] < off[j - 1]) || (i == 0u && off[j] != 0u))) err |= ERR_CSR;
                const uint32_t cc = ok ? chr[j] : 0u;            // the tables hold at least one entry
                const bool counted = (T.chr_owned[cc] != 0) && ok;
                base[j] = T.chr_base[cc];
                clen[j] = T.chr_len[cc];
                if (!counted) { base[j] = 0ull; clen[j] = 0u; }
                const uint32_t dir = (vflg >> (8 * j)) & 1u;
                const uint32_t li = tid * kVec + j;
                S.off[li] = in[j] ? (off[j] < B.n_blk ? off[j] : B.n_blk) : B.n_blk;
                S.fb[li] = counted ? ((uint32_t)(base[j] >> 6) | (dir ? kFragDir : 0u) | kFragValid) : 0u;
                S.fclen[li] = clen[j];
                key[j] = counted ? chr[j] * 2u + dir : kEmpty;
            }
            if (tid == 0) {
                // the window starts at the chunk's first fragment (coordinate-sorted input: the leftmost one)
                unsigned long long w = 0;
                if (key[0] != kEmpty && fs[0] <= clen[0]) {
                    const uint64_t gp = base[0] + fs[0];
                    w = plane_of(T, T.pmap ? T.pmap[gp >> kBucketShift] : 0u, gp) & ~63ull;
                }
                S.win = w;
            }
            // FragmentsInROI: first ROI (by end) whose end >= fragment end; counted if it contains the fragment
            if (!(T.ablate & 4u) && T.n_roi_dev) {
                uint32_t r[kVec];
                bool extent_ok[kVec];
#pragma unroll
                for (int j = 0; j < kVec; j++) {
                    extent_ok[j] = key[j] != kEmpty && fs[j] <= fe[j] && fe[j] <= clen[j];
                    r[j] = T.roi_bucket[extent_ok[j] ? (base[j] + fe[j]) >> kBucketShift : 0ull];
                }
                uint64_t rge[kVec], rgs[kVec];
#pragma unroll
                for (int j = 0; j < kVec; j++) {
                    if (r[j] >= T.n_roi_dev) { extent_ok[j] = false; r[j] = 0u; }
                    rge[j] = T.roi_gend[r[j]];
                    rgs[j] = T.roi_gstart[r[j]];
                }
#pragma unroll
                for (int j = 0; j < kVec; j++) {
                    if (!extent_ok[j]) continue;
                    const uint64_t gfs = base[j] + fs[j], gfe = base[j] + fe[j];
                    uint32_t q = r[j];
                    if (rge[j] < gfe) {          // several ROIs end inside the bucket before the fragment does
                        q++;
                        while (q < T.n_roi_dev && T.roi_gend[q] < gfe) q++;
                        if (q >= T.n_roi_dev) continue;
                        rgs[j] = T.roi_gstart[q];
                    }
                    if (rgs[j] <= gfs) atomicAdd(&T.roi_cnt[(uint64_t)(key[j] & 1u) * T.n_roi + T.roi_idx[q]], 1u);
                }
            }
            // FragmentsInChr: one LDS atomic per distinct (chromosome, strand) of the wave
            if (!(T.ablate & 16u)) {
#pragma unroll
                for (int j = 0; j < kVec; j++) {
                    const uint32_t k = key[j];
                    uint64_t todo = __ballot(k != kEmpty);
                    while (todo) {
                        const int lead = __ffsll((long long)todo) - 1;
                        const uint32_t lk = (uint32_t)__builtin_amdgcn_readlane((int)k, lead);
                        const uint64_t same = __ballot(k == lk);
                        if ((int)lane == lead) {
                            const uint32_t n = (uint32_t)__popcll(same);
                            if (lk < (uint32_t)kChrBins) atomicAdd(&S.chr[lk], n);
                            else atomicAdd(&T.chr_cnt[lk], n);
                        }
                        todo &= ~same;
                    }
                }
            }
        }
        __syncthreads();
        CPROF(0);

        // ---------------- 2. block-parallel: 4 consecutive blocks per lane ----------------
        const uint32_t kb0 = S.off[0], kb1 = S.off[kChunk];
        // (S.off[kChunk] is the offset behind the chunk's last fragment: the next group's first, or frag_blk_off[n_frag])
        if (S.off[tid * kVec + kVec - 1] > S.off[tid * kVec + kVec] && f0 + tid * kVec + kVec - 1 < B.n_frag) err |= ERR_CSR;
        if (tid == 0 && f0 + kChunk >= B.n_frag && B.frag_blk_off[B.n_frag] != B.n_blk) err |= ERR_CSR;
        const unsigned long long win = S.win;
        const bool use_win = kb1 - kb0 <= kWinMaxBlocks;
        uint32_t wmax = 0;
        const uint32_t g_lastb = B.n_blk ? (B.n_blk - 1u) >> kVecLog2 : 0u;
        // The block columns of an iteration are loaded one iteration ahead (the loads of the first iteration are issued
        // here, before anything waits): a round trip less on the chain columns -> bucket entries -> key / point tables.
        // Every lane also loads the block after its four itself (the junction to the right of its last block).
        struct BlkCols { U32V vbs, vbl; uint32_t vrd, nbs, nbl, nrd; };
        auto load_cols = [&](uint32_t k0) -> BlkCols {
            BlkCols c;
            const uint32_t g = (k0 >> kVecLog2) < g_lastb ? (k0 >> kVecLog2) : g_lastb;
            const uint32_t kn = k0 + kVec < B.n_blk ? k0 + kVec : (B.n_blk ? B.n_blk - 1u : 0u);
            c.vbs = bs4[g]; c.vbl = bl4[g]; c.vrd = rd4[g];
            c.nbs = B.blk_start[kn]; c.nbl = B.blk_len[kn]; c.nrd = B.blk_read[kn];
            return c;
        };
        BlkCols nxt = load_cols((kb0 & ~(uint32_t)(kVec - 1)) + tid * kVec);
        for (uint32_t ka = kb0 & ~(uint32_t)(kVec - 1); ka < kb1 && !(T.ablate & 32u); ka += kThreads * kVec) {
            const uint32_t k0 = ka + tid * kVec;
            const BlkCols cur = nxt;
            nxt = load_cols(k0 + kThreads * kVec);
            uint32_t bs[kVec + 1], bl[kVec + 1], rd[kVec + 1];
            unpack(cur.vbs, bs); bs[kVec] = cur.nbs;
            unpack(cur.vbl, bl); bl[kVec] = cur.nbl;
#pragma unroll
            for (int j = 0; j < kVec; j++) rd[j] = (cur.vrd >> (8 * j)) & 0xffu;
            rd[kVec] = cur.nrd;
            // owner = the last fragment of the chunk whose offset is <= k (empty fragments share an offset with
            // their successor and are never picked): binary search for the first block, a short walk for the others
            uint32_t fbv[kVec], clen[kVec];
            bool act[kVec], first[kVec], next_same[kVec];
            {
                uint32_t lo = 0, hi = kChunk;
                const uint32_t kq = k0 < kb0 ? kb0 : k0;
#pragma unroll
                for (int s = 0; s < kChunkLog2; s++) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (S.off[mid] <= kq) lo = mid; else hi = mid;
                }
                uint32_t f = lo;
#pragma unroll
                for (int j = 0; j < kVec; j++) {
                    const uint32_t k = k0 + j;
                    act[j] = k >= kb0 && k < kb1;
                    if (act[j]) { while (S.off[f + 1] <= k) f++; }        // off[kChunk] = kb1 > k ends the walk
                    fbv[j] = act[j] ? S.fb[f] : 0u;
                    clen[j] = S.fclen[f];
                    first[j] = k == S.off[f];
                    next_same[j] = k + 1u < S.off[f + 1];
                }
            }
            // per block: validity, full coordinates, bucket-table entries of the start and of the end
            bool ok[kVec + 1];
            uint64_t gbs[kVec + 1], gbe[kVec];
            uint2 es[kVec + 1];
            uint4 ee[kVec];
#pragma unroll
            for (int j = 0; j < kVec; j++) {
                const uint32_t be = bs[j] + bl[j];
                const bool valid = (fbv[j] & kFragValid) != 0u;
                const bool in_range = be >= bs[j] && be <= clen[j];
                if (valid && !in_range) err |= ERR_BLOCK_RANGE;
                ok[j] = valid && in_range;
                const uint64_t base = (uint64_t)(fbv[j] & kFragBaseMask) << 6;
                gbs[j] = base + bs[j];
                gbe[j] = base + be;
            }
            {   // the successor of the lane's last block: same fragment (next_same[3]) or unused
                const uint32_t be = bs[kVec] + bl[kVec];
                ok[kVec] = ok[kVec - 1] && next_same[kVec - 1] && be >= bs[kVec] && be <= clen[kVec - 1];
                gbs[kVec] = ((uint64_t)(fbv[kVec - 1] & kFragBaseMask) << 6) + bs[kVec];
            }
#pragma unroll
            for (int j = 0; j <= kVec; j++) {
                const uint4* e = T.bkt + (ok[j] ? gbs[j] >> kBucketShift : 0ull);
                es[j] = *reinterpret_cast<const uint2*>(e);
            }
#pragma unroll
            for (int j = 0; j < kVec; j++) ee[j] = T.bkt[ok[j] ? gbe[j] >> kBucketShift : 0ull];
            CPROF(1);
            // junction to the right of the block: both flanks long enough, same read, same fragment
            bool jc[kVec];
            uint64_t jkey[kVec], kfirst[kVec];
#pragma unroll
            for (int j = 0; j < kVec; j++) {
                jc[j] = ok[j] && ok[j + 1] && next_same[j] && rd[j + 1] == rd[j] && bl[j] >= T.junc_min_flank && bl[j + 1] >= T.junc_min_flank &&
                        !(T.ablate & 8u);
                const uint32_t be = bs[j] + bl[j];
                const uint32_t span = bs[j + 1] - be;
                if (jc[j] && (bs[j + 1] < be || span >= (1u << kJuncSpanBits))) { err |= ERR_JUNC_RANGE; jc[j] = false; }
                jkey[j] = (gbe[j] << kJuncSpanBits) | span;
                const bool kq = jc[j] && (ee[j].w >> 16);
                kfirst[j] = ~0ull;
                if (kq) kfirst[j] = T.known_keys[ee[j].z];                    // (only the lanes with a candidate: see the SpansPoint load)
            }
            // SpansPoint: the first four points under every block are requested now, together with the junction keys
            // (one round trip for all four blocks of the lane; points are compared modulo 2^32)
            uint32_t sp_p[kVec], sp_p1[kVec], sp_g[kVec][4];
            {
#pragma unroll
                for (int j = 0; j < kVec; j++) {
                    const bool on = ok[j] && bl[j] > T.sp_min_len && !(T.ablate & 2u);
                    sp_p[j] = es[j].y;
                    sp_p1[j] = on ? ee[j].y + (ee[j].w & 0xffffu) : sp_p[j];      // saturated count: a lower bound, fixed below
                    // ONE 16-byte load (dword-aligned; the table has four spare entries behind its end) and only by the lanes
                    // that have candidates: the vector L1 handles about one lane's access per clock whatever its width,
                    // and this kernel keeps it busy ~90 % of the time (profiles/r02q_count_pmc.txt)
                    Quad q4 = {{0u, 0u, 0u, 0u}};
                    if (sp_p[j] < sp_p1[j]) q4 = *reinterpret_cast<const Quad*>(T.sp_pos32 + sp_p[j]);
#pragma unroll
                    for (int q = 0; q < 4; q++) sp_g[j][q] = q4.v[q];
                }
            }
            // known junction?  (index into known_keys, or kEmpty)
            uint32_t hit[kVec];
#pragma unroll
            for (int j = 0; j < kVec; j++) {
                hit[j] = kEmpty;
                const uint32_t kn = ee[j].w >> 16;
                if (jc[j] && kn) {
                    const uint32_t i0 = ee[j].z;
                    if (kfirst[j] >= jkey[j]) {
                        if (kfirst[j] == jkey[j]) hit[j] = i0;
                    } else {
                        uint32_t lo = i0 + 1u;
                        uint32_t hi = kn == 0xffffu ? T.bkt[(gbe[j] >> kBucketShift) + 1u].z : i0 + kn;
                        while (hi - lo > 4u) {        // many junctions share this donor bucket: bisect first
                            const uint32_t mid = lo + ((hi - lo) >> 1);
                            if (T.known_keys[mid] < jkey[j]) lo = mid + 1u; else hi = mid + 1u;
                        }
                        for (; lo < hi; lo++) {
                            const uint64_t kq = T.known_keys[lo];
                            if (kq >= jkey[j]) { if (kq == jkey[j]) hit[j] = lo; break; }
                        }
                    }
                }
            }
            CPROF(2);
            // JunctionCount on known junctions: the junction's LDS counter (same-address LDS atomics of a wave are
            // serialised by the hardware at about a lane per clock: cheaper than electing leaders in software).
            // folded = the junction's counter also stands for this read's two difference events (applied at finalize)
            bool folded[kVec];
#pragma unroll
            for (int j = 0; j < kVec; j++) {
                folded[j] = hit[j] != kEmpty && !(T.ablate & 64u);
                if (hit[j] != kEmpty) junction_add(T, S, hit[j], (fbv[j] & kFragDir) ? 1u : 0u, 1u);
            }
            CPROF(3);
            // difference events, SpansPoint, novel junctions
            if (!(T.ablate & 1u)) {
                // CoverageBlocks / FragmentsMap.  Every block end is one event site (unless a known junction's counter
                // stands for it), every block start is one: of the lane's own first block only when it opens a fragment,
                // of the others (the block after the lane's last included) when they open a fragment or follow a
                // predecessor whose junction was not folded.  9 sites for 4 blocks, each with a predicate, not a branch.
#pragma unroll
                for (int j = 0; j < kVec; j++)
                    depth_event_p(T, S, win, use_win, plane_of(T, ee[j].x, gbe[j]), (fbv[j] & kFragDir) ? 1u : 0u, -1, ok[j] && !folded[j], wmax);
                depth_event_p(T, S, win, use_win, plane_of(T, es[0].x, gbs[0]), (fbv[0] & kFragDir) ? 1u : 0u, 1, ok[0] && first[0], wmax);
#pragma unroll
                for (int k = 1; k <= kVec; k++) {
                    const bool opens = k < kVec && ok[k < kVec ? k : 0] && first[k < kVec ? k : 0];
                    const bool follows = ok[k - 1] && !folded[k - 1] && next_same[k - 1] && ok[k];
                    const uint32_t f = opens ? fbv[k < kVec ? k : 0] : fbv[k - 1];
                    depth_event_p(T, S, win, use_win, plane_of(T, es[k].x, gbs[k]), (f & kFragDir) ? 1u : 0u, 1, opens || follows, wmax);
                }
            }
#pragma unroll
            for (int j = 0; j < kVec; j++) {
                if (!ok[j]) continue;
                const uint32_t dir = (fbv[j] & kFragDir) ? 1u : 0u;
                // SpansPoint: the points of the buckets under the block, filtered to [lo, hi].  The first four were
                // requested above (they usually are the block's own exon boundaries, just outside [lo, hi], and the
                // few points inside); more only for a block over a dense stretch
                if (sp_p[j] < sp_p1[j]) {
                    uint32_t p = sp_p[j];
                    const uint32_t pn = ee[j].w & 0xffffu;
                    const uint32_t p1 = pn == 0xffffu ? T.bkt[(gbe[j] >> kBucketShift) + 1u].y : sp_p1[j];
                    const uint32_t lo = (uint32_t)gbs[j] + T.sp_lo_add, hi = (uint32_t)gbe[j] - T.sp_hi_sub;
                    uint32_t* __restrict__ sp_cnt = T.sp_cnt + (uint64_t)dir * T.n_sp;
                    auto point = [&](uint32_t pi) {
                        const uint32_t slot = pi & (kPCache - 1u);
                        uint32_t tag = S.ptag[slot];
                        if (tag == kEmpty) { const uint32_t old = atomicCAS(&S.ptag[slot], kEmpty, pi); tag = old == kEmpty ? pi : old; }
                        if (tag == pi) atomicAdd(&S.pcnt[dir][slot], 1u);
                        else atomicAdd(&sp_cnt[T.sp_row[pi]], 1u);
                    };
                    // which of the four requested candidates lie inside [lo, hi] (the points are sorted: the first one beyond
                    // hi ends the search); the counter updates then run ONCE per qualifying point, not once per candidate
                    bool done = false;
                    uint32_t qual = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (p + q < p1 && !done) {
                            const uint32_t g = sp_g[j][q];
                            if ((int32_t)(g - hi) > 0) done = true;
                            else if ((int32_t)(g - lo) >= 0) qual |= 1u << q;
                        }
                    }
                    while (qual) {
                        const uint32_t q = (uint32_t)__ffs((int)qual) - 1u;
                        qual &= qual - 1u;
                        point(p + q);
                    }
                    for (p += 4u; !done && p < p1; p++) {          // a block over a dense stretch of points (rare)
                        const uint32_t g = T.sp_pos32[p];
                        if ((int32_t)(g - hi) > 0) done = true;
                        else if ((int32_t)(g - lo) >= 0) point(p);
                    }
                }
                if (jc[j] && hit[j] == kEmpty) {
                    // novel junction: a packed event for the sort-reduce at finalize.  A chunk sits on one locus, so its
                    // novel events repeat a handful of keys: they are merged in an LDS table and leave as (key, weight)
                    const uint64_t ev = (gbe[j] << kJuncLeftShift) | ((jkey[j] & ((1ull << kJuncSpanBits) - 1ull)) << 1) | dir;
                    const uint32_t h = (uint32_t)((ev * 0x9E3779B97F4A7C15ull) >> 57);
                    bool merged = false;
#pragma unroll
                    for (uint32_t pr = 0; pr < 4u; pr++) {
                        if (merged) continue;
                        const uint32_t sl = (h + pr) & (kNovelH - 1u);
                        unsigned long long cur = S.nh_key[sl];
                        if (cur == kEmpty64) {
                            const unsigned long long old = atomicCAS(&S.nh_key[sl], kEmpty64, (unsigned long long)ev);
                            cur = old == kEmpty64 ? (unsigned long long)ev : old;
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
            }
            CPROF(7);
        }
        __syncthreads();
        CPROF(4);

        // ---------------- 3. flush the LDS accumulators ----------------
        // 3a. the junction cache and the SpansPoint cache: counters to the tables
        const uint32_t nq = S.nq_n < kNovelQ ? S.nq_n : kNovelQ;
        const uint32_t nh = S.nh_n;
        if (tid == 0 && nq + nh) S.nq_base = atomicAdd(T.junc_count, (unsigned long long)(nq + nh));
        for (uint32_t e = tid; e < kJCache; e += kThreads) {
            const uint32_t idx = S.jtag[e];
            if (idx != kEmpty) {
                const uint32_t c0 = S.jcnt[0][e], c1 = S.jcnt[1][e];
                const uint32_t row = T.known_kjidx[idx];
                if (c0) atomicAdd(&T.known_cnt[row], c0);
                if (c1) atomicAdd(&T.known_cnt[(uint64_t)T.n_kj + row], c1);
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
        if (wmax) atomicMax(&S.wmax, wmax);
        __syncthreads();
        CPROF(5);
        // 3b. line cache and window: consecutive lanes = consecutive positions, so that a wave's atomics fall into
        //     whole 64-byte lines (the memory-side atomic unit works per line)
        const uint32_t wtop = S.wmax;
        static_assert((kLines * 16u) % kThreads == 0u && kWin % kThreads == 0u, "the flush loops run with whole waves (DPP sums)");
        for (uint32_t e = tid; e < kLines * 16u; e += kThreads) {
            const uint32_t slot = e >> 4, cell = e & 15u;               // a 16-lane row = one line = inside one segment
            const uint32_t tag = S.ltag[slot];
            const int32_t v = tag != kEmpty ? S.lcnt[slot][cell] : 0;
            const int32_t v0 = (int32_t)(int16_t)(uint16_t)(v & 0xffff);
            const int32_t v1 = (v - v0) >> 16;
            const uint64_t c = ((uint64_t)tag << 4) + cell;
            if (v) {
                if (v0 && !(T.ablate & 256u)) __hip_atomic_fetch_add(T.diff + c, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v1 && !(T.ablate & 256u)) __hip_atomic_fetch_add(T.diff + T.plane_stride + c, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                S.lcnt[slot][cell] = 0;
            }
            // (the two strands' sums in one pass: the packed halves of a sum cannot overflow while use_win holds, see far_event)
            const int32_t sp = row_sum_dpp(v);
            const int32_t s0 = (int32_t)(int16_t)(uint16_t)(sp & 0xffff), s1 = (sp - s0) >> 16;
            if (cell == 15u && s0) seg_add(T, 0u, c, s0);
            if (cell == 15u && s1) seg_add(T, 1u, c, s1);
            // (a claimed line is marked for both strands whatever it holds: marking too much only costs the clear a line)
            if (cell == 15u && tag != kEmpty) *reinterpret_cast<uint16_t*>(T.dirty + ((uint64_t)tag << 1)) = 0x0101u;
        }
        for (uint32_t e0 = 0; e0 <= wtop; e0 += kThreads) {
            const uint32_t e = e0 + tid;                                // < kWin: the window is a multiple of the workgroup
            const int32_t v = S.tile[e];
            const int32_t v0 = (int32_t)(int16_t)(uint16_t)(v & 0xffff);
            const int32_t v1 = (v - v0) >> 16;
            if (v) {
                int32_t* g = T.diff + win + e;
                if (v0 && !(T.ablate & 256u)) __hip_atomic_fetch_add(g, v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v1 && !(T.ablate & 256u)) __hip_atomic_fetch_add(g + T.plane_stride, v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                S.tile[e] = 0;
            }
            // the window starts on a multiple of 64: a wave's 64 positions share one segment
            const int32_t sp = wave_sum_dpp(v);
            const int32_t s0 = (int32_t)(int16_t)(uint16_t)(sp & 0xffff), s1 = (sp - s0) >> 16;
            if (lane == 63u) {
                if (s0) seg_add(T, 0u, win + e, s0);
                if (s1) seg_add(T, 1u, win + e, s1);
                // the wave's four lines, both strands: eight consecutive marks (the window starts on a multiple of 64)
                *reinterpret_cast<unsigned long long*>(T.dirty + (((win + e) >> 6) << 3)) = 0x0101010101010101ull;
            }
        }
        __syncthreads();
        CPROF(6);
        for (uint32_t e = tid; e < kLines; e += kThreads) S.ltag[e] = kEmpty;
        if (tid == 0) { S.nq_n = 0; S.nh_n = 0; S.wmax = 0; }
        if (nq + nh) {
            // records of the chunk: the merged table first (wave 0 compacts it with ballots), then the queue (weight 1)
            const unsigned long long qb = S.nq_base;
            if (nh && tid < 64u) {
                uint32_t at = 0;
#pragma unroll
                for (uint32_t e0 = 0; e0 < kNovelH; e0 += 64u) {
                    const unsigned long long k = S.nh_key[e0 + tid];
                    const uint32_t w = S.nh_cnt[e0 + tid];
                    const uint64_t occ = __ballot(k != kEmpty64);
                    if (k != kEmpty64) {
                        const unsigned long long slot = qb + at + (unsigned long long)__popcll(occ & ((1ull << tid) - 1ull));
                        if (slot < T.junc_capacity) { T.junc_keys[slot] = k; T.junc_w[slot] = w; } else err |= ERR_JUNC_CAPACITY;
                        S.nh_key[e0 + tid] = kEmpty64; S.nh_cnt[e0 + tid] = 0;
                    }
                    at += (uint32_t)__popcll(occ);
                }
            }
            for (uint32_t e = tid; e < nq; e += kThreads) {
                if (qb + nh + e < T.junc_capacity) { T.junc_keys[qb + nh + e] = S.nq[e]; T.junc_w[qb + nh + e] = 1u; } else err |= ERR_JUNC_CAPACITY;
            }
        }
    }
#ifdef COUNT_PROF
    if (tid == 0 && blockIdx.x < 1024u) for (int k = 0; k < 8; k++) g_cprof[blockIdx.x][k] = prof_acc[k];
#endif
    if (err) atomicOr(T.err_flags, err);
    __syncthreads();
    for (int k = tid; k < kChrBins; k += kThreads) {
        const uint32_t v = S.chr[k];
        if (v && (uint32_t)(k >> 1) < T.n_chr) atomicAdd(&T.chr_cnt[k], v);
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
        if ((T.ablate & (1u | 64u)) || !(cnt[0] | cnt[1])) continue;
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

}   // namespace

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
    const uint32_t n_chunks = (B.n_frag + kChunk - 1) / kChunk;
    uint32_t blocks = ((n_chunks + 7u) / 8u) * 8u;            // a multiple of 8 for the XCD mapping
    if (blocks > 256u * K_WGS_PER_CU) blocks = 256u * K_WGS_PER_CU;
    hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(kThreads), 0, s, T, B);
#ifdef COUNT_PROF
    {
        static uint32_t h[1024][8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cprof), sizeof h);
        double a[8] = {0};
        for (uint32_t w = 0; w < blocks && w < 1024u; w++) for (int k = 0; k < 8; k++) a[k] += h[w][k];
        fprintf(stderr, "count_prof ticks/wg: fragment phase %.0f | block loop: columns+owner+bucket requests %.0f, junction keys + SpansPoint requests + key search %.0f, "
                        "junction counters %.0f, events+points+novel %.0f | barrier %.0f | flush junction/point caches %.0f, flush lines+window %.0f\n",
                a[0] / blocks, a[1] / blocks, a[2] / blocks, a[3] / blocks, a[7] / blocks, a[4] / blocks, a[5] / blocks, a[6] / blocks);
    }
#endif
}

}   // namespace dev
}   // namespace irf
