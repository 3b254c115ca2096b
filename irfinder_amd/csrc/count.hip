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
// results.  HBM-/latency-bound integer work: no MFMA.
//
// Shape of the kernel.  The input is CSR (fragments -> blocks); walking it fragment by fragment gives every lane a
// chain of dependent loads (offset -> block -> bucket table -> counters) of data-dependent length, and the kernel
// then waits on memory 85 % of its cycles.  So a workgroup processes a chunk of kChunk consecutive fragments in two
// bulk phases:
//   1. fragment-parallel: the fragment columns are read coalesced; per-fragment facts (chromosome base, strand,
//      chromosome length, CSR offset) go to LDS; FragmentsInChr and FragmentsInROI are done here;
//   2. block-parallel: the chunk's blocks are one contiguous range of the block columns, read coalesced, kUnroll
//      independent blocks per lane in flight; the owning fragment comes from a binary search of the LDS offsets;
//      ONE 16-byte bucket-table entry per block end gives the plane map, the SpansPoint range and the known-junction
//      range; a block owns the junction to its right;
//   3. the LDS accumulators (difference window, far-event cache, junction cache, novel-junction queue) are flushed.
#include "kernels.h"

namespace irf {
namespace dev {

namespace {

#ifndef K_THREADS
#define K_THREADS 512
#endif
#ifndef K_FPT
#define K_FPT 2
#endif
#ifndef K_UNROLL
#define K_UNROLL 2
#endif
#ifndef K_WIN
#define K_WIN 4096
#endif
#ifndef K_ECACHE
#define K_ECACHE 1024
#endif
#ifndef K_JCACHE
#define K_JCACHE 512
#endif
#ifndef K_NOVELQ
#define K_NOVELQ 512
#endif
#ifndef K_WGS_PER_CU
#define K_WGS_PER_CU 2
#endif
constexpr int kThreads = K_THREADS;
constexpr int kFragsPerThread = K_FPT;
constexpr int kChunk = kThreads * kFragsPerThread;   // fragments per workgroup iteration
constexpr int kChunkLog2 = kChunk == 1024 ? 10 : kChunk == 512 ? 9 : kChunk == 256 ? 8 : kChunk == 2048 ? 11 : 0;
static_assert((1 << kChunkLog2) == kChunk, "kChunkLog2");
constexpr int kUnroll = K_UNROLL;          // blocks in flight per lane in phase 2
constexpr int kChrBins = 512;       // LDS histogram covers refIDs < 256; larger ones go straight to global
constexpr uint32_t kJCache = K_JCACHE;   // LDS cache of known-junction counters (direct-mapped on the known-key index)
constexpr uint32_t kEmpty = 0xffffffffu;
constexpr uint32_t kECache = K_ECACHE;  // LDS cache of difference events outside the window (direct-mapped on plane address)
constexpr int kECacheBits = kECache == 1024 ? 10 : kECache == 512 ? 9 : kECache == 2048 ? 11 : kECache == 256 ? 8 : 0;
static_assert(kECacheBits > 0, "K_ECACHE");
constexpr uint32_t kWin = K_WIN;     // positions of the LDS difference-array window (per strand)
constexpr uint32_t kNovelQ = K_NOVELQ;   // LDS queue of novel-junction events per chunk
constexpr unsigned long long kFragValid = 1ull << 63, kFragDir = 1ull << 62, kFragBaseMask = (1ull << 40) - 1ull;

// A difference event outside the chunk's window (typically the far side of a splice: many reads of a chunk share
// the same exon boundary).  Direct-mapped LDS cache on the plane address; a slot held by another address falls
// through to a global atomic.
__device__ __forceinline__ void far_event(const CountTables& T, uint32_t* s_etag, int32_t* s_ecnt, uint64_t g, uint32_t dir, int v) {
    const uint64_t gd = g * 2u + dir;
    const uint32_t c = (uint32_t)gd & (kECache - 1), want = (uint32_t)(gd >> kECacheBits);
    uint32_t tag = s_etag[c];
    if (tag == kEmpty) { const uint32_t old = atomicCAS(&s_etag[c], kEmpty, want); tag = old == kEmpty ? want : old; }
    if (tag == want) atomicAdd(&s_ecnt[c], v);
    else __hip_atomic_fetch_add(T.diff + (uint64_t)dir * T.plane_stride + g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t plane_of(const CountTables& T, uint32_t m, uint64_t g) {
    if (!T.pmap) return g;
    const uint64_t c = (uint64_t)(m >> 1) << kBucketShift;
    return (m & 1u) ? c + (g & ((1u << kBucketShift) - 1u)) : c;
}

__global__ __launch_bounds__(kThreads) void count_kernel(CountTables T, BatchDev B) {
    __shared__ uint32_t s_off[kChunk + 1];               // CSR offsets of the chunk's fragments (clamped to n_blk)
    __shared__ unsigned long long s_fbase[kChunk];       // chr_base | strand << 62 | counted << 63
    __shared__ uint32_t s_fclen[kChunk];                 // chromosome length
    __shared__ __align__(16) int32_t s_tile[2 * kWin];
    __shared__ uint32_t s_etag[kECache];
    __shared__ int32_t s_ecnt[kECache];
    __shared__ uint32_t s_jtag[kJCache];
    __shared__ uint32_t s_jcnt[2][kJCache];
    __shared__ unsigned long long s_nq[kNovelQ];
    __shared__ uint32_t s_chr[kChrBins];
    __shared__ unsigned long long s_win, s_nq_base;
    __shared__ uint32_t s_nq_n, s_wmax;
    const uint32_t tid = threadIdx.x;
    for (int k = tid; k < kChrBins; k += kThreads) s_chr[k] = 0;
    for (int k = tid; k < (int)(2 * kWin); k += kThreads) s_tile[k] = 0;
    for (int k = tid; k < (int)kECache; k += kThreads) { s_etag[k] = kEmpty; s_ecnt[k] = 0; }
    for (int k = tid; k < (int)kJCache; k += kThreads) { s_jtag[k] = kEmpty; s_jcnt[0][k] = 0; s_jcnt[1][k] = 0; }
    if (tid == 0) { s_nq_n = 0; s_wmax = 0; }
    __syncthreads();

    const uint32_t lane = tid & 63u;
    uint32_t err = 0;
    const uint32_t n_chunks = (B.n_frag + kChunk - 1) / kChunk;
    // XCD-aware chunk order: workgroup b runs on XCD b % 8 (round-robin dispatch); every XCD gets a contiguous
    // eighth of the (coordinate-sorted) batch, so that neighbouring chunks — same table lines, overlapping
    // difference windows — share one L2
    const uint32_t xcd = blockIdx.x & 7u, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
    const uint32_t per_xcd = (n_chunks + 7u) / 8u;
    const uint32_t c_end = (xcd + 1u) * per_xcd < n_chunks ? (xcd + 1u) * per_xcd : n_chunks;
    for (uint32_t c = xcd * per_xcd + wg_in_xcd; c < c_end; c += wgs_per_xcd) {
        const uint32_t f0 = c * kChunk;
        // ---------------- 1. fragment-parallel ----------------
        {
            uint32_t chr[kFragsPerThread], fs[kFragsPerThread], fe[kFragsPerThread], off[kFragsPerThread], flg[kFragsPerThread];
#pragma unroll
            for (int j = 0; j < kFragsPerThread; j++) {
                // loads are unconditional on a clamped index (a conditional load becomes a branch with a full wait
                // behind it, which serialises the round trips); the values of lanes past the end are discarded
                const uint32_t i = f0 + j * kThreads + tid;
                const bool in = i < B.n_frag;
                const uint32_t ic = in ? i : B.n_frag - 1u;
                chr[j] = B.frag_chr[ic];
                flg[j] = B.frag_flags[ic];
                fs[j] = B.frag_start[ic];
                fe[j] = B.frag_end[ic];
                off[j] = B.frag_blk_off[in ? i : B.n_frag];
                if (!in) chr[j] = kEmpty;
            }
            if (tid == 0) {
                const uint32_t last = f0 + kChunk < B.n_frag ? f0 + kChunk : B.n_frag;
                const uint32_t o = B.frag_blk_off[last];
                s_off[kChunk] = o < B.n_blk ? o : B.n_blk;
            }
            uint32_t key[kFragsPerThread], clen[kFragsPerThread];
            uint64_t base[kFragsPerThread];
#pragma unroll
            for (int j = 0; j < kFragsPerThread; j++) {
                const bool in = (f0 + j * kThreads + tid) < B.n_frag;
                const bool ok = chr[j] < T.n_chr;
                if (in && !ok) err |= ERR_CHR_RANGE;
                const uint32_t cc = ok ? chr[j] : 0u;            // tables hold at least one entry
                const bool counted = (T.chr_owned[cc] != 0) && ok;
                base[j] = T.chr_base[cc];
                clen[j] = T.chr_len[cc];
                if (!counted) { base[j] = 0ull; clen[j] = 0u; }
                const uint32_t dir = flg[j] & 1u;
                const uint32_t li = j * kThreads + tid;
                s_off[li] = off[j] < B.n_blk ? off[j] : B.n_blk;
                s_fbase[li] = counted ? (base[j] | (dir ? kFragDir : 0ull) | kFragValid) : 0ull;
                s_fclen[li] = clen[j];
                key[j] = counted ? chr[j] * 2u + dir : kEmpty;
            }
            if (tid == 0) {
                // the window starts at the chunk's first fragment (coordinate-sorted input: the leftmost one)
                unsigned long long w = 0;
                if (key[0] != kEmpty && fs[0] <= clen[0]) {
                    const uint64_t g = base[0] + fs[0];
                    w = plane_of(T, T.pmap ? T.bkt[g >> kBucketShift].x : 0u, g) & ~63ull;
                }
                s_win = w;
            }
            // FragmentsInROI: first ROI (by end) whose end >= fragment end; counted if it contains the fragment
            if (!(T.ablate & 4u) && T.n_roi_dev) {
                uint32_t r[kFragsPerThread];
                bool extent_ok[kFragsPerThread];
#pragma unroll
                for (int j = 0; j < kFragsPerThread; j++) {
                    extent_ok[j] = key[j] != kEmpty && fs[j] <= fe[j] && fe[j] <= clen[j];
                    r[j] = T.roi_bucket[extent_ok[j] ? (base[j] + fe[j]) >> kBucketShift : 0ull];
                }
                uint64_t rge[kFragsPerThread], rgs[kFragsPerThread];
#pragma unroll
                for (int j = 0; j < kFragsPerThread; j++) {
                    if (r[j] >= T.n_roi_dev) { extent_ok[j] = false; r[j] = 0u; }
                    rge[j] = T.roi_gend[r[j]];
                    rgs[j] = T.roi_gstart[r[j]];
                }
#pragma unroll
                for (int j = 0; j < kFragsPerThread; j++) {
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
                for (int j = 0; j < kFragsPerThread; j++) {
                    const uint32_t k = key[j];
                    uint64_t todo = __ballot(k != kEmpty);
                    while (todo) {
                        const int lead = __ffsll((long long)todo) - 1;
                        const uint32_t lk = (uint32_t)__builtin_amdgcn_readlane((int)k, lead);
                        const uint64_t same = __ballot(k == lk);
                        if ((int)lane == lead) {
                            const uint32_t n = (uint32_t)__popcll(same);
                            if (lk < (uint32_t)kChrBins) atomicAdd(&s_chr[lk], n);
                            else atomicAdd(&T.chr_cnt[lk], n);
                        }
                        todo &= ~same;
                    }
                }
            }
        }
        __syncthreads();

        // ---------------- 2. block-parallel ----------------
        const uint32_t kb0 = s_off[0], kb1 = s_off[kChunk];
        const unsigned long long win = s_win;
        uint32_t wmax = 0;
        for (uint32_t kk = kb0; kk < kb1 && !(T.ablate & 32u); kk += kThreads * kUnroll) {
            uint32_t k[kUnroll], bs[kUnroll], bl[kUnroll], rd[kUnroll], bs2[kUnroll], bl2[kUnroll], rd2[kUnroll];
            bool act[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                k[u] = kk + u * kThreads + tid;
                act[u] = k[u] < kb1;
                const bool nx = k[u] + 1u < kb1;
                const uint32_t kc = act[u] ? k[u] : kb1 - 1u, kn = nx ? k[u] + 1u : kb1 - 1u;     // kb1 <= n_blk
                bs[u] = B.blk_start[kc];
                bl[u] = B.blk_len[kc];
                rd[u] = B.blk_read[kc];
                bs2[u] = B.blk_start[kn];
                bl2[u] = B.blk_len[kn];
                rd2[u] = nx ? (uint32_t)B.blk_read[kn] : 0xffu;
            }
            // owner = the last fragment of the chunk whose offset is <= k (empty fragments share an offset with
            // their successor and are never picked)
            unsigned long long fb[kUnroll];
            uint32_t clen[kUnroll];
            bool next_same[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                uint32_t lo = 0, hi = kChunk;
                if (T.ablate & 64u) { lo = (k[u] - kb0) / 3u; lo = lo < (uint32_t)kChunk ? lo : kChunk - 1; hi = lo; }
#pragma unroll
                for (int s = 0; s < kChunkLog2; s++) {
                    if (T.ablate & 64u) break;
                    const uint32_t mid = (lo + hi) >> 1;
                    if (s_off[mid] <= k[u]) lo = mid; else hi = mid;
                }
                fb[u] = act[u] ? s_fbase[lo] : 0ull;
                clen[u] = s_fclen[lo];
                next_same[u] = k[u] + 1u < s_off[lo + 1];
            }
            uint4 ea[kUnroll], eb[kUnroll];
            uint64_t gbs[kUnroll], gbe[kUnroll];
            bool ok[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                const uint32_t be = bs[u] + bl[u];
                const bool valid = (fb[u] & kFragValid) != 0ull;
                const bool in_range = be >= bs[u] && be <= clen[u];
                if (valid && !in_range) err |= ERR_BLOCK_RANGE;
                ok[u] = valid && in_range;
                const uint64_t base = fb[u] & kFragBaseMask;
                gbs[u] = base + bs[u];
                gbe[u] = base + be;
                ea[u] = T.bkt[ok[u] ? gbs[u] >> kBucketShift : 0ull];
                eb[u] = T.bkt[ok[u] ? gbe[u] >> kBucketShift : 0ull];
            }
            // junction to the right of the block: both flanks long enough, same read, same fragment
            bool jc[kUnroll];
            uint64_t jkey[kUnroll], kfirst[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                jc[u] = ok[u] && next_same[u] && rd2[u] == rd[u] && bl[u] >= T.junc_min_flank && bl2[u] >= T.junc_min_flank &&
                        !(T.ablate & 8u);
                const uint32_t be = bs[u] + bl[u];
                const uint32_t span = bs2[u] - be;
                if (jc[u] && (bs2[u] < be || span >= (1u << kJuncSpanBits))) { err |= ERR_JUNC_RANGE; jc[u] = false; }
                jkey[u] = (gbe[u] << kJuncSpanBits) | span;
                const bool kq = jc[u] && (eb[u].w >> 16);
                kfirst[u] = T.known_keys[kq ? eb[u].z : 0u];                  // the table holds at least one entry
                if (!kq) kfirst[u] = ~0ull;
            }
#pragma unroll
            for (int u = 0; u < kUnroll; u++) {
                if (!ok[u]) continue;
                const uint32_t dir = (fb[u] & kFragDir) ? 1u : 0u;
                if (bl[u] > 0u) {
                    if (!(T.ablate & 1u)) {
                        // CoverageBlocks / FragmentsMap: difference events, LDS window first
                        const uint64_t cs = plane_of(T, ea[u].x, gbs[u]), ce = plane_of(T, eb[u].x, gbe[u]);
                        const uint64_t os = cs - win, oe = ce - win;
                        int32_t* tile = s_tile + dir * kWin;
                        if (os < kWin) { atomicAdd(&tile[os], 1); wmax = (uint32_t)os > wmax ? (uint32_t)os : wmax; }
                        else far_event(T, s_etag, s_ecnt, cs, dir, 1);
                        if (oe < kWin) { atomicAdd(&tile[oe], -1); wmax = (uint32_t)oe > wmax ? (uint32_t)oe : wmax; }
                        else far_event(T, s_etag, s_ecnt, ce, dir, -1);
                    }
                    // SpansPoint: the points of the buckets under the block, filtered to [lo, hi]
                    if (bl[u] > T.sp_min_len && !(T.ablate & 2u)) {
                        uint32_t p = ea[u].y;
                        const uint32_t pn = eb[u].w & 0xffffu;
                        const uint32_t p1 = pn == 0xffffu ? T.bkt[(gbe[u] >> kBucketShift) + 1u].y : eb[u].y + pn;
                        if (p < p1) {
                            const uint64_t lo = gbs[u] + T.sp_lo_add, hi = gbe[u] - T.sp_hi_sub;
                            uint32_t* __restrict__ sp_cnt = T.sp_cnt + (uint64_t)dir * T.n_sp;
                            for (; p < p1; p++) {
                                const uint64_t g = T.sp_gpos[p];
                                if (g > hi) break;
                                if (g >= lo) atomicAdd(&sp_cnt[T.sp_row[p]], 1u);
                            }
                        }
                    }
                }
                if (jc[u]) {
                    // JunctionCount: known junction -> its counter (through the per-chunk LDS cache); else a novel event
                    int64_t hit = -1;
                    const uint32_t kn = eb[u].w >> 16;
                    if (kn) {
                        const uint32_t i0 = eb[u].z;
                        if (kfirst[u] >= jkey[u]) {
                            if (kfirst[u] == jkey[u]) hit = i0;
                        } else {
                            uint32_t lo = i0 + 1u;
                            uint32_t hi = kn == 0xffffu ? T.bkt[(gbe[u] >> kBucketShift) + 1u].z : i0 + kn;
                            while (hi - lo > 4u) {        // many junctions share this donor bucket: bisect first
                                const uint32_t mid = lo + ((hi - lo) >> 1);
                                if (T.known_keys[mid] < jkey[u]) lo = mid + 1u; else hi = mid + 1u;
                            }
                            for (; lo < hi; lo++) {
                                const uint64_t kq = T.known_keys[lo];
                                if (kq >= jkey[u]) { if (kq == jkey[u]) hit = lo; break; }
                            }
                        }
                    }
                    if (hit >= 0) {
                        const uint32_t idx = (uint32_t)hit, cslot = idx & (kJCache - 1);
                        uint32_t tag = s_jtag[cslot];
                        if (tag == kEmpty) { const uint32_t old = atomicCAS(&s_jtag[cslot], kEmpty, idx); tag = old == kEmpty ? idx : old; }
                        if (tag == idx) atomicAdd(&s_jcnt[dir][cslot], 1u);
                        else atomicAdd(&T.known_cnt[(uint64_t)dir * T.n_kj + T.known_kjidx[idx]], 1u);
                    } else {
                        const uint64_t ev = (gbe[u] << kJuncLeftShift) | ((jkey[u] & ((1ull << kJuncSpanBits) - 1ull)) << 1) | dir;
                        const uint32_t q = atomicAdd(&s_nq_n, 1u);
                        if (q < kNovelQ) {
                            s_nq[q] = ev;
                        } else {
                            // queue full (a run without ref-sj.ref): the lanes that got here reserve slots together
                            const uint64_t m = __ballot(1);
                            unsigned long long b = 0;
                            if ((int)lane == __ffsll((long long)m) - 1) b = atomicAdd(T.junc_count, (unsigned long long)__popcll(m));
                            b = __shfl(b, __ffsll((long long)m) - 1);
                            const unsigned long long slot = b + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
                            if (slot < T.junc_capacity) T.junc_keys[slot] = ev; else err |= ERR_JUNC_CAPACITY;
                        }
                    }
                }
            }
        }
        if (wmax) atomicMax(&s_wmax, wmax);
        __syncthreads();

        // ---------------- 3. flush the LDS accumulators ----------------
        const uint32_t nq = s_nq_n < kNovelQ ? s_nq_n : kNovelQ;
        const uint32_t wtop = s_wmax;
        if (tid == 0 && nq) s_nq_base = atomicAdd(T.junc_count, (unsigned long long)nq);
        for (uint32_t e = tid; e < kECache; e += kThreads) {
            const uint32_t tag = s_etag[e];
            if (tag != kEmpty) {
                const int32_t v = s_ecnt[e];
                const uint64_t gd = ((uint64_t)tag << kECacheBits) | e;
                if (v) __hip_atomic_fetch_add(T.diff + (gd & 1u) * T.plane_stride + (gd >> 1), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_etag[e] = kEmpty; s_ecnt[e] = 0;
            }
        }
        for (uint32_t e = tid; e < kJCache; e += kThreads) {
            const uint32_t idx = s_jtag[e];
            if (idx != kEmpty) {
                const uint32_t c0 = s_jcnt[0][e], c1 = s_jcnt[1][e];
                const uint32_t row = T.known_kjidx[idx];
                if (c0) atomicAdd(&T.known_cnt[row], c0);
                if (c1) atomicAdd(&T.known_cnt[(uint64_t)T.n_kj + row], c1);
                s_jtag[e] = kEmpty; s_jcnt[0][e] = 0; s_jcnt[1][e] = 0;
            }
        }
        // the window, only as far as it was touched
        for (uint32_t e = tid * 4u; e <= wtop; e += kThreads * 4u) {
#pragma unroll
            for (uint32_t d = 0; d < 2u; d++) {
                int4* cell = reinterpret_cast<int4*>(&s_tile[d * kWin + e]);
                const int4 v = *cell;
                if (v.x | v.y | v.z | v.w) {
                    int32_t* g = T.diff + (uint64_t)d * T.plane_stride + win + e;
                    if (v.x) __hip_atomic_fetch_add(g + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v.y) __hip_atomic_fetch_add(g + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v.z) __hip_atomic_fetch_add(g + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v.w) __hip_atomic_fetch_add(g + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *cell = make_int4(0, 0, 0, 0);
                }
            }
        }
        __syncthreads();
        if (tid == 0) { s_nq_n = 0; s_wmax = 0; }
        if (nq) {
            const unsigned long long qb = s_nq_base;
            for (uint32_t e = tid; e < nq; e += kThreads) {
                if (qb + e < T.junc_capacity) T.junc_keys[qb + e] = s_nq[e]; else err |= ERR_JUNC_CAPACITY;
            }
        }
    }
    if (err) atomicOr(T.err_flags, err);
    __syncthreads();
    for (int k = tid; k < kChrBins; k += kThreads) {
        const uint32_t v = s_chr[k];
        if (v && (uint32_t)(k >> 1) < T.n_chr) atomicAdd(&T.chr_cnt[k], v);
    }
}

}   // namespace

void launch_count(const CountTables& T, const BatchDev& B, hipStream_t s) {
    if (B.n_frag == 0) return;
    // 70 KB of LDS per workgroup: two workgroups per CU, 256 CUs; a multiple of 8 for the XCD mapping
    const uint32_t n_chunks = (B.n_frag + kChunk - 1) / kChunk;
    uint32_t blocks = ((n_chunks + 7u) / 8u) * 8u;
    if (blocks > 256u * K_WGS_PER_CU) blocks = 256u * K_WGS_PER_CU;
    hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(kThreads), 0, s, T, B);
}

}   // namespace dev
}   // namespace irf
