// count.hip — the fused per-batch counting kernel (gfx950).
//
// One pass over a structure-of-arrays fragment batch replaces, for every
// fragment, the ProcessBlocks calls of ⟨upstream⟩ ReadBlockProcessor.cpp
// FragmentsInChr / FragmentsInROI / JunctionCount / SpansPoint and of
// ReadBlockProcessor_CoverageBlocks.cpp CoverageBlocks (+ FragmentsMap):
//   * FragmentsInChr : LDS histogram per workgroup, flushed with global atomics
//   * FragmentsInROI : bucket index over the end-sorted ROI table + short walk
//   * CoverageBlocks : +1 / -1 into the per-strand difference array (depth comes
//                      from the prefix scan at finalize)
//   * SpansPoint     : bucket index over the plane + short walk of the sorted points
//   * JunctionCount  : events on a known junction (ref-sj.ref) bump that junction's
//                      counter (bucket index + walk; LDS-cached per chunk); novel junctions
//                      are appended as packed (chr,left,span,strand) events to a
//                      buffer and sort-reduced at finalize
// All outputs are commutative integer sums, so any batch order gives identical
// results.  HBM-bound integer work: no MFMA.
#include "kernels.h"

namespace irf {
namespace dev {

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kFragsPerThread = 2;
constexpr int kChunk = kThreads * kFragsPerThread;   // fragments per workgroup iteration
constexpr int kChrBins = 512;       // LDS histogram covers refIDs < 256; larger ones go straight to global
constexpr uint32_t kJCache = 512;   // LDS cache of known-junction counters (direct-mapped on the table row)
constexpr uint32_t kEmpty = 0xffffffffu;
constexpr uint32_t kECache = 1024;  // LDS cache of difference events outside the window (direct-mapped on plane address)
constexpr uint32_t kWin = 4096;     // positions of the LDS difference-array window (per strand)

// row of (gleft, span) in the known-junction keys, or -1: bucket index on gleft, then a short walk
__device__ __forceinline__ int32_t find_known(const CountTables& T, uint64_t gleft, uint32_t span) {
    const uint64_t key = (gleft << kJuncSpanBits) | span;
    if (gleft >= T.full_size) return -1;             // malformed block: reported by the range checks
    const uint64_t b = gleft >> kBucketShift;
    uint32_t i = T.known_bucket[b];
    const uint32_t i1 = T.known_bucket[b + 1];
    for (; i < i1; i++) {
        const uint64_t k = T.known_keys[i];
        if (k >= key) return k == key ? (int32_t)i : -1;
    }
    return -1;
}

// A difference event outside the chunk's window (typically the far side of a splice: many reads of a chunk share
// the same exon boundary).  Direct-mapped LDS cache on the plane address; a slot held by another address falls
// through to a global atomic.
__device__ __forceinline__ void far_event(const CountTables& T, uint32_t* s_etag, int32_t* s_ecnt, uint64_t g, uint32_t dir, int v) {
    const uint64_t gd = g * 2u + dir;
    const uint32_t c = (uint32_t)gd & (kECache - 1), want = (uint32_t)(gd >> 10);
    uint32_t tag = s_etag[c];
    if (tag == kEmpty) { const uint32_t old = atomicCAS(&s_etag[c], kEmpty, want); tag = old == kEmpty ? want : old; }
    if (tag == want) atomicAdd(&s_ecnt[c], v);
    else __hip_atomic_fetch_add(T.diff + (uint64_t)dir * T.plane_stride + g, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t lo, uint32_t hi, uint32_t key) {
    // first index in [lo,hi) with a[idx] >= key
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// A workgroup walks chunks of kChunk consecutive fragments.  Per chunk:
//   1. every thread walks the junctions of its fragments: hits on a known junction are counted in place, novel ones
//      are counted per thread; one block-wide scan and ONE global atomic reserve the chunk's slice of the novel-event
//      buffer (a single shared counter bumped per wave costs >15 ms at 50 M pairs);
//   2. the fragments are processed; difference events are mapped to plane coordinates (kernels.h: plane_coord); those
//      that fall in the chunk's window [win, win + kWin) of the plane
//      (coordinate-sorted input: nearly all of them) are accumulated in LDS, the rest go to global atomics;
//   3. the window is flushed: one global atomic per non-zero position instead of one per event.
__global__ __launch_bounds__(kThreads) void count_kernel(CountTables T, BatchDev B) {
    __shared__ uint32_t s_chr[kChrBins];
    __shared__ int32_t s_tile[2 * kWin];
    __shared__ uint32_t s_etag[kECache];
    __shared__ int32_t s_ecnt[kECache];
    __shared__ uint32_t s_jtag[kJCache];
    __shared__ uint32_t s_jcnt[2][kJCache];
    __shared__ uint32_t s_wsum[kWaves];
    __shared__ unsigned long long s_base;
    __shared__ unsigned long long s_win;
    for (int k = threadIdx.x; k < kChrBins; k += kThreads) s_chr[k] = 0;
    for (int k = threadIdx.x; k < (int)(2 * kWin); k += kThreads) s_tile[k] = 0;
    for (int k = threadIdx.x; k < (int)kECache; k += kThreads) { s_etag[k] = kEmpty; s_ecnt[k] = 0; }
    for (int k = threadIdx.x; k < (int)kJCache; k += kThreads) { s_jtag[k] = kEmpty; s_jcnt[0][k] = 0; s_jcnt[1][k] = 0; }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t err = 0;
    const uint32_t n_chunks = (B.n_frag + kChunk - 1) / kChunk;
    for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const uint32_t f0 = c * kChunk;
        // ---- 1. junction events of this thread's fragments: known junctions are counted here, novel ones are
        //         numbered (bit e of `novel` = this thread's e-th event is novel) so that pass 2 can place them
        uint32_t my_events = 0, n_ev = 0;
        uint64_t novel = 0;
        if (!(T.ablate & 8u)) {
#pragma unroll 1
            for (int j = 0; j < kFragsPerThread; j++) {
                const uint32_t i = f0 + j * kThreads + threadIdx.x;
                if (i >= B.n_frag) break;
                const uint32_t chr = B.frag_chr[i];
                if (chr >= T.n_chr || !T.chr_owned[chr]) continue;
                const uint32_t dir = B.frag_flags[i] & 1u;
                const uint64_t base = T.chr_base[chr];
                const uint32_t k0 = B.frag_blk_off[i], k1 = B.frag_blk_off[i + 1];
                uint32_t prev_end = 0, prev_len = 0, prev_read = 0xffu;
                for (uint32_t k = k0; k < k1; k++) {
                    const uint32_t bs = B.blk_start[k], bl = B.blk_len[k], rd = B.blk_read[k];
                    if (rd == prev_read && prev_len >= T.junc_min_flank && bl >= T.junc_min_flank) {
                        const uint64_t gleft = base + prev_end;
                        const uint32_t span = bs - prev_end;
                        int32_t hit = -1;
                        if (bs >= prev_end && span < (1u << kJuncSpanBits) && gleft < kJuncMaxGlobal)
                            hit = find_known(T, gleft, span);
                        if (hit >= 0) {
                            // per-chunk LDS cache of the hot junctions (sorted input hits the same few again and again)
                            const uint32_t row = T.known_kjidx[hit], c = row & (kJCache - 1);
                            uint32_t tag = s_jtag[c];
                            if (tag == kEmpty) { const uint32_t old = atomicCAS(&s_jtag[c], kEmpty, row); tag = old == kEmpty ? row : old; }
                            if (tag == row) atomicAdd(&s_jcnt[dir][c], 1u);
                            else atomicAdd(&T.known_cnt[(uint64_t)dir * T.n_kj + row], 1u);
                        } else {
                            my_events++;
                            if (n_ev < 64u) novel |= 1ull << n_ev;
                        }
                        n_ev++;
                    }
                    prev_end = bs + bl; prev_len = bl; prev_read = rd;
                }
            }
        }
        uint32_t incl = my_events;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if ((int)lane >= d) incl += o;
        }
        if (lane == 63) s_wsum[wave] = incl;
        if (threadIdx.x == 0) {
            const uint32_t chr = B.frag_chr[f0];
            unsigned long long w = 0;
            if (chr < T.n_chr && T.chr_owned[chr] && B.frag_start[f0] <= T.chr_len[chr])
                w = plane_coord(T.pmap, T.chr_base[chr] + B.frag_start[f0]) & ~63ull;
            s_win = w;
        }
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) { const uint32_t v = s_wsum[w]; if (w < (int)wave) before += v; total += v; }
        if (threadIdx.x == 0) s_base = total ? atomicAdd(T.junc_count, (unsigned long long)total) : 0ull;
        __syncthreads();
        unsigned long long slot = s_base + before + incl - my_events;
        const unsigned long long win = s_win;
        uint32_t ev_i = 0;

        // ---- 2. the fragments
#pragma unroll 1
        for (int j = 0; j < kFragsPerThread; j++) {
            const uint32_t i = f0 + j * kThreads + threadIdx.x;
            if (i >= B.n_frag) break;
            const uint32_t chr = B.frag_chr[i];
            if (chr >= T.n_chr) { err |= ERR_CHR_RANGE; continue; }
            if (!T.chr_owned[chr]) continue;
            const uint32_t dir = B.frag_flags[i] & 1u;
            // FragmentsInChr
            if (!(T.ablate & 16u)) {
                if (chr < kChrBins / 2) atomicAdd(&s_chr[chr * 2 + dir], 1u);
                else atomicAdd(&T.chr_cnt[chr * 2 + dir], 1u);
            }
            const uint64_t base = T.chr_base[chr];
            const uint32_t clen = T.chr_len[chr];
            const uint32_t fs = B.frag_start[i], fe = B.frag_end[i];
            const bool extent_ok = fs <= fe && fe <= clen;
            // FragmentsInROI: first ROI (by end) whose end >= fragment end; counted if it contains the fragment
            if (!(T.ablate & 4u) && extent_ok && T.n_roi_dev) {
                const uint64_t gfs = base + fs, gfe = base + fe;
                uint32_t r = T.roi_bucket[gfe >> kBucketShift];
                while (r < T.n_roi_dev && T.roi_gend[r] < gfe) r++;
                if (r < T.n_roi_dev && T.roi_gstart[r] <= gfs)
                    atomicAdd(&T.roi_cnt[(uint64_t)dir * T.n_roi + T.roi_idx[r]], 1u);
            }
            // SpansPoint: does any point lie under the fragment at all?  (two adjacent bucket entries)
            const bool sp_any = !(T.ablate & 2u) &&
                                (!extent_ok || T.sp_bucket[(base + fs) >> kBucketShift] != T.sp_bucket[((base + fe) >> kBucketShift) + 1]);
            // per aligned block
            int32_t* tile = s_tile + dir * kWin;
            uint32_t* __restrict__ sp_cnt = T.sp_cnt + (uint64_t)dir * T.n_sp;
            const uint32_t k0 = B.frag_blk_off[i], k1 = B.frag_blk_off[i + 1];
            uint32_t prev_end = 0, prev_len = 0, prev_read = 0xffu;
            for (uint32_t k = k0; k < k1; k++) {
                const uint32_t bs = B.blk_start[k], bl = B.blk_len[k], rd = B.blk_read[k];
                uint32_t be = bs + bl;
                if (be > clen || be < bs) { err |= ERR_BLOCK_RANGE; be = clen; }
                if (bs < be && !(T.ablate & 1u)) {
                    // CoverageBlocks / FragmentsMap: difference events, LDS window first
                    const uint64_t gs = plane_coord(T.pmap, base + bs), ge = plane_coord(T.pmap, base + be);
                    const uint64_t os = gs - win, oe = ge - win;
                    if (os < kWin) atomicAdd(&tile[os], 1);
                    else far_event(T, s_etag, s_ecnt, gs, dir, 1);
                    if (oe < kWin) atomicAdd(&tile[oe], -1);
                    else far_event(T, s_etag, s_ecnt, ge, dir, -1);
                }
                if (bs < be && sp_any) {
                    // SpansPoint
                    if (bl > T.sp_min_len) {
                        const uint64_t lo = base + bs + T.sp_lo_add, hi = base + be - T.sp_hi_sub;
                        uint32_t p = T.sp_bucket[lo >> kBucketShift];
                        const uint32_t p1 = T.sp_bucket[(hi >> kBucketShift) + 1];
                        for (; p < p1; p++) {
                            const uint64_t g = T.sp_gpos[p];
                            if (g > hi) break;
                            if (g >= lo) atomicAdd(&sp_cnt[T.sp_row[p]], 1u);
                        }
                    }
                }
                // JunctionCount: gap between two consecutive blocks of the same read (novel junctions only; pass 1
                // has already counted the known ones)
                if ((rd == prev_read) && prev_len >= T.junc_min_flank && bl >= T.junc_min_flank && !(T.ablate & 8u)) {
                    const uint64_t gleft = base + prev_end;
                    const uint32_t span = bs - prev_end;
                    const bool bad = bs < prev_end || span >= (1u << kJuncSpanBits) || gleft >= kJuncMaxGlobal;
                    bool is_novel;
                    if (ev_i < 64u) is_novel = (novel >> ev_i) & 1ull;
                    else is_novel = bad || find_known(T, gleft, span) < 0;
                    ev_i++;
                    if (is_novel) {
                        if (bad) {
                            err |= ERR_JUNC_RANGE;
                            if (slot < T.junc_capacity) T.junc_keys[slot] = 0ull;      // keep the slot defined; the run is rejected
                        } else if (slot >= T.junc_capacity) {
                            err |= ERR_JUNC_CAPACITY;
                        } else {
                            T.junc_keys[slot] = (gleft << kJuncLeftShift) | ((uint64_t)span << 1) | dir;
                        }
                        slot++;
                    }
                }
                prev_end = bs + bl; prev_len = bl; prev_read = rd;
            }
        }
        __syncthreads();
        // ---- 3. flush the far-event cache, the known-junction cache and the window
        for (uint32_t k = threadIdx.x; k < kECache; k += kThreads) {
            const uint32_t tag = s_etag[k];
            if (tag != kEmpty) {
                const int32_t v = s_ecnt[k];
                const uint64_t gd = ((uint64_t)tag << 10) | k;
                if (v) __hip_atomic_fetch_add(T.diff + (gd & 1u) * T.plane_stride + (gd >> 1), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_etag[k] = kEmpty; s_ecnt[k] = 0;
            }
        }
        for (uint32_t k = threadIdx.x; k < kJCache; k += kThreads) {
            const uint32_t row = s_jtag[k];
            if (row != kEmpty) {
                const uint32_t c0 = s_jcnt[0][k], c1 = s_jcnt[1][k];
                if (c0) atomicAdd(&T.known_cnt[row], c0);
                if (c1) atomicAdd(&T.known_cnt[(uint64_t)T.n_kj + row], c1);
                s_jtag[k] = kEmpty; s_jcnt[0][k] = 0; s_jcnt[1][k] = 0;
            }
        }
        for (uint32_t k = threadIdx.x; k < 2 * kWin; k += kThreads) {
            const int32_t v = s_tile[k];
            if (v) {
                const uint32_t d = k >= kWin ? 1u : 0u;
                __hip_atomic_fetch_add(T.diff + (uint64_t)d * T.plane_stride + win + (k - d * kWin), v, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                s_tile[k] = 0;
            }
        }
        __syncthreads();
    }
    if (err) atomicOr(T.err_flags, err);
    __syncthreads();
    for (int k = threadIdx.x; k < kChrBins; k += kThreads) {
        const uint32_t v = s_chr[k];
        if (v && (uint32_t)(k >> 1) < T.n_chr) atomicAdd(&T.chr_cnt[k], v);
    }
}

}   // namespace

void launch_count(const CountTables& T, const BatchDev& B, hipStream_t s) {
    if (B.n_frag == 0) return;
    uint32_t blocks = (B.n_frag + kChunk - 1) / kChunk;
    if (blocks > 256u * 3u) blocks = 256u * 3u;
    hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(kThreads), 0, s, T, B);
}

}   // namespace dev
}   // namespace irf
