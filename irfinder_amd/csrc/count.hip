// count.hip — the fused per-batch counting kernel (gfx950).
//
// One pass over a structure-of-arrays fragment batch replaces, for every
// fragment, the ProcessBlocks calls of ⟨upstream⟩ ReadBlockProcessor.cpp
// FragmentsInChr / FragmentsInROI / JunctionCount / SpansPoint and of
// ReadBlockProcessor_CoverageBlocks.cpp CoverageBlocks (+ FragmentsMap):
//   * FragmentsInChr : LDS histogram per workgroup, flushed with global atomics
//   * FragmentsInROI : binary search over the (chr,end)-sorted ROI table
//   * CoverageBlocks : +1 / -1 into the per-strand difference array (depth comes
//                      from the prefix scan at finalize)
//   * SpansPoint     : binary search over the sorted boundary points
//   * JunctionCount  : packed (chr,left,span,strand) events appended to a buffer
//                      (wave-aggregated reservation), sort-reduced at finalize
// All outputs are commutative integer sums, so any batch order gives identical
// results.  HBM-bound integer work: no MFMA.
#include "kernels.h"

namespace irf {
namespace dev {

namespace {

constexpr int kThreads = 256;
constexpr int kChrBins = 2048;      // LDS histogram covers refIDs < 1024; larger ones go straight to global

__device__ __forceinline__ uint32_t lower_bound_u32(const uint32_t* __restrict__ a, uint32_t lo, uint32_t hi, uint32_t key) {
    // first index in [lo,hi) with a[idx] >= key
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(kThreads) void count_kernel(CountTables T, BatchDev B) {
    __shared__ uint32_t s_chr[kChrBins];
    for (int k = threadIdx.x; k < kChrBins; k += kThreads) s_chr[k] = 0;
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lane_lt = (1ull << lane) - 1ull;
    uint32_t err = 0;
    const uint32_t stride = gridDim.x * kThreads;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < B.n_frag; i += stride) {
        const uint32_t chr = B.frag_chr[i];
        if (chr >= T.n_chr) { err |= ERR_CHR_RANGE; continue; }
        if (!T.chr_owned[chr]) continue;
        const uint32_t dir = B.frag_flags[i] & 1u;
        // ---- FragmentsInChr
        if (chr < kChrBins / 2) atomicAdd(&s_chr[chr * 2 + dir], 1u);
        else atomicAdd(&T.chr_cnt[chr * 2 + dir], 1u);
        // ---- FragmentsInROI: first ROI (by end) whose end >= fragment end; counted if it contains the fragment
        {
            const uint32_t r0 = T.roi_off[chr], r1 = T.roi_off[chr + 1];
            if (r1 > r0) {
                const uint32_t fs = B.frag_start[i], fe = B.frag_end[i];
                const uint32_t r = lower_bound_u32(T.roi_end, r0, r1, fe);
                if (r < r1 && T.roi_start[r] <= fs)
                    atomicAdd(&T.roi_cnt[(uint64_t)dir * T.n_roi + T.roi_idx[r]], 1u);
            }
        }
        // ---- per aligned block
        const uint64_t base = T.chr_base[chr];
        const uint32_t clen = T.chr_len[chr];
        int32_t* __restrict__ plane = T.diff + (uint64_t)dir * T.plane_stride + base;
        const uint32_t p0 = T.sp_off[chr], p1 = T.sp_off[chr + 1];
        uint32_t* __restrict__ sp_cnt = T.sp_cnt + (uint64_t)dir * T.n_sp;
        const uint32_t k0 = B.frag_blk_off[i], k1 = B.frag_blk_off[i + 1];
        uint32_t prev_end = 0, prev_len = 0, prev_read = 0xffu;
        for (uint32_t k = k0; k < k1; k++) {
            const uint32_t bs = B.blk_start[k], bl = B.blk_len[k], rd = B.blk_read[k];
            uint32_t be = bs + bl;
            if (be > clen || be < bs) { err |= ERR_BLOCK_RANGE; be = clen; }
            if (bs < be) {
                // CoverageBlocks / FragmentsMap: difference events
                __hip_atomic_fetch_add(plane + bs, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(plane + be, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // SpansPoint
                if (bl > T.sp_min_len && p1 > p0) {
                    const uint32_t lo = bs + T.sp_lo_add, hi = be - T.sp_hi_sub;
                    for (uint32_t p = lower_bound_u32(T.sp_pos, p0, p1, lo); p < p1 && T.sp_pos[p] <= hi; p++)
                        atomicAdd(&sp_cnt[p], 1u);
                }
            }
            // JunctionCount: gap between two consecutive blocks of the same read
            const bool emit = (rd == prev_read) && prev_len >= T.junc_min_flank && bl >= T.junc_min_flank;
            const uint64_t mask = __ballot(emit);
            if (emit) {
                const uint64_t gleft = base + prev_end;
                const uint32_t span = bs - prev_end;
                const int leader = __ffsll((long long)mask) - 1;
                unsigned long long slot = 0;
                if ((int)lane == leader) slot = atomicAdd(T.junc_count, (unsigned long long)__popcll(mask));
                slot = __shfl(slot, leader) + (unsigned long long)__popcll(mask & lane_lt);
                if (bs < prev_end || span >= (1u << kJuncSpanBits) || gleft >= kJuncMaxGlobal) err |= ERR_JUNC_RANGE;
                else if (slot >= T.junc_capacity) err |= ERR_JUNC_CAPACITY;
                else T.junc_keys[slot] = (gleft << kJuncLeftShift) | ((uint64_t)span << 1) | dir;
            }
            prev_end = bs + bl; prev_len = bl; prev_read = rd;
        }
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
    uint32_t blocks = (B.n_frag + kThreads - 1) / kThreads;
    if (blocks > 256u * 8u) blocks = 256u * 8u;
    hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(kThreads), 0, s, T, B);
}

}   // namespace dev
}   // namespace irf
