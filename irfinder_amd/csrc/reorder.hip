// reorder.hip — a batch in any order -> the same batch binned by coordinate (gfx950).
//
// The count kernel (count.hip) is built for coordinate-sorted input: a chunk of consecutive fragments shares one LDS
// window of the difference planes, and neighbouring chunks share table lines in one XCD's L2.  Name-collated input
// (the order irf_dreader_* delivers without the name join; upstream's own input order) scatters every chunk over the
// whole genome: seven times the kernel's time.  All outputs are commutative sums, so the engine may count
// the fragments in any order it likes: with irf_set_input_order(IRF_ORDER_ANY) every batch is binned by
// (chromosome base + start) >> kReorderShift with the radix passes of radix.hip, its columns are gathered into that
// order, and the count kernel runs on the copy.  No upstream counterpart.
//
// Shape.  A scattered 4-byte read costs a whole 64-byte sector, so everything the gather needs of a fragment sits in ONE
// 64-byte record that a coalesced pass builds first: the four columns, the block count, the first block's index and the
// first kReorderInline blocks themselves (a fragment has three blocks on average; the few with more than five fetch the
// rest from the caller's arrays).  The sort key carries the block count next to the bin (bits 32..37), so the new CSR
// offsets come from a coalesced pass over the sorted keys and one scan; the gather then reads one record per fragment
// and writes fragment columns and blocks in place.  Per fragment: 116 B streamed in, 72 B out, 8 B x 2 x passes sorted,
// one 64-byte random read.
#include "kernels.h"

namespace irf {
namespace dev {

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kReadShift = 30;          // packed block: y = len | read << 30
constexpr uint32_t kInline = 5;              // blocks inside a fragment's record

struct __attribute__((aligned(64))) FragRec { uint32_t chr, start, end, flags, nblk, off0; uint2 blk[kInline]; };
static_assert(sizeof(FragRec) == 64, "one sector per fragment");

// key = bin << kReorderKeyBinShift | min(nblk, kReorderKeyNblkMax) << 32 | fragment index
// A workgroup takes kThreads consecutive fragments.  Their blocks are one contiguous range of the block columns: it is
// loaded coalesced into LDS (checked and packed on the way), every thread picks its own blocks there, and the records
// leave through LDS as well, so that a wave's store covers 1 KiB without gaps (a thread storing its own 64 bytes in four
// pieces left every sector to be completed by three later instructions).  A range beyond the staging area (very long
// fragments, malformed offsets) is read per thread instead.
constexpr uint32_t kStageBlocks = 2048;
__global__ __launch_bounds__(kThreads) void reorder_keys_kernel(BatchDev B, const uint64_t* __restrict__ chr_base, uint32_t n_chr,
                                                                uint64_t* __restrict__ keys, FragRec* __restrict__ recs, uint32_t* err_flags) {
    __shared__ uint2 s_blk[kStageBlocks];
    __shared__ uint4 s_rec[kThreads * 4];
    __shared__ uint32_t s_range[2];
    const uint32_t tid = threadIdx.x;
    const uint32_t i0 = blockIdx.x * kThreads, i = i0 + tid;
    const bool in = i < B.n_frag;
    const uint32_t ic = in ? i : B.n_frag - 1u;
    const uint32_t c = B.frag_chr[ic], st = B.frag_start[ic], en = B.frag_end[ic], fl = B.frag_flags[ic];
    const uint32_t o0 = B.frag_blk_off[ic], o1 = B.frag_blk_off[ic + 1u];
    uint32_t err = 0;
    // ERR_CSR as the count kernel reports it in sorted mode: not non-decreasing, beyond n_blk, not spanning [0, n_blk];
    // a fragment with bad offsets gets no blocks, so nothing is read out of bounds
    const bool ok = o0 <= o1 && o1 <= B.n_blk;
    if (in && (!ok || (i == 0u && o0 != 0u) || (i == B.n_frag - 1u && o1 != B.n_blk))) err |= (uint32_t)ERR_CSR;
    const uint32_t nb = (in && ok) ? o1 - o0 : 0u;
    const uint32_t last = (B.n_frag - i0 < (uint32_t)kThreads ? B.n_frag - i0 : (uint32_t)kThreads) - 1u;
    if (tid == 0u) s_range[0] = o0 < B.n_blk ? o0 : B.n_blk;
    if (tid == last) s_range[1] = o1 < B.n_blk ? o1 : B.n_blk;
    wg_barrier();
    const uint32_t wb0 = s_range[0], wb1 = s_range[1];
    const bool staged = wb1 >= wb0 && wb1 - wb0 <= kStageBlocks;
    if (staged) {
        for (uint32_t k = wb0 + tid; k < wb1; k += kThreads) {
            const uint32_t len = B.blk_len[k], rd = B.blk_read[k];
            // (a block of 2^30 bases or a read index above 3 cannot be valid input: reported like a block beyond its chromosome)
            if (len >= (1u << kReadShift) || rd > 3u) err |= (uint32_t)ERR_BLOCK_RANGE;
            s_blk[k - wb0] = make_uint2(B.blk_start[k], (len & ((1u << kReadShift) - 1u)) | ((rd & 3u) << kReadShift));
        }
    }
    wg_barrier();
    uint2 blk[kInline];
#pragma unroll
    for (uint32_t k = 0; k < kInline; k++) blk[k] = make_uint2(0u, 0u);
    // (with well-formed offsets every fragment's range lies inside the workgroup's; anything else is read directly)
    if (staged && o0 >= wb0 && o1 <= wb1) {
#pragma unroll
        for (uint32_t k = 0; k < kInline; k++) if (k < nb) blk[k] = s_blk[o0 - wb0 + k];
    } else {
        for (uint32_t k = 0; k < nb; k++) {
            const uint32_t len = B.blk_len[o0 + k], rd = B.blk_read[o0 + k];
            if (len >= (1u << kReadShift) || rd > 3u) err |= (uint32_t)ERR_BLOCK_RANGE;
            const uint2 v = make_uint2(B.blk_start[o0 + k], (len & ((1u << kReadShift) - 1u)) | ((rd & 3u) << kReadShift));
#pragma unroll
            for (uint32_t q = 0; q < kInline; q++) if (q == k) blk[q] = v;         // (registers: no dynamic index)
        }
    }
    if (err) atomicOr(err_flags, err);
    if (in) {
        const uint64_t bin = c < n_chr ? (chr_base[c] + st) >> kReorderShift : 0ull;   // (a bad id is reported by the count kernel)
        keys[i] = bin << kReorderKeyBinShift | (uint64_t)(nb < kReorderKeyNblkMax ? nb : kReorderKeyNblkMax) << 32 | i;
    }
    s_rec[tid * 4u + 0u] = make_uint4(c, st, en, fl);
    s_rec[tid * 4u + 1u] = make_uint4(nb, o0, blk[0].x, blk[0].y);
    s_rec[tid * 4u + 2u] = make_uint4(blk[1].x, blk[1].y, blk[2].x, blk[2].y);
    s_rec[tid * 4u + 3u] = make_uint4(blk[3].x, blk[3].y, blk[4].x, blk[4].y);
    wg_barrier();
    uint4* dst = reinterpret_cast<uint4*>(recs + i0);
    const uint32_t n16 = (last + 1u) * 4u;
#pragma unroll
    for (uint32_t m = 0; m < 4u; m++) {
        const uint32_t q = m * kThreads + tid;
        if (q < n16) dst[q] = s_rec[q];
    }
}

// block count of the fragment now at j, off the sorted keys (the record only for the few whose count did not fit the key)
__global__ __launch_bounds__(kThreads) void reorder_counts_kernel(uint32_t n_frag, const uint64_t* __restrict__ sorted_keys, const FragRec* __restrict__ recs,
                                                                  uint32_t* __restrict__ nblk) {
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j > n_frag) return;
    if (j == n_frag) { nblk[j] = 0u; return; }           // (the scan's last element: the total lands behind it)
    const uint64_t key = sorted_keys[j];
    uint32_t nb = (uint32_t)(key >> 32) & kReorderKeyNblkMax;
    if (nb == kReorderKeyNblkMax) nb = recs[(uint32_t)key].nblk;
    nblk[j] = nb;
}

// Fragment columns and blocks in the new order: one 64-byte record read per fragment.  Offsets of a malformed
// device-resident batch (ranges that overlap: every one of them inside [0, n_blk], their sum beyond it, or wrapped past
// 2^32) were reported as ERR_CSR by reorder_keys_kernel; here they must only stay inside the arrays: reads are inside
// [0, n_blk) because off0 + nblk <= n_blk held per fragment, writes are cut at n_blk.
__global__ __launch_bounds__(kThreads) void reorder_gather_kernel(BatchDev B, const FragRec* __restrict__ recs, const uint64_t* __restrict__ sorted_keys,
                                                                  const uint32_t* __restrict__ new_off, uint32_t* __restrict__ chr, uint8_t* __restrict__ flags,
                                                                  uint32_t* __restrict__ start, uint32_t* __restrict__ end, uint32_t* __restrict__ bs,
                                                                  uint32_t* __restrict__ bl, uint8_t* __restrict__ rd) {
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= B.n_frag) return;
    const uint32_t idx = (uint32_t)sorted_keys[j];
    const uint4* src = reinterpret_cast<const uint4*>(recs + idx);
    const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
    chr[j] = a.x; start[j] = a.y; end[j] = a.z; flags[j] = (uint8_t)a.w;
    const uint32_t dst = new_off[j], cnt = new_off[j + 1u] - dst, off0 = b.y;
    const uint2 in[kInline] = {make_uint2(b.z, b.w), make_uint2(c.x, c.y), make_uint2(c.z, c.w), make_uint2(d.x, d.y), make_uint2(d.z, d.w)};
#pragma unroll
    for (uint32_t k = 0; k < kInline; k++) {
        if (k < cnt && (uint64_t)dst + k < B.n_blk) {
            bs[dst + k] = in[k].x;
            bl[dst + k] = in[k].y & ((1u << kReadShift) - 1u);
            rd[dst + k] = (uint8_t)(in[k].y >> kReadShift);
        }
    }
    for (uint32_t k = kInline; k < cnt; k++) {           // (rare: the rest comes from the caller's arrays)
        if ((uint64_t)dst + k >= B.n_blk || (uint64_t)off0 + k >= B.n_blk) break;
        bs[dst + k] = B.blk_start[off0 + k];
        bl[dst + k] = B.blk_len[off0 + k] & ((1u << kReadShift) - 1u);
        rd[dst + k] = (uint8_t)(B.blk_read[off0 + k] & 3u);
    }
}

}   // namespace

size_t reorder_frag_rec_bytes() { return sizeof(FragRec); }

void launch_reorder_keys(const BatchDev& B, const uint64_t* chr_base, uint32_t n_chr, uint64_t* keys, void* frag_recs, uint32_t* err_flags, hipStream_t s) {
    hipLaunchKernelGGL(reorder_keys_kernel, dim3((B.n_frag + kThreads - 1) / kThreads), dim3(kThreads), 0, s, B, chr_base, n_chr, keys,
                       static_cast<FragRec*>(frag_recs), err_flags);
}

void launch_reorder_counts(uint32_t n_frag, const uint64_t* sorted_keys, const void* frag_recs, uint32_t* nblk, hipStream_t s) {
    hipLaunchKernelGGL(reorder_counts_kernel, dim3(n_frag / kThreads + 1u), dim3(kThreads), 0, s, n_frag, sorted_keys, static_cast<const FragRec*>(frag_recs), nblk);
}

void launch_reorder_gather(const BatchDev& B, const void* frag_recs, const uint64_t* sorted_keys, const uint32_t* new_off, uint32_t* chr, uint8_t* flags,
                           uint32_t* start, uint32_t* end, uint32_t* bs, uint32_t* bl, uint8_t* rd, hipStream_t s) {
    hipLaunchKernelGGL(reorder_gather_kernel, dim3((B.n_frag + kThreads - 1) / kThreads), dim3(kThreads), 0, s, B, static_cast<const FragRec*>(frag_recs), sorted_keys,
                       new_off, chr, flags, start, end, bs, bl, rd);
}

}   // namespace dev
}   // namespace irf
