// reorder.hip — a batch in any order -> the same batch binned by coordinate (gfx950).
//
// The count kernel (count.hip) is built for coordinate-sorted input: a chunk of 1024 consecutive fragments shares one
// LDS window of the difference planes, and neighbouring chunks share table lines in one XCD's L2.  Name-collated input
// (the order irf_dreader_* delivers without the name join; upstream's own input order) scatters every chunk over the
// whole genome: seven times the kernel's time.  All outputs are commutative sums, so the engine may count
// the fragments in any order it likes: with irf_set_input_order(IRF_ORDER_ANY) every batch is binned by
// (chromosome base + start) >> kReorderShift with the radix passes of radix.hip (key = bin, payload = fragment index),
// its columns are gathered into that order, and the count kernel runs on the copy.  No upstream counterpart.
#include "kernels.h"

namespace irf {
namespace dev {

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void reorder_keys_kernel(const uint32_t* __restrict__ chr, const uint32_t* __restrict__ start,
                                                                const uint64_t* __restrict__ chr_base, uint32_t n_chr, uint32_t n,
                                                                uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    const uint32_t c = chr[i];
    keys[i] = c < n_chr ? (chr_base[c] + start[i]) >> kReorderShift : 0ull;      // (a bad id is reported by the count kernel)
    idx[i] = i;
}

// fragment columns in the new order + the block count of every fragment.  The CSR offsets of a device-resident batch are
// checked here as the count kernel checks them in sorted mode (ERR_CSR: not non-decreasing, beyond n_blk, not spanning
// [0, n_blk]); a fragment with bad offsets gets no blocks, so nothing is read out of bounds.
__global__ __launch_bounds__(kThreads) void reorder_frags_kernel(BatchDev B, const uint32_t* __restrict__ perm, uint32_t* __restrict__ chr,
                                                                 uint8_t* __restrict__ flags, uint32_t* __restrict__ start,
                                                                 uint32_t* __restrict__ end, uint32_t* __restrict__ nblk, uint32_t* err_flags) {
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j > B.n_frag) return;
    if (j == B.n_frag) { nblk[j] = 0u; return; }           // (the scan's last element: the total lands behind it)
    const uint32_t i = perm[j];
    const uint32_t o0 = B.frag_blk_off[i], o1 = B.frag_blk_off[i + 1u];
    chr[j] = B.frag_chr[i];
    flags[j] = B.frag_flags[i];
    start[j] = B.frag_start[i];
    end[j] = B.frag_end[i];
    const bool ok = o0 <= o1 && o1 <= B.n_blk;
    nblk[j] = ok ? o1 - o0 : 0u;
    if (!ok || (i == 0u && o0 != 0u) || (i == B.n_frag - 1u && o1 != B.n_blk)) atomicOr(err_flags, (uint32_t)ERR_CSR);
}

// the blocks of fragment perm[j] move to new_off[j]; four lanes per fragment (fragments have a handful of blocks)
__global__ __launch_bounds__(kThreads) void reorder_blocks_kernel(BatchDev B, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ new_off,
                                                                  uint32_t* __restrict__ bs, uint32_t* __restrict__ bl, uint8_t* __restrict__ rd) {
    const uint64_t t = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const uint64_t j = t >> 2;
    if (j >= B.n_frag) return;
    const uint32_t src = B.frag_blk_off[perm[j]];
    const uint32_t dst = new_off[j], cnt = new_off[j + 1u] - dst;
    for (uint32_t k = (uint32_t)(t & 3u); k < cnt; k += 4u) {
        bs[dst + k] = B.blk_start[src + k];
        bl[dst + k] = B.blk_len[src + k];
        rd[dst + k] = B.blk_read[src + k];
    }
}

}   // namespace

void launch_reorder_keys(const BatchDev& B, const uint64_t* chr_base, uint32_t n_chr, uint64_t* keys, uint32_t* idx, hipStream_t s) {
    hipLaunchKernelGGL(reorder_keys_kernel, dim3((B.n_frag + kThreads - 1) / kThreads), dim3(kThreads), 0, s, B.frag_chr, B.frag_start, chr_base, n_chr, B.n_frag, keys, idx);
}

void launch_reorder_frags(const BatchDev& B, const uint32_t* perm, uint32_t* chr, uint8_t* flags, uint32_t* start, uint32_t* end, uint32_t* nblk, uint32_t* err_flags, hipStream_t s) {
    hipLaunchKernelGGL(reorder_frags_kernel, dim3(B.n_frag / kThreads + 1u), dim3(kThreads), 0, s, B, perm, chr, flags, start, end, nblk, err_flags);
}

void launch_reorder_blocks(const BatchDev& B, const uint32_t* perm, const uint32_t* new_off, uint32_t* bs, uint32_t* bl, uint8_t* rd, hipStream_t s) {
    const uint64_t threads = (uint64_t)B.n_frag * 4u;
    hipLaunchKernelGGL(reorder_blocks_kernel, dim3((uint32_t)((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, B, perm, new_off, bs, bl, rd);
}

}   // namespace dev
}   // namespace irf
