// reorder.hip — a batch in any order -> the same batch binned by coordinate (gfx950).
//
// The count kernel (count.hip) is built for coordinate-sorted input: a chunk of 1024 consecutive fragments shares one
// LDS window of the difference planes, and neighbouring chunks share table lines in one XCD's L2.  Name-collated input
// (the order irf_dreader_* delivers without the name join; upstream's own input order) scatters every chunk over the
// whole genome: seven times the kernel's time.  All outputs are commutative sums, so the engine may count
// the fragments in any order it likes: with irf_set_input_order(IRF_ORDER_ANY) every batch is binned by
// (chromosome base + start) >> kReorderShift with the radix passes of radix.hip (key = bin << 32 | fragment index),
// its columns are gathered into that order, and the count kernel runs on the copy.  No upstream counterpart.
#include "kernels.h"

namespace irf {
namespace dev {

namespace {

constexpr int kThreads = 256;

// A scattered 4-byte read costs a whole 64-byte sector.  The two gathers therefore read array-of-structs copies that two
// coalesced passes build first: one 32-byte record per fragment (the four columns, its first block, its block count, with
// the CSR check of a device-resident batch: neighbours are adjacent here) and one 8-byte record per block.
struct __attribute__((aligned(32))) FragRec { uint32_t chr, start, end, off0, nblk, flags, pad0, pad1; };
constexpr uint32_t kReadShift = 30;          // BlkRec8.y = len | read << 30

__global__ __launch_bounds__(kThreads) void reorder_keys_kernel(BatchDev B, const uint64_t* __restrict__ chr_base, uint32_t n_chr,
                                                                uint64_t* __restrict__ keys, FragRec* __restrict__ recs, uint32_t* err_flags) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= B.n_frag) return;
    const uint32_t c = B.frag_chr[i], st = B.frag_start[i];
    const uint64_t bin = c < n_chr ? (chr_base[c] + st) >> kReorderShift : 0ull;      // (a bad id is reported by the count kernel)
    keys[i] = bin << 32 | i;
    const uint32_t o0 = B.frag_blk_off[i], o1 = B.frag_blk_off[i + 1u];
    // ERR_CSR as the count kernel reports it in sorted mode: not non-decreasing, beyond n_blk, not spanning [0, n_blk];
    // a fragment with bad offsets gets no blocks, so nothing is read out of bounds
    const bool ok = o0 <= o1 && o1 <= B.n_blk;
    if (!ok || (i == 0u && o0 != 0u) || (i == B.n_frag - 1u && o1 != B.n_blk)) atomicOr(err_flags, (uint32_t)ERR_CSR);
    FragRec r;
    r.chr = c; r.start = st; r.end = B.frag_end[i]; r.off0 = o0; r.nblk = ok ? o1 - o0 : 0u; r.flags = B.frag_flags[i]; r.pad0 = 0u; r.pad1 = 0u;
    recs[i] = r;
}

__global__ __launch_bounds__(kThreads) void reorder_pack_blocks_kernel(BatchDev B, uint2* __restrict__ pk, uint32_t* err_flags) {
    const uint32_t b = blockIdx.x * kThreads + threadIdx.x;
    if (b >= B.n_blk) return;
    const uint32_t len = B.blk_len[b], rd = B.blk_read[b];
    // (a block of 2^30 bases or a read index above 3 cannot be valid input: reported like a block beyond its chromosome)
    if (len >= (1u << kReadShift) || rd > 3u) atomicOr(err_flags, (uint32_t)ERR_BLOCK_RANGE);
    pk[b] = make_uint2(B.blk_start[b], (len & ((1u << kReadShift) - 1u)) | (rd << kReadShift));
}

// fragment columns in the new order, the block count of every fragment and where its blocks lie in the old order
__global__ __launch_bounds__(kThreads) void reorder_frags_kernel(uint32_t n_frag, const FragRec* __restrict__ recs, const uint64_t* __restrict__ sorted_keys,
                                                                 uint32_t* __restrict__ chr, uint8_t* __restrict__ flags, uint32_t* __restrict__ start,
                                                                 uint32_t* __restrict__ end, uint32_t* __restrict__ nblk, uint32_t* __restrict__ src_off) {
    const uint32_t j = blockIdx.x * kThreads + threadIdx.x;
    if (j > n_frag) return;
    if (j == n_frag) { nblk[j] = 0u; return; }           // (the scan's last element: the total lands behind it)
    const FragRec r = recs[(uint32_t)sorted_keys[j]];
    chr[j] = r.chr;
    flags[j] = (uint8_t)r.flags;
    start[j] = r.start;
    end[j] = r.end;
    nblk[j] = r.nblk;
    src_off[j] = r.off0;
}

// the blocks of the fragment now at j move to new_off[j]; four lanes per fragment (fragments have a handful of blocks).
// Offsets of a malformed device-resident batch (ranges that overlap: every one of them inside [0, n_blk], their sum beyond
// it, or wrapped past 2^32) were reported as ERR_CSR by reorder_keys_kernel; here they must only stay inside the arrays:
// reads are inside [0, n_blk) because off0 + nblk <= n_blk held per fragment, writes are cut at n_blk.
__global__ __launch_bounds__(kThreads) void reorder_blocks_kernel(uint32_t n_frag, uint32_t n_blk, const uint2* __restrict__ pk, const uint32_t* __restrict__ src_off,
                                                                  const uint32_t* __restrict__ new_off, uint32_t* __restrict__ bs,
                                                                  uint32_t* __restrict__ bl, uint8_t* __restrict__ rd) {
    const uint64_t t = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    const uint64_t j = t >> 2;
    if (j >= n_frag) return;
    const uint32_t src = src_off[j];
    const uint32_t dst = new_off[j], cnt = new_off[j + 1u] - dst;
    for (uint32_t k = (uint32_t)(t & 3u); k < cnt; k += 4u) {
        if ((uint64_t)dst + k >= n_blk || (uint64_t)src + k >= n_blk) break;
        const uint2 v = pk[src + k];
        bs[dst + k] = v.x;
        bl[dst + k] = v.y & ((1u << kReadShift) - 1u);
        rd[dst + k] = (uint8_t)(v.y >> kReadShift);
    }
}

}   // namespace

size_t reorder_frag_rec_bytes() { return sizeof(FragRec); }

void launch_reorder_keys(const BatchDev& B, const uint64_t* chr_base, uint32_t n_chr, uint64_t* keys, void* frag_recs, void* blk_recs,
                         uint32_t* err_flags, hipStream_t s) {
    hipLaunchKernelGGL(reorder_keys_kernel, dim3((B.n_frag + kThreads - 1) / kThreads), dim3(kThreads), 0, s, B, chr_base, n_chr, keys,
                       static_cast<FragRec*>(frag_recs), err_flags);
    if (B.n_blk) hipLaunchKernelGGL(reorder_pack_blocks_kernel, dim3((B.n_blk + kThreads - 1) / kThreads), dim3(kThreads), 0, s, B, static_cast<uint2*>(blk_recs), err_flags);
}

void launch_reorder_frags(uint32_t n_frag, const void* frag_recs, const uint64_t* sorted_keys, uint32_t* chr, uint8_t* flags, uint32_t* start, uint32_t* end,
                          uint32_t* nblk, uint32_t* src_off, hipStream_t s) {
    hipLaunchKernelGGL(reorder_frags_kernel, dim3(n_frag / kThreads + 1u), dim3(kThreads), 0, s, n_frag, static_cast<const FragRec*>(frag_recs), sorted_keys, chr, flags,
                       start, end, nblk, src_off);
}

void launch_reorder_blocks(uint32_t n_frag, uint32_t n_blk, const void* blk_recs, const uint32_t* src_off, const uint32_t* new_off, uint32_t* bs, uint32_t* bl, uint8_t* rd,
                           hipStream_t s) {
    const uint64_t threads = (uint64_t)n_frag * 4u;
    hipLaunchKernelGGL(reorder_blocks_kernel, dim3((uint32_t)((threads + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, n_frag, n_blk,
                       static_cast<const uint2*>(blk_recs), src_off, new_off, bs, bl, rd);
}

}   // namespace dev
}   // namespace irf
