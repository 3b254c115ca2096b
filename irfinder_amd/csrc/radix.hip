// radix.hip — junction sort-reduce (gfx950).
//
// Replaces the three std::map<…> updates per junction of ⟨upstream⟩
// ReadBlockProcessor.cpp JunctionCount::ProcessBlocks (RB-tree insert per
// event) with: append packed 64-bit (chr+left, span, strand) events — weighted:
// the count kernel merges a chunk's events on one junction in LDS first — during
// counting, then one stable LSD radix sort (8-bit digits, constant digits
// skipped) + reduction (sum of the weights of every run) at finalize — for NOVEL junctions; reads on a
// junction listed in ref-sj.ref (the vast majority) are counted in place by the
// count kernel and never enter the sort.  juncLeft / juncRight are not
// materialised as maps: their only consumers are the lookups at intron Start /
// End (CoverageBlocksIRFinder::WriteOutput), so the unique junctions are
// scattered once into per-intron-boundary counters (launch_junction_lookups for
// the novel table, launch_known_lookups for the known one, which also takes the
// Directional() vote).
#include "kernels.h"

namespace irf {
namespace dev {

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kKeysPerThread = kRadixTile / kThreads;      // 16
constexpr int kWaveChunk = kRadixTile / kWaves;            // 1024 keys per wave, 16 iterations of 64

__global__ __launch_bounds__(kThreads) void key_bits_kernel(const uint64_t* __restrict__ keys, uint64_t n,
                                                            unsigned long long* out) {
    __shared__ unsigned long long s_o[kWaves], s_a[kWaves];
    unsigned long long o = 0ull, a = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) {
        const unsigned long long k = keys[i];
        o |= k; a &= k;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { o |= __shfl_xor(o, d); a &= __shfl_xor(a, d); }
    if ((threadIdx.x & 63) == 0) { s_o[threadIdx.x >> 6] = o; s_a[threadIdx.x >> 6] = a; }
    wg_barrier();
    if (threadIdx.x == 0) {          // one pair of atomics per workgroup (they all hit the same two words)
        for (int w = 1; w < kWaves; w++) { o |= s_o[w]; a &= s_a[w]; }
        atomicOr(&out[0], o); atomicAnd(&out[1], a);
    }
}

// cnt[digit * n_tiles + tile] = number of keys of the tile with that digit
__global__ __launch_bounds__(kThreads) void radix_hist_kernel(const uint64_t* __restrict__ keys, uint64_t n, int shift,
                                                              uint32_t n_tiles, uint32_t* __restrict__ cnt) {
    __shared__ uint32_t s_h[256];
    s_h[threadIdx.x] = 0;
    wg_barrier();
    const uint64_t tile0 = (uint64_t)blockIdx.x * kRadixTile;
#pragma unroll
    for (int j = 0; j < kKeysPerThread; j++) {
        const uint64_t i = tile0 + (uint64_t)j * kThreads + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
    }
    wg_barrier();
    cnt[(uint64_t)threadIdx.x * n_tiles + blockIdx.x] = s_h[threadIdx.x];
}

// Stable scatter.  Order inside a tile is (wave, iteration, lane) = ascending index.
__global__ __launch_bounds__(kThreads) void radix_scatter_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                                 const uint32_t* __restrict__ win, uint32_t* __restrict__ wout,
                                                                 uint64_t n, int shift, uint32_t n_tiles,
                                                                 const uint32_t* __restrict__ off) {
    __shared__ uint32_t s_cnt[kWaves][256];
    for (int k = threadIdx.x; k < kWaves * 256; k += kThreads) (&s_cnt[0][0])[k] = 0;
    wg_barrier();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t lane_lt = (1ull << lane) - 1ull;
    const uint64_t base = (uint64_t)blockIdx.x * kRadixTile + (uint64_t)wave * kWaveChunk + lane;
    uint64_t key[kKeysPerThread];
    uint32_t rank[kKeysPerThread], wgt[kKeysPerThread];
#pragma unroll
    for (int it = 0; it < kKeysPerThread; it++) {
        const uint64_t i = base + (uint64_t)it * 64u;
        const bool valid = i < n;
        key[it] = valid ? in[i] : 0ull;
        wgt[it] = (valid && win) ? win[i] : 0u;           // (win == nullptr: keys only)
        const uint32_t d = (uint32_t)(key[it] >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        rank[it] = 0;
        if (valid) {
            const uint32_t old = s_cnt[wave][d];
            rank[it] = old + (uint32_t)__popcll(peers & lane_lt);
            if ((peers & lane_lt) == 0ull) s_cnt[wave][d] = old + (uint32_t)__popcll(peers);
        }
    }
    wg_barrier();
    {   // digit = threadIdx.x: turn the per-wave counts into global start offsets
        uint32_t run = off[(uint64_t)threadIdx.x * n_tiles + blockIdx.x];
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const uint32_t c = s_cnt[w][threadIdx.x];
            s_cnt[w][threadIdx.x] = run;
            run += c;
        }
    }
    wg_barrier();
#pragma unroll
    for (int it = 0; it < kKeysPerThread; it++) {
        const uint64_t i = base + (uint64_t)it * 64u;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[it] >> shift) & 255u;
            const uint32_t o = s_cnt[wave][d] + rank[it];
            out[o] = key[it];
            if (wout) wout[o] = wgt[it];
        }
    }
}

__global__ __launch_bounds__(kThreads) void head_flags_kernel(const uint64_t* __restrict__ keys, uint64_t n,
                                                              uint32_t* __restrict__ flags) {
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(kThreads) void unique_runs_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ flags,
                                                               const uint32_t* __restrict__ uidx, uint64_t n,
                                                               uint64_t* __restrict__ ukeys, uint32_t* __restrict__ ustart) {
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i < n && flags[i]) {
        const uint32_t u = uidx[i];
        ukeys[u] = keys[i];
        ustart[u] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(kThreads) void run_sums_kernel(const uint32_t* __restrict__ ustart, uint64_t nu, uint64_t n,
                                                            const uint32_t* __restrict__ wex, const uint32_t* __restrict__ w,
                                                            uint32_t* __restrict__ ucount) {
    const uint64_t u = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (u >= nu) return;
    const uint32_t end = u + 1 < nu ? wex[ustart[u + 1]] : wex[n - 1] + w[n - 1];
    ucount[u] = end - wex[ustart[u]];
}

__device__ __forceinline__ int64_t find_u64(const uint64_t* __restrict__ a, uint32_t n, uint64_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < n && a[lo] == key) ? (int64_t)lo : -1;
}

// novel unique junctions -> SpliceLeft / SpliceRight counters of the intron boundaries they touch
__global__ __launch_bounds__(kThreads) void junction_lookups_kernel(JuncLookupTables J) {
    const uint64_t u = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (u >= J.n_unique) return;
    const uint64_t key = J.ukeys[u];
    const uint32_t cnt = J.ucount[u];
    const uint32_t dir = (uint32_t)(key & 1ull);
    const uint64_t gleft = key >> kJuncLeftShift;
    const uint64_t span = (key >> 1) & ((1ull << kJuncSpanBits) - 1ull);
    const int64_t a = find_u64(J.ustart_pos, J.n_ustart, gleft);
    if (a >= 0) atomicAdd(&J.left_cnt[(uint64_t)dir * J.n_ustart + (uint64_t)a], cnt);
    const int64_t b = find_u64(J.uend_pos, J.n_uend, gleft + span);
    if (b >= 0) atomicAdd(&J.right_cnt[(uint64_t)dir * J.n_uend + (uint64_t)b], cnt);
}

// known junctions (counted in place by the count kernel) -> the same counters, plus the strandedness vote of
// JunctionCount::Directional(): reads on a junction of known strand agree ("same") or disagree ("opposite") with it
__global__ __launch_bounds__(kThreads) void known_lookups_kernel(JuncLookupTables J) {
    unsigned long long same = 0, opp = 0;
    const uint32_t k = blockIdx.x * kThreads + threadIdx.x;
    if (k < J.n_known) {
        const uint32_t row = J.known_kjidx[k];
        const uint32_t c0 = J.known_cnt[row], c1 = J.known_cnt[(uint64_t)J.n_kj + row];
        if (c0 | c1) {
            // where the junction's ends sit among the intron starts / ends was looked up by the host (static)
            const int64_t a = J.known_us_idx[k];
            if (a >= 0) {
                if (c0) atomicAdd(&J.left_cnt[(uint64_t)a], c0);
                if (c1) atomicAdd(&J.left_cnt[(uint64_t)J.n_ustart + (uint64_t)a], c1);
            }
            const int64_t b = J.known_ue_idx[k];
            if (b >= 0) {
                if (c0) atomicAdd(&J.right_cnt[(uint64_t)b], c0);
                if (c1) atomicAdd(&J.right_cnt[(uint64_t)J.n_uend + (uint64_t)b], c1);
            }
            const uint32_t kd = J.known_dir[k];
            if (kd == 1u) { same += c0; opp += c1; }
            else if (kd == 2u) { same += c1; opp += c0; }
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { same += __shfl_xor(same, d); opp += __shfl_xor(opp, d); }
    if ((threadIdx.x & 63) == 0 && J.dir_votes) {
        if (same) atomicAdd(&J.dir_votes[0], same);
        if (opp) atomicAdd(&J.dir_votes[1], opp);
    }
}

inline uint32_t blocks_for(uint64_t n) { return (uint32_t)((n + kThreads - 1) / kThreads); }

}   // namespace

uint64_t radix_workspace_elems(uint64_t n) {
    const uint64_t n_tiles = (n + kRadixTile - 1) / kRadixTile;
    const uint64_t c = 256 * n_tiles;
    return c + scan_workspace_elems(c) + 16;
}

void launch_key_bits(const uint64_t* keys, uint64_t n, unsigned long long* out, hipStream_t s) {
    if (n == 0) return;
    uint32_t blocks = blocks_for(n);
    if (blocks > 1024u) blocks = 1024u;
    hipLaunchKernelGGL(key_bits_kernel, dim3(blocks), dim3(kThreads), 0, s, keys, n, out);
}

void launch_radix_pass(const uint64_t* in, uint64_t* out, const uint32_t* win, uint32_t* wout, uint64_t n, int shift, uint32_t* ws,
                       hipStream_t s) {
    if (n == 0) return;
    const uint32_t n_tiles = (uint32_t)((n + kRadixTile - 1) / kRadixTile);
    uint32_t* cnt = ws;
    uint32_t* scan_ws = ws + (uint64_t)256 * n_tiles;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(n_tiles), dim3(kThreads), 0, s, in, n, shift, n_tiles, cnt);
    launch_exclusive_scan_u32(cnt, cnt, (uint64_t)256 * n_tiles, scan_ws, s);
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(n_tiles), dim3(kThreads), 0, s, in, out, win, wout, n, shift, n_tiles, (const uint32_t*)cnt);
}

void launch_head_flags(const uint64_t* keys, uint64_t n, uint32_t* flags, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(head_flags_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, s, keys, n, flags);
}

void launch_unique_runs(const uint64_t* keys, const uint32_t* flags, const uint32_t* uidx, uint64_t n, uint64_t* ukeys,
                        uint32_t* ustart, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(unique_runs_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, s, keys, flags, uidx, n, ukeys, ustart);
}

void launch_run_sums(const uint32_t* ustart, uint64_t nu, uint64_t n, const uint32_t* wex, const uint32_t* w, uint32_t* ucount,
                     hipStream_t s) {
    if (nu == 0) return;
    hipLaunchKernelGGL(run_sums_kernel, dim3(blocks_for(nu)), dim3(kThreads), 0, s, ustart, nu, n, wex, w, ucount);
}

void launch_junction_lookups(const JuncLookupTables& J, hipStream_t s) {
    if (J.n_unique == 0) return;
    hipLaunchKernelGGL(junction_lookups_kernel, dim3(blocks_for(J.n_unique)), dim3(kThreads), 0, s, J);
}

void launch_known_lookups(const JuncLookupTables& J, hipStream_t s) {
    if (J.n_known == 0) return;
    hipLaunchKernelGGL(known_lookups_kernel, dim3(blocks_for(J.n_known)), dim3(kThreads), 0, s, J);
}

}   // namespace dev
}   // namespace irf
