// scan.hip — device-wide prefix sums (gfx950).
//
// The per-strand difference arrays written by count.hip become per-base depth
// through a prefix sum over the whole plane pair: every aligned
// block contributes +1 and -1 inside its own chromosome, so each chromosome's
// events sum to zero and the flat scan over the concatenated planes equals the
// segmented (per-chromosome, per-strand) scan.  This is what turns upstream's
// CoverageBlock::RecordCover event lists (⟨upstream⟩ CoverageBlock.cpp) into
// depth.  The planes are read ONCE here (segment_sums_kernel: 4 B per base, one
// sum per 256 bases), the 256x smaller array of segment sums is scanned, and
// the statistics kernel (stats.hip) rebuilds depth inside each segment on the
// fly — the scanned planes are never written back (that would cost another
// 4 B read + 4 B write per base).  The generic reduce-then-scan below (tiles of
// 4096) scans the segment sums and serves the radix sort and the junction
// compaction.
#include "kernels.h"

namespace irf {
namespace dev {

namespace {

constexpr int kThreads = 256;
constexpr int kRows = 4;                 // tile = kRows x (256 threads x 4 elements)
constexpr uint32_t kRowElems = kThreads * 4;
static_assert(kRows * kRowElems == kScanTile, "tile geometry");

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if ((int)lane >= d) v += o;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// partial[tile] = sum of the tile
__global__ __launch_bounds__(kThreads) void reduce_tiles_kernel(const uint32_t* __restrict__ in, uint64_t n,
                                                                uint32_t* __restrict__ partial) {
    __shared__ uint32_t s_w[kThreads / 64];
    const uint64_t tile0 = (uint64_t)blockIdx.x * kScanTile;
    uint32_t acc = 0;
    if (tile0 + kScanTile <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(in + tile0);
#pragma unroll
        for (int r = 0; r < kRows; r++) {
            uint4 v = p[r * kThreads + threadIdx.x];
            acc += v.x + v.y + v.z + v.w;
        }
    } else {
        for (uint64_t i = tile0 + threadIdx.x; i < n; i += kThreads) acc += in[i];
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    wg_barrier();
    if (threadIdx.x == 0) partial[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// out[i] = offset(tile) + prefix within the tile (inclusive or exclusive); out may alias in
template <bool INCLUSIVE>
__global__ __launch_bounds__(kThreads) void scan_tiles_kernel(const uint32_t* in, uint32_t* out, uint64_t n,
                                                              const uint32_t* __restrict__ tile_offset) {
    __shared__ uint32_t s_tot[kRows][kThreads / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t tile0 = (uint64_t)blockIdx.x * kScanTile;
    const bool full = tile0 + kScanTile <= n;
    uint4 v[kRows];
    if (full) {
        const uint4* p = reinterpret_cast<const uint4*>(in + tile0);
#pragma unroll
        for (int r = 0; r < kRows; r++) v[r] = p[r * kThreads + threadIdx.x];
    } else {
#pragma unroll
        for (int r = 0; r < kRows; r++) {
            const uint64_t e = tile0 + (uint64_t)r * kRowElems + threadIdx.x * 4u;
            v[r].x = e + 0 < n ? in[e + 0] : 0u;
            v[r].y = e + 1 < n ? in[e + 1] : 0u;
            v[r].z = e + 2 < n ? in[e + 2] : 0u;
            v[r].w = e + 3 < n ? in[e + 3] : 0u;
        }
    }
    uint32_t incl[kRows];
#pragma unroll
    for (int r = 0; r < kRows; r++) {
        incl[r] = wave_inclusive_scan(v[r].x + v[r].y + v[r].z + v[r].w, lane);
        if (lane == 63) s_tot[r][wave] = incl[r];
    }
    wg_barrier();
    const uint32_t base = tile_offset ? tile_offset[blockIdx.x] : 0u;
#pragma unroll
    for (int r = 0; r < kRows; r++) {
        // everything before (row r, this wave) in (row, wave) order
        uint32_t before = base;
#pragma unroll
        for (int rr = 0; rr < kRows; rr++)
#pragma unroll
            for (int w = 0; w < kThreads / 64; w++)
                if (rr < r || (rr == r && w < (int)wave)) before += s_tot[rr][w];
        uint32_t run = before + incl[r] - (v[r].x + v[r].y + v[r].z + v[r].w);   // exclusive prefix of this thread's 4
        uint4 o;
        if (INCLUSIVE) {
            o.x = run + v[r].x; o.y = o.x + v[r].y; o.z = o.y + v[r].z; o.w = o.z + v[r].w;
        } else {
            o.x = run; o.y = o.x + v[r].x; o.z = o.y + v[r].y; o.w = o.z + v[r].z;
        }
        if (full) {
            reinterpret_cast<uint4*>(out + tile0)[r * kThreads + threadIdx.x] = o;
        } else {
            const uint64_t e = tile0 + (uint64_t)r * kRowElems + threadIdx.x * 4u;
            if (e + 0 < n) out[e + 0] = o.x;
            if (e + 1 < n) out[e + 1] = o.y;
            if (e + 2 < n) out[e + 2] = o.z;
            if (e + 3 < n) out[e + 3] = o.w;
        }
    }
}

// One 256-position segment per (row, wave): the 64 lanes of a wave load 64 x uint4 = 1 KiB contiguous.
// seg_sums[tile * 16 + row * 4 + wave] = sum of that segment.  Reads 4 B per base, writes 4 B per 256 bases.
__global__ __launch_bounds__(kThreads) void segment_sums_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ seg_sums) {
    const uint64_t tile0 = (uint64_t)blockIdx.x * kScanTile;
    const uint4* p = reinterpret_cast<const uint4*>(in + tile0);
    uint4 v[kRows];
#pragma unroll
    for (int r = 0; r < kRows; r++) v[r] = p[r * kThreads + threadIdx.x];
#pragma unroll
    for (int r = 0; r < kRows; r++) {
        const uint32_t s = wave_sum(v[r].x + v[r].y + v[r].z + v[r].w);
        if ((threadIdx.x & 63) == 0) seg_sums[(uint64_t)blockIdx.x * (kScanTile / 256) + r * (kThreads / 64) + (threadIdx.x >> 6)] = s;
    }
}

inline uint64_t tiles_of(uint64_t n) { return (n + kScanTile - 1) / kScanTile; }

// exclusive scan of `in` (n elements) into `out`, using ws for the tile partials of every level
void exclusive_scan_rec(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* ws, hipStream_t s) {
    const uint64_t nt = tiles_of(n);
    if (nt <= 1) {
        hipLaunchKernelGGL(scan_tiles_kernel<false>, dim3(1), dim3(kThreads), 0, s, in, out, n, (const uint32_t*)nullptr);
        return;
    }
    hipLaunchKernelGGL(reduce_tiles_kernel, dim3((uint32_t)nt), dim3(kThreads), 0, s, in, n, ws);
    exclusive_scan_rec(ws, ws, nt, ws + nt, s);
    hipLaunchKernelGGL(scan_tiles_kernel<false>, dim3((uint32_t)nt), dim3(kThreads), 0, s, in, out, n, (const uint32_t*)ws);
}

}   // namespace

uint64_t scan_workspace_elems(uint64_t n) {
    uint64_t total = 0;
    while (true) {
        uint64_t nt = tiles_of(n);
        if (nt <= 1) break;
        total += nt;
        n = nt;
    }
    return total + 16;
}

void launch_segment_sums(const int32_t* diff, uint64_t n, uint32_t* seg_sums, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(segment_sums_kernel, dim3((uint32_t)(n / kScanTile)), dim3(kThreads), 0, s,
                       reinterpret_cast<const uint32_t*>(diff), seg_sums);
}

void launch_exclusive_scan_u32(const uint32_t* in, uint32_t* out, uint64_t n, uint32_t* ws, hipStream_t s) {
    if (n == 0) return;
    exclusive_scan_rec(in, out, n, ws, s);
}

}   // namespace dev
}   // namespace irf
