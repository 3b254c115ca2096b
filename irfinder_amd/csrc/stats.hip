// stats.hip — per-intron depth statistics and IR-table rows (gfx950).
//
// Replaces CoverageBlocks::fillHist / coverageFromHist / percentileFromHist /
// trimmedMeanFromHist and the per-row part of
// CoverageBlocksIRFinder::WriteOutput (⟨upstream⟩
// ReadBlockProcessor_CoverageBlocks.cpp).  Upstream builds a
// std::map<depth,count> per intron and walks it; every statistic is a function
// of the multiset of depths over the intron's measured bases, so here one
// workgroup per intron answers a handful of rank queries ("value at rank r" and
// "sum of the r smallest values") exactly, with a 10-bit-per-level radix
// select over an LDS histogram:
//   depth < 1024 (the common case): ONE pass over the depth values;
//   deeper introns: one filtered pass per level and query (rare).
// The integers that come out are identical to what the map walk yields; the
// few fp64 operations (divisions, one interpolation) are done in the same order
// as the CPU restatement, with contraction off, so doubles match bit for bit.
//
// Depth is never materialised: the workgroup rebuilds it on the fly from the
// difference planes.  finalize() leaves, for every 256-position segment of a
// plane, the depth just before the segment ("carry", scan.hip); a wave loads a
// segment as 64 x int4, one wave-wide prefix sum gives the 256 depths.  Both
// strands (non-directional table) are summed BEFORE the prefix (it is linear).
// HBM-bound: 4 B (one strand) or 8 B (both) per measured base, read once; the
// first/last-50-nt windows are captured from the same pass.
//
// Shape.  200 k introns of very different sizes (median 1.7 k measured bases, tail to 0.5 M).  Measured on MI355X the
// kernel is bound by instruction issue and by chains of dependent round trips, not by HBM bandwidth, so:
//   * two size classes, each a persistent launch with its own workgroup size (256 / 64 threads); the host deals the
//     introns of a class, longest first, to the least loaded workgroup (no long intron trails the launch);
//   * everything the row needs besides depth sits in ONE 128-byte record per intron, rank queries included; the
//     record two introns ahead, the block descriptors and the row's counters one intron ahead are fetched while the
//     workgroup still works on the current intron (header pipeline);
//   * the measured blocks of an intron are flattened into one list of segments: lane b of every wave keeps the
//     descriptor of block b in registers (no LDS, no barrier), a ballot finds the block of a segment; the loads of
//     kInFlight segments are issued back to back before any result is touched;
//   * per segment, everything wave-uniform is computed relative to the segment start on the scalar unit; a lane
//     whose four positions are measured and equal is one histogram entry of weight 4, and equal lanes are merged by
//     the wave (depth is piecewise constant: a segment that holds one value 256 times costs ONE LDS atomic);
//   * flank medians by radix select over ballots, rank queries by a DPP scan over the owning thread's bins.
#include "kernels.h"

#include <cstdio>
#include <vector>

namespace irf {
namespace dev {

namespace {

#ifndef STATS_INFLIGHT_BIG
#define STATS_INFLIGHT_BIG 2
#endif
#ifndef STATS_INFLIGHT_SMALL
#define STATS_INFLIGHT_SMALL 2
#endif
#ifndef STATS_MIN_WAVES
#define STATS_MIN_WAVES 6
#endif
#ifndef STATS_BITS_SMALL
#define STATS_BITS_SMALL 10
#endif
#ifndef STATS_MIN_WAVES_DUAL
#define STATS_MIN_WAVES_DUAL 6      // (4 = 128 registers, no spills: measured slower, 3.37 vs 3.17 ms on configs[2])
#endif
constexpr int kQueries = 8;
constexpr uint32_t kFastBins = 256; // one-wave class: depths below this take the four-bins-per-lane rank queries
constexpr int kSeg = 256;           // positions per carry segment (kernels.h: kDepthSegment)
constexpr uint32_t kMaxBlk = 64;    // measured blocks per descriptor chunk: one per lane

#ifdef STATS_PROF
// diagnostic build only (tools/build_variant.sh -DSTATS_PROF): wall-clock ticks (10 ns) per phase and workgroup
constexpr uint32_t kProfMax = 1u << 16;
__device__ uint32_t g_prof[kProfMax][8];
#define PROF_MARK(k) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); prof_acc[k] += (uint32_t)(now_ - prof_t); prof_t = now_; } } while (0)
#else
#define PROF_MARK(k) do { } while (0)
#endif

// pointers that come out of the table block in device memory are generic to the compiler: cast to the global address
// space, their loads stay global_load (a flat_load also counts against the LDS counter and waits with it)
#define AS1 __attribute__((address_space(1)))
template <class X> __device__ __forceinline__ AS1 X* gp(X* p) { return (AS1 X*)p; }
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4 load_int4(const AS1 int32_t* p) {
    const i32x4 v = *reinterpret_cast<const AS1 i32x4*>(p);
    return make_int4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ BlkRec load_blk(const AS1 BlkRec* p) {
    BlkRec r;
    r.gstart = p->gstart; r.cstart = p->cstart; r.len = p->len; r.pad = 0;
    return r;
}

struct Ctx {
    const AS1 int32_t* d0;          // difference planes
    const AS1 int32_t* d1;
    const AS1 uint32_t* c0;         // carries, one per 256-position segment
    const AS1 uint32_t* c1;
    int plane;                      // 0,1: one strand; 2: both
};

struct Seg { int4 a; uint32_t carry; };

// Inclusive prefix sum over the 64 lanes with DPP row shifts / broadcasts (6 VALU ops, no LDS traffic).
__device__ __forceinline__ uint32_t wave_inclusive_scan_dpp(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);    // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2,3
    return (uint32_t)v;
}

// Inclusive prefix sum inside every row of 16 lanes.
__device__ __forceinline__ uint32_t row_inclusive_scan_dpp(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    return (uint32_t)v;
}

// OR over the 64 lanes (same DPP ladder as the scan; the result is what lane 63 ends up with)
__device__ __forceinline__ uint32_t wave_or_dpp(uint32_t x) {
    int v = (int)x;
    v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

// k-th smallest (0-based) of the values held by the lanes of cand (wave-uniform mask, k < popcount), v < 2^nbits:
// binary radix select, one ballot per bit; everything but the bit test runs on the scalar unit.
__device__ __forceinline__ uint32_t wave_kth(uint32_t v, uint64_t cand, uint32_t k, int nbits) {
    uint32_t res = 0;
    for (int b = nbits - 1; b >= 0; b--) {
        const uint64_t ones = __ballot(((v >> b) & 1u) != 0u) & cand;
        const uint64_t zeros = cand & ~ones;
        const uint32_t nz = (uint32_t)__popcll(zeros);
        if (k < nz) { cand = zeros; } else { cand = ones; k -= nz; res |= 1u << b; }
    }
    return res;
}

__device__ __forceinline__ uint32_t lane_u32(uint32_t v, uint32_t src) {          // src is wave-uniform
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src);
}
__device__ __forceinline__ uint64_t lane_u64(uint64_t v, uint32_t src) {
    return (uint64_t)lane_u32((uint32_t)v, src) | ((uint64_t)lane_u32((uint32_t)(v >> 32), src) << 32);
}

// 64-bit field of a record kept one dword per lane: dwords at, at + 1
__device__ __forceinline__ uint64_t lane_u64v(uint32_t rec, uint32_t at) {
    return (uint64_t)lane_u32(rec, at) | ((uint64_t)lane_u32(rec, at + 1u) << 32);
}

__device__ __forceinline__ Seg load_seg(const Ctx& C, uint64_t k, uint32_t lane) {
    Seg s;
    const uint64_t e0 = k * kSeg + lane * 4u;
    if (C.plane == 2) {
        const int4 u = load_int4(C.d0 + e0);
        const int4 v = load_int4(C.d1 + e0);
        s.a.x = u.x + v.x; s.a.y = u.y + v.y; s.a.z = u.z + v.z; s.a.w = u.w + v.w;
        s.carry = C.c0[k] + C.c1[k];
    } else if (C.plane == 1) {
        s.a = load_int4(C.d1 + e0);
        s.carry = C.c1[k];
    } else {
        s.a = load_int4(C.d0 + e0);
        s.carry = C.c0[k];
    }
    return s;
}

// The loads of one segment, issued without touching their results, so that a wave can have several segments
// outstanding (load_seg adds the two strands at once, and with it waits for both loads).
struct RawSeg { int4 u, v; uint32_t cu, cv; };
__device__ __forceinline__ RawSeg issue_seg(const Ctx& C, uint64_t k, uint32_t lane) {
    RawSeg r;
    const uint64_t e0 = k * kSeg + lane * 4u;
    const AS1 int32_t* pu = C.plane == 1 ? C.d1 : C.d0;
    const AS1 uint32_t* cu = C.plane == 1 ? C.c1 : C.c0;
    r.u = load_int4(pu + e0);
    r.cu = cu[k];
    r.v = make_int4(0, 0, 0, 0);
    r.cv = 0;
    if (C.plane == 2) {
        r.v = load_int4(C.d1 + e0);
        r.cv = C.c1[k];
    }
    return r;
}
__device__ __forceinline__ Seg combine_seg(const RawSeg& r) {
    Seg s;
    s.a.x = r.u.x + r.v.x; s.a.y = r.u.y + r.v.y; s.a.z = r.u.z + r.v.z; s.a.w = r.u.w + r.v.w;
    s.carry = r.cu + r.cv;
    return s;
}

// One segment: f(seg0, e0, v0, v1, v2, v3) is called by ALL 64 lanes; seg0 = first plane position of the segment,
// e0 = this lane's first position, v0..v3 = depth at e0..e0+3.
template <class F>
__device__ __forceinline__ void emit_seg(const Seg& s, uint64_t k, uint32_t lane, F&& f) {
    const uint32_t t = (uint32_t)(s.a.x + s.a.y + s.a.z + s.a.w);
    const uint32_t incl = wave_inclusive_scan_dpp(t);
    const uint32_t v0 = s.carry + incl - t + (uint32_t)s.a.x;
    const uint32_t v1 = v0 + (uint32_t)s.a.y, v2 = v1 + (uint32_t)s.a.z, v3 = v2 + (uint32_t)s.a.w;
    f(k * kSeg, k * kSeg + lane * 4u, v0, v1, v2, v3);
}

// Depth of the PLANE positions of every 256-position segment that overlaps [gs, ge), spread over the workgroup: each
// wave walks whole segments (so every DPP / ballot inside f runs with 64 active lanes), two segments in flight.
// The callback clips to [gs, ge) itself.  (Side paths only: raw-coordinate windows, introns deeper than the histogram.)
template <int kWaves, class F>
__device__ __forceinline__ void for_each_segment(const Ctx& C, uint64_t gs, uint64_t ge, F&& f) {
    if (gs >= ge) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t k0 = gs / kSeg, k1 = (ge - 1) / kSeg;
    for (uint64_t k = k0 + wave; k <= k1; k += 2 * kWaves) {
        const bool two = k + kWaves <= k1;                 // wave-uniform
        const Seg sa = load_seg(C, k, lane);
        Seg sb = sa;
        if (two) sb = load_seg(C, k + kWaves, lane);
        emit_seg(sa, k, lane, f);
        if (two) emit_seg(sb, k + kWaves, lane, f);
    }
}

// per-position form: f(x, v, ok) for the four positions of every lane, ok = x in [gs, ge)
template <int kWaves, class F>
__device__ __forceinline__ void for_each_depth(const Ctx& C, uint64_t gs, uint64_t ge, F&& f) {
    for_each_segment<kWaves>(C, gs, ge, [&](uint64_t, uint64_t e0, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
        f(e0 + 0, v0, e0 + 0 >= gs && e0 + 0 < ge);
        f(e0 + 1, v1, e0 + 1 >= gs && e0 + 1 < ge);
        f(e0 + 2, v2, e0 + 2 >= gs && e0 + 2 < ge);
        f(e0 + 3, v3, e0 + 3 >= gs && e0 + 3 < ge);
    });
}

// Exclusive prefix over the threads of the workgroup of (count, sum); s_w* are LDS scratch of kWaves entries.
// The 64-bit sums (< 2^44: at most 2^28 values below 2^10 per thread... in fact bin * count summed over a thread's
// bins) are scanned as two 32-bit halves of 20 / 24 bits, so that neither wraps over 64 lanes: DPP only, no LDS.
template <int kWaves>
__device__ __forceinline__ void block_excl_scan(uint32_t tot, unsigned long long tsum, uint32_t& pre,
                                                unsigned long long& pre_sum, uint32_t* s_wc, unsigned long long* s_ws) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t c = wave_inclusive_scan_dpp(tot);
    const uint32_t lo = wave_inclusive_scan_dpp((uint32_t)tsum & 0xfffffu);
    const uint32_t hi = wave_inclusive_scan_dpp((uint32_t)(tsum >> 20));
    const unsigned long long s = ((unsigned long long)hi << 20) + lo;
    if (kWaves == 1) {
        pre = c - tot;
        pre_sum = s - tsum;
        return;
    }
    if (lane == 63) { s_wc[wave] = c; s_ws[wave] = s; }
    wg_barrier();
    uint32_t bc = 0;
    unsigned long long bs = 0;
    for (int w = 0; w < kWaves; w++)
        if (w < (int)wave) { bc += s_wc[w]; bs += s_ws[w]; }
    pre = bc + c - tot;
    pre_sum = bs + s - tsum;
    wg_barrier();
}

// The two ranks (0-based) the p-th percentile of n sorted values reads (OQ-7 formula), and its value from them.
struct PctRanks { unsigned long long ra, rb; uint32_t rem; bool need_a, need_b; };
__device__ __forceinline__ PctRanks percentile_ranks(unsigned long long n, uint32_t pct, bool interpolate) {
    PctRanks q;
    q.ra = 0; q.rb = 0; q.need_a = n > 0; q.need_b = false;
    const unsigned long long num = (n + 1ull) * pct;
    const unsigned long long k = num / 100ull;
    q.rem = (uint32_t)(num % 100ull);
    if (n > 0) {
        if (k < 1) q.ra = 0;
        else if (k >= n) q.ra = n - 1;
        else { q.ra = k - 1; q.rb = k; q.need_b = interpolate && q.rem != 0; }
    }
    return q;
}
__device__ __forceinline__ double percentile_value(const PctRanks& q, uint32_t va, uint32_t vb) {
    if (!q.need_a) return 0.0;
    const double vk = (double)va;
    if (!q.need_b) return vk;
    return vk + ((double)q.rem / 100.0) * ((double)vb - vk);
}

// kTables = 2: the non-directional AND the directional table in ONE pass over the planes (both strands are loaded for
// the first anyway; the second takes the intron's own strand out of the same registers): two histograms, two sets of
// flank windows, then medians / rank queries / row once per table.  kTables = 1: one table, T.directionality selects it.
template <int kThreads, int kMaxFlank, int kInFlight, int kBinBits, int kTables>
__global__ __launch_bounds__(kThreads, (kTables == 2 ? STATS_MIN_WAVES_DUAL : STATS_MIN_WAVES)) void intron_stats_kernel(const IntronTables* __restrict__ Tp, const uint32_t* __restrict__ wg_off, uint32_t prof_slot) {
    const IntronTables& T = *Tp;
    constexpr int kWaves = kThreads / 64;
    constexpr int kBins = 1 << kBinBits;                   // histogram bins per level of the rank select
    constexpr int kBinsPerThread = kBins / kThreads;
    static_assert(kBinsPerThread >= 1 && kBinsPerThread <= 16, "a thread's bins must fit one DPP row");
    __shared__ __align__(16) uint32_t s_cnt[kTables * kBins];
    __shared__ uint32_t s_win[kTables][2][kMaxFlank];      // depths of the two flank windows, per table
    __shared__ uint32_t s_med[4];                          // per window: the two order statistics its median reads
    __shared__ uint32_t s_wc[kWaves];
    __shared__ unsigned long long s_ws[kWaves];
    __shared__ uint32_t s_max, s_overflow[kTables], s_hi[kTables], s_sel_bin, s_sel_cnt;
    __shared__ unsigned long long s_total, s_nz, s_acc;
    __shared__ uint32_t s_qval[kQueries];
    __shared__ unsigned long long s_qsum[kQueries];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const uint32_t k_begin = wg_off[blockIdx.x], k_end = wg_off[blockIdx.x + 1];
    if (k_begin >= k_end) return;
    int dirn = T.directionality;
    if (T.vote) {       // the decision is taken here (kernels.h); only the launch whose shape fits it goes on
        const unsigned long long same = gp(T.vote)[0], opp = gp(T.vote)[1];
        dirn = same > opp * (unsigned long long)T.dir_ratio ? 1 : (opp > same * (unsigned long long)T.dir_ratio ? -1 : 0);
        if ((kTables == 2) != (dirn != 0)) return;
    }
#ifdef STATS_PROF
    unsigned long long prof_t = wall_clock64();
    uint32_t prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    Ctx C;
    C.d0 = gp(T.diff);
    C.d1 = gp(T.diff) + T.plane_stride;
    C.c0 = gp(T.carry);
    C.c1 = gp(T.carry) + T.plane_stride / kSeg;
    C.plane = 2;

    // ---- the header pipeline -------------------------------------------------------------------------------------
    // Everything an intron needs before its first plane load is fetched while the workgroup still works on the
    // introns before it: the record two introns ahead, the block descriptors and the row's counters one ahead.
    // Vector loads on purpose (lane l holds dword l of the record; every wave keeps its own copy): they retire in
    // order under vmcnt, so waiting for plane data never waits for anything younger, and no LDS wait drains them.
    const AS1 uint32_t* __restrict__ recw = reinterpret_cast<const AS1 uint32_t*>(gp(T.recs));
    const AS1 BlkRec* __restrict__ blks = gp(T.blks);
    auto load_rec = [&](uint32_t k) -> uint32_t {
        const uint32_t kc = k < k_end ? k : k_end - 1u;
        return recw[(uint64_t)kc * (sizeof(IntronRec) / 4u) + (lane & 31u)];
    };
    auto load_hdr = [&](uint32_t rec, BlkRec& d, uint32_t& tc) {
        const uint32_t blk_off = lane_u32(rec, 6), n_blk = lane_u32(rec, 7), flags = lane_u32(rec, 14);
        d = load_blk(blks + blk_off + (lane < n_blk ? lane : 0u));                // the table ends with one spare entry
        tc = 0;
        // lanes 0..9: sp_cnt[sp_s], sp_cnt[sp_e], left_cnt[us], right_cnt[ue], known_cnt[kj], each for both strands
        const uint32_t which = lane >> 1, half = lane & 1u;
        int32_t ix = -1;
        const AS1 uint32_t* base = gp(T.sp_cnt);
        uint64_t n = T.n_sp;
        const int32_t sp_s = (int32_t)lane_u32(rec, 8), sp_e = (int32_t)lane_u32(rec, 9), us = (int32_t)lane_u32(rec, 10),
                      ue = (int32_t)lane_u32(rec, 11), kj = (int32_t)lane_u32(rec, 12);
        if (which == 0u) ix = sp_s;
        else if (which == 1u) ix = sp_e;
        else if (which == 2u) { ix = us; base = gp(T.left_cnt); n = T.n_ustart; }
        else if (which == 3u) { ix = ue; base = gp(T.right_cnt); n = T.n_uend; }
        else if (which == 4u) { ix = kj; base = gp(T.known_cnt); n = T.n_kj; }
        if ((flags & 1u) && !(T.ablate & 2u) && ix >= 0) tc = base[(uint64_t)half * n + (uint64_t)ix];
    };
    // The histograms are zero between introns: the rank queries read every bin that holds anything, and whoever read a
    // bin clears it (a zeroing pass per intron cost a tenth of the kernel).
    for (int k = tid; k < kTables * kBins / 4; k += kThreads) reinterpret_cast<uint4*>(s_cnt)[k] = make_uint4(0u, 0u, 0u, 0u);
    uint32_t recA = load_rec(k_begin), recB = load_rec(k_begin + 1u);
    BlkRec dA;
    uint32_t tcA;
    load_hdr(recA, dA, tcA);

    for (uint32_t kk = k_begin; kk < k_end; kk++) {
    const uint32_t recC = load_rec(kk + 2u);
    BlkRec dB;
    uint32_t tcB;
    load_hdr(recB, dB, tcB);

    const uint64_t gS = lane_u64v(recA, 0), gE = lane_u64v(recA, 2);
    const unsigned long long N = lane_u64v(recA, 4);
    const uint32_t b0 = lane_u32(recA, 6), nb = lane_u32(recA, 7);
    const int32_t r_kj = (int32_t)lane_u32(recA, 12);
    const uint32_t r_row = lane_u32(recA, 13), r_flags = lane_u32(recA, 14), r_prem = lane_u32(recA, 15);
    const uint32_t qneed = (r_flags >> 8) & 0xffu;
    const bool r_valid = (r_flags & 1u) != 0;
    const bool valid = r_valid && !(T.ablate & 1u);     // false: chromosome absent from the BAM header / not owned => depth 0 everywhere
    // the plane(s) a table reads: 2 = both strands (non-directional), else the intron's own strand under the vote
    const int plane_strand = (((r_flags & 2u) != 0) == (dirn > 0)) ? 1 : 0;
    int planes[kTables];
    if (kTables == 2) { planes[0] = 2; planes[kTables - 1] = plane_strand; }
    else planes[0] = dirn != 0 ? plane_strand : 2;
    C.plane = kTables == 2 ? 2 : planes[0];                  // what the main pass loads

    // first / last flank_len nt of the intron, flank_offset nt inside it (full coordinates)
    uint64_t w0[2], w1[2];
    {
        const uint64_t ilen = gE - gS;
        const uint64_t fo = T.flank_offset, fl = T.flank_len < (uint32_t)kMaxFlank ? T.flank_len : (uint32_t)kMaxFlank;
        w0[0] = gS + (fo < ilen ? fo : ilen);
        w1[0] = gS + (fo + fl < ilen ? fo + fl : ilen);
        w1[1] = gE - (fo < ilen ? fo : ilen);
        w0[1] = gE - (fo + fl < ilen ? fo + fl : ilen);
    }

    if (tid < kTables) { s_overflow[tid] = 0; s_hi[tid] = 0; }
    if (tid == 0) { s_max = 0; s_total = 0; s_nz = 0; s_acc = 0; }
    if (tid < 4) s_med[tid] = 0;
    if (tid < kQueries) { s_qval[tid] = 0; s_qsum[tid] = 0; }
    wg_barrier();
    PROF_MARK(0);

    // ---------------- pass A: histogram (clipped) + window capture ----------------
    // (total, non-zero count and maximum are read off the histogram afterwards)
    uint32_t nwin[2] = {0, 0};
    if (valid) {
        bool ovf[kTables], hib[kTables];       // a depth beyond the histogram / of kFastBins or more was seen
#pragma unroll
        for (int t = 0; t < kTables; t++) { ovf[t] = false; hib[t] = false; }
        if (!T.flank_use_measured) {
            // raw-coordinate windows: the engine keeps the buckets under them, contiguous in the plane
            for (int wi = 0; wi < 2; wi++) {
                if (w0[wi] >= w1[wi]) continue;
                const uint64_t len = w1[wi] - w0[wi];
                const uint64_t cs = plane_coord(T.pmap, w0[wi]);
#pragma unroll
                for (int t = 0; t < kTables; t++) {
                    Ctx Cw = C;
                    Cw.plane = planes[t];
                    uint32_t* dst = s_win[t][wi];
                    for_each_depth<kWaves>(Cw, cs, cs + len, [&](uint64_t x, uint32_t v, bool ok) { if (ok) dst[(uint32_t)(x - cs)] = v; });
                }
                nwin[wi] = (uint32_t)len;
            }
        }
        for (uint32_t cb = 0; cb < nb; cb += kMaxBlk) {
            // lane b keeps measured block cb + b: plane start, length, its slice of the flattened segment list, and
            // the parts of the two flank windows that fall inside it (every wave builds the same registers)
            const uint32_t nbc = nb - cb < kMaxBlk ? nb - cb : kMaxBlk;
            const bool has = lane < nbc;
            BlkRec br = dA;                                            // the first chunk came with the header
            if (cb) br = load_blk(blks + b0 + cb + (has ? lane : 0u));
            const uint64_t d_bc = br.cstart;
            const uint32_t d_bl = has ? br.len : 0u;
            const uint32_t d_nseg = d_bl ? (uint32_t)((d_bc + d_bl - 1u) / kSeg - d_bc / kSeg + 1u) : 0u;
            uint32_t d_wlo[2] = {0, 0}, d_wn[2] = {0, 0}, d_wat[2] = {0, 0};
            if (T.flank_use_measured) {
#pragma unroll
                for (int wi = 0; wi < 2; wi++) {
                    const uint64_t bs = br.gstart;
                    const uint64_t s = bs > w0[wi] ? bs : w0[wi], e = bs + d_bl < w1[wi] ? bs + d_bl : w1[wi];
                    if (d_bl && s < e) { d_wlo[wi] = (uint32_t)(s - bs); d_wn[wi] = (uint32_t)(e - s); }
                    const uint32_t incl = wave_inclusive_scan_dpp(d_wn[wi]);
                    d_wat[wi] = nwin[wi] + incl - d_wn[wi];
                    nwin[wi] += lane_u32(incl, 63u);
                }
            }
            const uint32_t d_seg_end = wave_inclusive_scan_dpp(d_nseg);       // lanes >= nbc hold the total
            const uint32_t d_seg_start = d_seg_end - d_nseg;
            const uint32_t S = lane_u32(d_seg_end, 63u);

            for (uint32_t s0 = wave; s0 < S; s0 += kWaves * kInFlight) {
                RawSeg raw[kInFlight];
                uint32_t sb[kInFlight];
                uint64_t sk[kInFlight];
                // addresses first (ballots and readlanes), then nothing but loads, then the work: the loads of all
                // kInFlight segments are outstanding together
#pragma unroll
                for (int j = 0; j < kInFlight; j++) {
                    const uint32_t s = s0 + j * kWaves;
                    const uint32_t sc = s < S ? s : S - 1u;           // past the end: load the last segment again, unused
                    sb[j] = (uint32_t)__popcll(__ballot(d_seg_end <= sc));        // blocks that end before segment sc
                    sk[j] = lane_u64(d_bc, sb[j]) / kSeg + (sc - lane_u32(d_seg_start, sb[j]));
                }
#pragma unroll
                for (int j = 0; j < kInFlight; j++) {
                    if (T.ablate & 8u) { raw[j].u = raw[j].v = make_int4(0, 0, 0, 0); raw[j].cu = raw[j].cv = (uint32_t)sk[j] & 3u; continue; }   // diagnostic: no loads
                    raw[j] = issue_seg(C, sk[j], lane);
                }
#pragma unroll
                for (int j = 0; j < kInFlight; j++) {
                    if (s0 + j * kWaves >= S) continue;                // wave-uniform
                    if (T.ablate & 4u) { const Seg sgj = combine_seg(raw[j]); if (sgj.a.x + sgj.a.y + sgj.a.z + sgj.a.w + (int)sgj.carry == 0x7fffffff) ovf[0] = true; continue; }     // diagnostic: loads only
                    // everything wave-uniform about the segment, relative to its first position (scalar unit): the
                    // measured part [lo, hi) and the parts of the two flank windows, [wl, wh) with element index
                    // wat + (p - wa) in the window buffer
                    const uint32_t b = sb[j];
                    const uint64_t seg0 = sk[j] * kSeg;
                    const int64_t rel = (int64_t)(lane_u64(d_bc, b) - seg0);           // block start - segment start: < 256
                    const int64_t rel_end = rel + (int64_t)lane_u32(d_bl, b);
                    const int32_t lo = rel > 0 ? (int32_t)rel : 0;
                    const int32_t hi = rel_end < (int64_t)kSeg ? (int32_t)rel_end : kSeg;
                    int32_t wa[2], wl[2], wh[2];
                    uint32_t wat[2];
#pragma unroll
                    for (int wi = 0; wi < 2; wi++) {
                        const uint32_t wn = lane_u32(d_wn[wi], b);
                        int64_t a = rel + (int64_t)lane_u32(d_wlo[wi], b);
                        a = a < -(int64_t)(1 << 20) ? -(int64_t)(1 << 20) : a;         // far before the segment: empty below
                        wa[wi] = (int32_t)a;
                        const int64_t e = a + (int64_t)wn;
                        wl[wi] = a > 0 ? (int32_t)a : 0;
                        wh[wi] = wn == 0u ? 0 : (e < (int64_t)kSeg ? (int32_t)e : kSeg);
                        wat[wi] = lane_u32(d_wat[wi], b);
                    }
                    const bool win_here = wl[0] < wh[0] || wl[1] < wh[1];
                    const int32_t p0 = (int32_t)(lane * 4u);
#pragma unroll
                    for (int t = 0; t < kTables; t++) {
                        // the series of this table: both strands added (the prefix is linear), or the intron's own strand
                        Seg sgj;
                        if (kTables == 1 || t == 0) sgj = combine_seg(raw[j]);
                        else if (plane_strand == 1) { sgj.a = raw[j].v; sgj.carry = raw[j].cv; }
                        else { sgj.a = raw[j].u; sgj.carry = raw[j].cu; }
                        uint32_t* const cnt = s_cnt + t * kBins;
                        // depths of this lane's four positions
                        const uint32_t tt = (uint32_t)(sgj.a.x + sgj.a.y + sgj.a.z + sgj.a.w);
                        const uint32_t incl = wave_inclusive_scan_dpp(tt);
                        const uint32_t v0 = sgj.carry + incl - tt + (uint32_t)sgj.a.x;
                        const uint32_t v1 = v0 + (uint32_t)sgj.a.y, v2 = v1 + (uint32_t)sgj.a.z, v3 = v2 + (uint32_t)sgj.a.w;
                        // (conservative: positions outside the measured part count too; the flag only selects the query path)
                        hib[t] = hib[t] || ((v0 | v1 | v2 | v3) >= kFastBins);
                        // Depth is piecewise constant.  A lane whose four positions are all measured and equal ("full") is
                        // one histogram entry of weight 4; full lanes are merged per value by the wave (two rounds of
                        // leader election: a segment rarely holds more than two frequent depths), the rest add on their own.
                        const bool full = v1 == v0 && v2 == v0 && v3 == v0 && p0 >= lo && p0 + 3 < hi;
                        uint64_t todo = __ballot(full);
                        bool pending = full;
#pragma unroll
                        for (int round = 0; round < 2; round++) {
                            if (!todo) break;                                  // wave-uniform
                            const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
                            const uint32_t lead = lane_u32(v0, leader);
                            const uint64_t same = __ballot(pending && v0 == lead);
                            if (lead < (uint32_t)kBins) { if (lane == leader) atomicAdd(&cnt[lead], 4u * (uint32_t)__popcll(same)); }
                            else ovf[t] = true;
                            pending = pending && v0 != lead;
                            todo &= ~same;
                        }
                        if (todo) {
                            if (pending) { if (v0 < (uint32_t)kBins) atomicAdd(&cnt[v0], 4u); else ovf[t] = true; }
                        }
                        if (__ballot(!full)) {
                            // lanes on a depth change or on the edge of the measured part: position by position
                            const uint32_t vv[4] = {v0, v1, v2, v3};
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                if (!full && p0 + q >= lo && p0 + q < hi) { if (vv[q] < (uint32_t)kBins) atomicAdd(&cnt[vv[q]], 1u); else ovf[t] = true; }
                            }
                        }
                        if (win_here) {
                            const uint32_t vv[4] = {v0, v1, v2, v3};
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const int32_t pq = p0 + q;
                                if (pq >= wl[0] && pq < wh[0]) s_win[t][0][wat[0] + (uint32_t)(pq - wa[0])] = vv[q];
                                if (pq >= wl[1] && pq < wh[1]) s_win[t][1][wat[1] + (uint32_t)(pq - wa[1])] = vv[q];
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < kTables; t++) { if (ovf[t]) s_overflow[t] = 1; if (hib[t]) s_hi[t] = 1; }
        PROF_MARK(1);
    } else {
        // all-zero depth: N zeros in the histogram, all-zero windows of the right size
        if (tid < kTables) s_cnt[tid * kBins] = (uint32_t)N;
        for (int wi = 0; wi < 2; wi++) {
            uint32_t n = 0;
            if (w0[wi] < w1[wi]) {
                if (T.flank_use_measured) {
                    for (uint32_t b = b0; b < b0 + nb; b++) {
                        const uint64_t bs = blks[b].gstart, be = bs + blks[b].len;
                        const uint64_t s = bs > w0[wi] ? bs : w0[wi], e = be < w1[wi] ? be : w1[wi];
                        if (s < e) n += (uint32_t)(e - s);
                    }
                } else {
                    n = (uint32_t)(w1[wi] - w0[wi]);
                }
            }
            for (uint32_t k = tid; k < n; k += kThreads) { s_win[0][wi][k] = 0; s_win[kTables - 1][wi][k] = 0; }
            nwin[wi] = n;
        }
    }
    wg_barrier();
    PROF_MARK(2);
    // ---------------- per table: medians, rank queries, the row ----------------
#pragma unroll
    for (int t = 0; t < kTables; t++) {
    if (t > 0) {            // the per-table results of the table before were consumed by its row (barrier at its end)
        if (tid == 0) { s_max = 0; s_total = 0; s_nz = 0; s_acc = 0; }
        if (tid < 4) s_med[tid] = 0;
        if (tid < kQueries) { s_qval[tid] = 0; s_qsum[tid] = 0; }
        wg_barrier();
    }
    uint32_t* const cnt = s_cnt + t * kBins;
    const int plane = planes[t];
    C.plane = plane;
    AS1 double* const out_f64 = gp(t == 0 ? T.out_f64 : T.out_f64_b);
    AS1 uint32_t* const out_u32 = gp(t == 0 ? T.out_u32 : T.out_u32_b);
    const bool overflow = s_overflow[t] != 0;
    if (overflow) {
        // depth beyond the single-level histogram somewhere: maximum, total and non-zero count by a pass of their own
        uint32_t vmax = 0;
        unsigned long long tsum = 0, nz = 0;
        for (uint32_t b = b0; b < b0 + nb; b++) {
            const uint64_t bc = blks[b].cstart;
            for_each_depth<kWaves>(C, bc, bc + blks[b].len, [&](uint64_t, uint32_t v, bool ok) {
                if (ok) { vmax = v > vmax ? v : vmax; tsum += v; nz += v ? 1u : 0u; }
            });
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            uint32_t om = __shfl_xor(vmax, d);
            vmax = om > vmax ? om : vmax;
            tsum += __shfl_xor(tsum, d);
            nz += __shfl_xor(nz, d);
        }
        if (lane == 0) { atomicMax(&s_max, vmax); atomicAdd(&s_total, tsum); atomicAdd(&s_nz, nz); }
    } else if (tid == 0) {
        s_nz = N - cnt[0];
    }

    // ---------------- flank medians: rank of every window element by counting ----------------
    PctRanks fq[2];
    fq[0] = percentile_ranks(nwin[0], 50u, T.pct_interpolate != 0);
    fq[1] = percentile_ranks(nwin[1], 50u, T.pct_interpolate != 0);
    if (nwin[0] <= 64u && nwin[1] <= 64u) {
        // a window per wave, an element per lane: the two order statistics by radix select over ballots
        const int wi1_wave = kWaves > 1 ? 1 : 0;
#pragma unroll
        for (int wi = 0; wi < 2; wi++) {
            if ((int)wave != (wi ? wi1_wave : 0)) continue;
            const uint32_t n = nwin[wi];
            if (n == 0u) continue;
            const uint32_t v = lane < n ? s_win[t][wi][lane] : 0u;
            const uint64_t cand = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
            const int nbits = 32 - __clz((int)(wave_or_dpp(v) | 1u));
            const uint32_t va = wave_kth(v, cand, (uint32_t)fq[wi].ra, nbits);
            const uint32_t vb = fq[wi].need_b ? wave_kth(v, cand, (uint32_t)fq[wi].rb, nbits) : 0u;
            if (lane == 0) { s_med[2 * wi] = va; s_med[2 * wi + 1] = vb; }
        }
    } else {
        const uint32_t n0 = nwin[0], n1 = nwin[1];
        for (uint32_t e = tid; e < n0 + n1; e += kThreads) {
            const int wi = e >= n0 ? 1 : 0;
            const uint32_t* w = s_win[t][wi];
            const uint32_t n = wi ? n1 : n0, idx = wi ? e - n0 : e;
            const uint32_t v = w[idx];
            uint32_t r = 0;
            for (uint32_t j = 0; j < n; j++) {
                const uint32_t o = w[j];
                r += (o < v || (o == v && j < idx)) ? 1u : 0u;
            }
            // the ranks are a permutation: one writer per slot
            if (wi == 0) {
                if (r == fq[0].ra) s_med[0] = v;
                if (fq[0].need_b && r == fq[0].rb) s_med[1] = v;
            } else {
                if (r == fq[1].ra) s_med[2] = v;
                if (fq[1].need_b && r == fq[1].rb) s_med[3] = v;
            }
        }
    }
    PROF_MARK(3);

    // ---------------- the rank queries ----------------
    // q0,q1: sums below ranks lo / hi of the trimmed mean; q2..q7: values for p25, p50, p75 (ranks k-1 and k);
    // the ranks were worked out by the host (kernels.h: stats_queries) and travel in the record
    const unsigned long long lo = lane_u32(recA, 16), hi = lane_u32(recA, 17);
    if (kWaves == 1 && N > 0 && !overflow && s_hi[t] == 0u && N < (1ull << 23)) {
        // One wave, every depth below kFastBins (nearly every intron of the small class): lane b takes bins 4b .. 4b + 3 in
        // one LDS read, two wave scans give the counts and the weighted sums below every lane, and a query is a ballot
        // (which lane holds rank r) plus a few selects inside that lane.  (N < 2^23: the weighted sums fit 32 bits.)
        static_assert(kFastBins == 256u, "four bins per lane");
        const uint4 c4 = reinterpret_cast<const uint4*>(cnt)[lane];
        const uint32_t b4 = lane * 4u;
        const uint32_t lt = c4.x + c4.y + c4.z + c4.w;
        const uint32_t w1 = c4.x * b4, w2 = w1 + c4.y * (b4 + 1u), w3 = w2 + c4.z * (b4 + 2u), lw = w3 + c4.w * (b4 + 3u);
        const uint32_t ic = wave_inclusive_scan_dpp(lt), iw = wave_inclusive_scan_dpp(lw);
        if (lane == 63u) s_total = iw;                       // sum of all depths
        const uint32_t cx = c4.x, cxy = c4.x + c4.y, cxyz = cxy + c4.z;
        for (int q = 0; q < kQueries; q++) {
            if (!((qneed >> q) & 1u)) continue;              // wave-uniform
            const uint32_t r = lane_u32(recA, 16u + q);
            const uint64_t holds = __ballot(ic > r);         // (r < N: some lane's inclusive count exceeds it)
            const uint32_t rr = r - (ic - lt);               // rank among this lane's bins (meaningful on the first lane of `holds`)
            const uint32_t j = (rr >= cx ? 1u : 0u) + (rr >= cxy ? 1u : 0u) + (rr >= cxyz ? 1u : 0u);
            const uint32_t below_c = j == 0u ? 0u : (j == 1u ? cx : (j == 2u ? cxy : cxyz));
            const uint32_t below_w = j == 0u ? 0u : (j == 1u ? w1 : (j == 2u ? w2 : w3));
            const uint32_t bin = b4 + j;
            if (holds && lane == (uint32_t)__ffsll((long long)holds) - 1u) {
                s_qval[q] = bin;
                s_qsum[q] = (unsigned long long)(iw - lw) + below_w + (unsigned long long)bin * (rr - below_c);
            }
        }
        if (lt) reinterpret_cast<uint4*>(cnt)[lane] = make_uint4(0u, 0u, 0u, 0u);
        wg_barrier();
    } else if (N > 0 && !overflow) {
        // single level: bins are the depths themselves
        uint32_t c[kBinsPerThread];
        uint32_t tot = 0;
        unsigned long long tsum = 0;
#pragma unroll
        for (int j = 0; j < kBinsPerThread; j++) {
            c[j] = cnt[tid * kBinsPerThread + j];
            tot += c[j];
            tsum += (unsigned long long)c[j] * (unsigned long long)(tid * kBinsPerThread + j);
        }
        uint32_t pre;
        unsigned long long pre_sum;
        block_excl_scan<kWaves>(tot, tsum, pre, pre_sum, s_wc, s_ws);
        if (tid == kThreads - 1) s_total = pre_sum + tsum;       // sum of all depths
        // the thread whose bins hold rank r is found by a ballot; then the lanes of its wave take one of its bins each.
        // The second rank of a percentile (q = 3, 5, 7: rank k after rank k - 1, kernels.h: stats_queries) mostly lies in
        // the same bin as the first: the wave that answered the first one then answers it from what it kept.
        bool prev_here = false;             // this wave answered query q - 1 ...
        uint32_t prev_left = 0, prev_bin = 0;   // ... and its bin holds prev_left more ranks behind that one
        for (int q = 0; q < kQueries; q++) {
            if (!((qneed >> q) & 1u)) { prev_here = false; continue; }             // wave-uniform
            if (q >= 3 && (q & 1) && prev_here && prev_left > 0u) {
                if (lane == 0) s_qval[q] = prev_bin;                               // (the sums of q >= 2 are never read)
                prev_here = false;
                continue;
            }
            prev_here = false;
            const uint32_t r = lane_u32(recA, 16u + q);
            const uint64_t own = __ballot(r >= pre && r - pre < tot);
            if (!own) continue;                                                    // another wave's bins
            const uint32_t ol = (uint32_t)__ffsll((long long)own) - 1u;
            const uint32_t rr = r - lane_u32(pre, ol);                             // rank among the owner's bins
            const unsigned long long o_psum = lane_u64(pre_sum, ol);
            const uint32_t base = (wave * 64u + ol) * (uint32_t)kBinsPerThread;    // the owner's first bin
            const bool mine = lane < (uint32_t)kBinsPerThread;
            const uint32_t cj = mine ? cnt[base + lane] : 0u;
            const uint32_t wj = cj * lane;                                         // (bin - base) * count: N < 2^28 (host check)
            const uint32_t ic = row_inclusive_scan_dpp(cj), iw = row_inclusive_scan_dpp(wj);
            const uint64_t hit = __ballot(mine && rr >= ic - cj && rr < ic);       // exactly one lane
            const uint32_t hl = (uint32_t)__ffsll((long long)hit) - 1u;
            const uint32_t ex_c = lane_u32(ic - cj, hl), ex_w = lane_u32(iw - wj, hl);
            const unsigned long long bin = base + hl;
            if (lane == 0) {
                s_qval[q] = (uint32_t)bin;
                s_qsum[q] = o_psum + (unsigned long long)base * ex_c + ex_w + bin * (unsigned long long)(rr - ex_c);
            }
            prev_here = true;
            prev_bin = (uint32_t)bin;
            prev_left = lane_u32(ic, hl) - 1u - rr;                                // ranks of this bin behind rank r
        }
        // (a wave's queries read only bins of its own threads, and a wave's LDS accesses keep their order: no barrier needed)
        if (tot) {
#pragma unroll
            for (int j = 0; j < kBinsPerThread; j++) cnt[tid * kBinsPerThread + j] = 0;
        }
        wg_barrier();
    } else if (N > 0) {
        // multi-level radix select, one descent per query; a sum query (q < 2) then adds the values below the selected one
        wg_barrier();
        int bits = 32 - __clz((int)s_max);
        int levels = (bits + kBinBits - 1) / kBinBits;
        if (levels < 1) levels = 1;
        for (int q = 0; q < kQueries; q++) {
            if (!((qneed >> q) & 1u)) continue;            // uniform across the block
            unsigned long long r = lane_u32(recA, 16u + q);
            uint32_t prefix = 0;
            for (int lv = 0; lv < levels; lv++) {
                const int shift = kBinBits * (levels - 1 - lv);
                for (int k = tid; k < kBins; k += kThreads) cnt[k] = 0;
                wg_barrier();
                for (uint32_t b = b0; b < b0 + nb; b++) {
                    const uint64_t bc = blks[b].cstart;
                    for_each_depth<kWaves>(C, bc, bc + blks[b].len, [&](uint64_t, uint32_t v, bool ok) {
                        const bool in = ok && (lv == 0 || (v >> (shift + kBinBits)) == prefix);
                        if (in) atomicAdd(&cnt[(v >> shift) & (uint32_t)(kBins - 1)], 1u);
                    });
                }
                wg_barrier();
                uint32_t c[kBinsPerThread];
                uint32_t tot = 0;
#pragma unroll
                for (int j = 0; j < kBinsPerThread; j++) {
                    c[j] = cnt[tid * kBinsPerThread + j];
                    tot += c[j];
                }
                uint32_t pre;
                unsigned long long pre_sum;
                block_excl_scan<kWaves>(tot, 0ull, pre, pre_sum, s_wc, s_ws);
                if (r >= pre && r < (unsigned long long)pre + tot) {
                    unsigned long long run = pre;
#pragma unroll
                    for (int j = 0; j < kBinsPerThread; j++) {
                        if (r < run + c[j]) { s_sel_bin = tid * kBinsPerThread + j; s_sel_cnt = (uint32_t)run; break; }
                        run += c[j];
                    }
                }
                wg_barrier();
                prefix = (prefix << kBinBits) | s_sel_bin;
                r -= s_sel_cnt;
                wg_barrier();
            }
            // prefix = the value at the rank; r = the rank's position among the copies of that value
            unsigned long long below = 0;
            if (q < 2) {
                unsigned long long part = 0;
                for (uint32_t b = b0; b < b0 + nb; b++) {
                    const uint64_t bc = blks[b].cstart;
                    for_each_depth<kWaves>(C, bc, bc + blks[b].len, [&](uint64_t, uint32_t v, bool ok) {
                        if (ok && v < prefix) part += v;
                    });
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
                if (lane == 0 && part) atomicAdd(&s_acc, part);
                wg_barrier();
                below = s_acc;
                wg_barrier();
                if (tid == 0) s_acc = 0;
            }
            if (tid == 0) { s_qval[q] = prefix; s_qsum[q] = below + (unsigned long long)prefix * r; }
        }
        for (int k = tid; k < kBins; k += kThreads) cnt[k] = 0;            // (the histograms are zero between introns)
        wg_barrier();
    } else {
        wg_barrier();
    }
    PROF_MARK(4);

    // ---------------- the row ----------------
    // the counters that came with the header: lanes 2w, 2w+1 = strand 0 / 1 of counter w
    auto pick = [&](uint32_t w) -> uint32_t {
        const uint32_t a = lane_u32(tcA, 2u * w), b = lane_u32(tcA, 2u * w + 1u);
        return plane == 2 ? a + b : (plane == 1 ? b : a);
    };
    const uint32_t spl = pick(0), spr = pick(1), jl = pick(2), jr = pick(3);
    uint32_t je = pick(4);                  // (Start,End) is a ref-sj junction: its reads were counted in place
    if (tid == 0) {
        const double flank0 = percentile_value(fq[0], s_med[0], s_med[1]);
        const double flank1 = percentile_value(fq[1], s_med[2], s_med[3]);
        const unsigned long long total = s_total, nzc = s_nz;
        double coverage = 0.0, depth = 0.0, pv[3] = {0.0, 0.0, 0.0};
        if (N > 0) {
            coverage = (double)nzc / (double)N;
            const unsigned long long s_hi = hi >= N ? total : (hi == 0 ? 0ull : s_qsum[1]);
            const unsigned long long s_lo = lo == 0 ? 0ull : (lo >= N ? total : s_qsum[0]);
            depth = hi > lo ? (double)(s_hi - s_lo) / (double)(hi - lo) : 0.0;
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const double vk = (double)s_qval[2 + 2 * p];
                if ((qneed >> (3 + 2 * p)) & 1u) pv[p] = vk + ((double)((r_prem >> (8 * p)) & 0xffu) / 100.0) * ((double)s_qval[3 + 2 * p] - vk);
                else pv[p] = vk;
            }
        }
        if (r_valid && !(T.ablate & 2u) && r_kj < 0) {
            // (Start,End) is not in ref-sj: look it up among the novel junctions
            const uint64_t span = gE - gS;
            if (span < (1ull << kJuncSpanBits) && gS < kJuncMaxGlobal && T.n_unique > 0) {
                const uint64_t key0 = (gS << kJuncLeftShift) | (span << 1);
                uint64_t lo_i = 0, hi_i = T.n_unique;
                while (lo_i < hi_i) {
                    const uint64_t mid = lo_i + ((hi_i - lo_i) >> 1);
                    if (gp(T.ukeys)[mid] < key0) lo_i = mid + 1; else hi_i = mid;
                }
                for (uint64_t u = lo_i; u < T.n_unique && u < lo_i + 2 && (gp(T.ukeys)[u] >> 1) == (key0 >> 1); u++) {
                    const int d = (int)(gp(T.ukeys)[u] & 1ull);
                    if (plane == 2 || plane == d) je += gp(T.ucount)[u];
                }
            }
        }
        const uint32_t smax = jl > jr ? jl : jr;
        const double den = depth + (double)smax;
        const double irratio = den > 0.0 ? depth / den : 0.0;
        const double sp_max = (double)(spl > spr ? spl : spr);
        const double fl_max = flank0 > flank1 ? flank0 : flank1;
        const double maxflank = sp_max > fl_max ? sp_max : fl_max;
        uint32_t warn;
        if ((double)je + depth < T.warn_lowcover_min) warn = 1;
        else if ((double)je < T.warn_lowsplice_min) warn = 2;
        else if ((double)je * T.warn_minor_ratio < (double)smax) warn = 3;
        else if (T.warn_nonuniform_and ? (maxflank > depth + T.warn_nonuniform_abs && maxflank > depth * T.warn_nonuniform_rel)
                                       : (maxflank > depth * T.warn_nonuniform_rel + T.warn_nonuniform_abs)) warn = 4;
        else warn = 0;
        const uint64_t n = T.n_intron;
        const uint64_t i = r_row;
        out_f64[0 * n + i] = coverage;
        out_f64[1 * n + i] = depth;
        out_f64[2 * n + i] = pv[0];
        out_f64[3 * n + i] = pv[1];
        out_f64[4 * n + i] = pv[2];
        out_f64[5 * n + i] = flank0;
        out_f64[6 * n + i] = flank1;
        out_f64[7 * n + i] = irratio;
        out_u32[0 * n + i] = spl;
        out_u32[1 * n + i] = spr;
        out_u32[2 * n + i] = jl;
        out_u32[3 * n + i] = jr;
        out_u32[4 * n + i] = je;
        out_u32[5 * n + i] = warn;
    }
    wg_barrier();            // the row has read the LDS results: the next table / intron may clear them
    }       // tables
    PROF_MARK(5);
    recA = recB; recB = recC; dA = dB; tcA = tcB;
    }
#ifdef STATS_PROF
    if (tid == 0 && prof_slot + blockIdx.x < kProfMax) {
        for (int k = 0; k < 6; k++) g_prof[prof_slot + blockIdx.x][k] = prof_acc[k];
        g_prof[prof_slot + blockIdx.x][6] = k_end - k_begin;
    }
#endif
}

// One thread per intron: the row tail of intron_stats_kernel (same expressions, same order) on the finished junction
// tables.  Reads Depth / flank medians back from the table, writes the five counters, IRratio and the warning.
__global__ __launch_bounds__(256) void ir_rows_kernel(IntronTables T) {
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= T.n_intron) return;
    const IntronRec R = T.recs[k];
    const bool valid = (R.flags & 1u) != 0u;
    const bool counters = valid && !(T.ablate & 2u);
    uint32_t cnt[5][2];
    {
        const int32_t ix[5] = {R.sp_s, R.sp_e, R.us, R.ue, R.kj};
        const uint32_t* base[5] = {T.sp_cnt, T.sp_cnt, T.left_cnt, T.right_cnt, T.known_cnt};
        const uint64_t n[5] = {T.n_sp, T.n_sp, T.n_ustart, T.n_uend, T.n_kj};
#pragma unroll
        for (int w = 0; w < 5; w++)
#pragma unroll
            for (int h = 0; h < 2; h++) cnt[w][h] = (counters && ix[w] >= 0) ? base[w][(uint64_t)h * n[w] + (uint64_t)ix[w]] : 0u;
    }
    // novel junctions on (Start, End), per strand
    uint32_t nov[2] = {0u, 0u};
    if (counters && R.kj < 0) {
        const uint64_t span = R.g_end - R.g_start;
        if (span < (1ull << kJuncSpanBits) && R.g_start < kJuncMaxGlobal && T.n_unique > 0) {
            const uint64_t key0 = (R.g_start << kJuncLeftShift) | (span << 1);
            uint64_t lo_i = 0, hi_i = T.n_unique;
            while (lo_i < hi_i) {
                const uint64_t mid = lo_i + ((hi_i - lo_i) >> 1);
                if (T.ukeys[mid] < key0) lo_i = mid + 1; else hi_i = mid;
            }
            for (uint64_t u = lo_i; u < T.n_unique && u < lo_i + 2 && (T.ukeys[u] >> 1) == (key0 >> 1); u++) nov[T.ukeys[u] & 1ull] += T.ucount[u];
        }
    }
    const int plane_strand = (((R.flags & 2u) != 0) == (T.directionality > 0)) ? 1 : 0;
    const int n_tables = T.dual ? 2 : 1;
    const uint64_t n = T.n_intron, i = R.row;
    for (int t = 0; t < n_tables; t++) {
        const int plane = T.dual ? (t == 0 ? 2 : plane_strand) : (T.directionality != 0 ? plane_strand : 2);
        double* __restrict__ out_f64 = t == 0 ? T.out_f64 : T.out_f64_b;
        uint32_t* __restrict__ out_u32 = t == 0 ? T.out_u32 : T.out_u32_b;
        auto pick = [&](const uint32_t* c) -> uint32_t { return plane == 2 ? c[0] + c[1] : (plane == 1 ? c[1] : c[0]); };
        const uint32_t spl = pick(cnt[0]), spr = pick(cnt[1]), jl = pick(cnt[2]), jr = pick(cnt[3]);
        const uint32_t je = pick(cnt[4]) + pick(nov);
        const double depth = out_f64[1 * n + i], flank0 = out_f64[5 * n + i], flank1 = out_f64[6 * n + i];
        const uint32_t smax = jl > jr ? jl : jr;
        const double den = depth + (double)smax;
        const double irratio = den > 0.0 ? depth / den : 0.0;
        const double sp_max = (double)(spl > spr ? spl : spr);
        const double fl_max = flank0 > flank1 ? flank0 : flank1;
        const double maxflank = sp_max > fl_max ? sp_max : fl_max;
        uint32_t warn;
        if ((double)je + depth < T.warn_lowcover_min) warn = 1;
        else if ((double)je < T.warn_lowsplice_min) warn = 2;
        else if ((double)je * T.warn_minor_ratio < (double)smax) warn = 3;
        else if (T.warn_nonuniform_and ? (maxflank > depth + T.warn_nonuniform_abs && maxflank > depth * T.warn_nonuniform_rel)
                                       : (maxflank > depth * T.warn_nonuniform_rel + T.warn_nonuniform_abs)) warn = 4;
        else warn = 0;
        out_f64[7 * n + i] = irratio;
        out_u32[0 * n + i] = spl;
        out_u32[1 * n + i] = spr;
        out_u32[2 * n + i] = jl;
        out_u32[3 * n + i] = jr;
        out_u32[4 * n + i] = je;
        out_u32[5 * n + i] = warn;
    }
}

// the statistics kernels' table block in device memory (launch_intron_stats)
__global__ void store_intron_tables_kernel(IntronTables T, IntronTables* dst) { *dst = T; }

}   // namespace

void launch_ir_rows(const IntronTables& T, hipStream_t s) {
    if (!T.n_intron) return;
    hipLaunchKernelGGL(ir_rows_kernel, dim3((T.n_intron + 255u) / 256u), dim3(256), 0, s, T);
}

// T.recs is laid out by size class (kernels.h: StatsClasses) and, inside a class, by workgroup: workgroup w of class c
// walks records [wg_off[c][w], wg_off[c][w + 1]) (the host balanced the lists).  The 64-thread class keeps its flank
// windows in a 64-entry LDS buffer: the host puts every intron into the 256-thread class when flank_len is larger.
void launch_intron_stats(const IntronTables& T, IntronTables* d_T, const StatsClasses& K, hipStream_t s) {
    hipLaunchKernelGGL(store_intron_tables_kernel, dim3(1), dim3(1), 0, s, T, d_T);
    if (T.dual) {       // both tables in one pass (T.directionality != 0 or the vote; outputs out_* and out_*_b)
        if (K.n_wg[0]) hipLaunchKernelGGL((intron_stats_kernel<256, kStatsMaxFlank, STATS_INFLIGHT_BIG, 10, 2>), dim3(K.n_wg[0]), dim3(256), 0, s, d_T, K.wg_off[0], 0u);
        if (K.n_wg[1]) hipLaunchKernelGGL((intron_stats_kernel<64, kStatsSmallFlank, STATS_INFLIGHT_SMALL, STATS_BITS_SMALL, 2>), dim3(K.n_wg[1]), dim3(64), 0, s, d_T, K.wg_off[1], K.n_wg[0]);
    } else {
        if (K.n_wg[0]) hipLaunchKernelGGL((intron_stats_kernel<256, kStatsMaxFlank, STATS_INFLIGHT_BIG, 10, 1>), dim3(K.n_wg[0]), dim3(256), 0, s, d_T, K.wg_off[0], 0u);
        if (K.n_wg[1]) hipLaunchKernelGGL((intron_stats_kernel<64, kStatsSmallFlank, STATS_INFLIGHT_SMALL, STATS_BITS_SMALL, 1>), dim3(K.n_wg[1]), dim3(64), 0, s, d_T, K.wg_off[1], K.n_wg[0]);
    }
#ifdef STATS_PROF
    {
        static std::vector<uint32_t> h((size_t)kProfMax * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_prof), h.size() * 4);
        uint32_t at = 0;
        for (int c = 0; c < 2; c++) {
            double a[8] = {0};
            for (uint32_t k = at; k < at + K.n_wg[c] && k < kProfMax; k++)
                for (int j = 0; j < 7; j++) a[j] += h[(size_t)k * 8 + j];
            at += K.n_wg[c];
            if (a[6] > 0)
                fprintf(stderr, "stats_prof class %d n=%.0f wgs=%u ticks/intron: zero %.0f segments %.0f sync %.0f medians %.0f queries %.0f row+sync %.0f\n", c, a[6], K.n_wg[c],
                        a[0] / a[6], a[1] / a[6], a[2] / a[6], a[3] / a[6], a[4] / a[6], a[5] / a[6]);
        }
    }
#endif
}

}   // namespace dev
}   // namespace irf
