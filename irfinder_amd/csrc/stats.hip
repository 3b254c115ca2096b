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
// Shape.  200 k introns of very different sizes (median 1.7 k measured bases, tail to 0.5 M): a short intron is
// a chain of dependent round trips (record -> block descriptors -> planes), a long one is a stream.  So
//   * the host sorts the introns into three size classes, longest first inside a class, and each class gets its own
//     workgroup size (1024 / 256 / 64 threads): a CU holds 32 short introns at once instead of 8;
//   * everything the row needs besides depth sits in ONE 64-byte record per intron (scalar loads), and the counters
//     of the row's tail are fetched at the top, under the depth pass;
//   * the measured blocks of an intron are flattened into one list of segments: lane b of every wave keeps the
//     descriptor of block b in registers (no LDS, no barrier), a ballot finds the block of a segment, and a wave has
//     kInFlight segments of loads outstanding before it touches the first;
//   * depth is piecewise constant, so a segment that holds one value 256 times (most do) costs ONE LDS atomic, and
//     mixed segments aggregate equal values per wave before touching the histogram.
#include "kernels.h"

namespace irf {
namespace dev {

namespace {

constexpr int kBinBits = 10;
constexpr int kBins = 1 << kBinBits;
constexpr int kQueries = 8;
constexpr int kSeg = 256;           // positions per carry segment (kernels.h: kDepthSegment)
constexpr uint32_t kMaxBlk = 64;    // measured blocks per descriptor chunk: one per lane

struct Ctx {
    const int32_t* d0;              // difference planes
    const int32_t* d1;
    const uint32_t* c0;             // carries, one per 256-position segment
    const uint32_t* c1;
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

__device__ __forceinline__ uint32_t lane_u32(uint32_t v, uint32_t src) {          // src is wave-uniform
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src);
}
__device__ __forceinline__ uint64_t lane_u64(uint64_t v, uint32_t src) {
    return (uint64_t)lane_u32((uint32_t)v, src) | ((uint64_t)lane_u32((uint32_t)(v >> 32), src) << 32);
}

__device__ __forceinline__ Seg load_seg(const Ctx& C, uint64_t k, uint32_t lane) {
    Seg s;
    const uint64_t e0 = k * kSeg + lane * 4u;
    if (C.plane == 2) {
        const int4 u = *reinterpret_cast<const int4*>(C.d0 + e0);
        const int4 v = *reinterpret_cast<const int4*>(C.d1 + e0);
        s.a.x = u.x + v.x; s.a.y = u.y + v.y; s.a.z = u.z + v.z; s.a.w = u.w + v.w;
        s.carry = C.c0[k] + C.c1[k];
    } else if (C.plane == 1) {
        s.a = *reinterpret_cast<const int4*>(C.d1 + e0);
        s.carry = C.c1[k];
    } else {
        s.a = *reinterpret_cast<const int4*>(C.d0 + e0);
        s.carry = C.c0[k];
    }
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
template <int kWaves>
__device__ __forceinline__ void block_excl_scan(uint32_t tot, unsigned long long tsum, uint32_t& pre,
                                                unsigned long long& pre_sum, uint32_t* s_wc, unsigned long long* s_ws) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t c = tot;
    unsigned long long s = tsum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t oc = __shfl_up(c, d);
        unsigned long long os = __shfl_up(s, d);
        if ((int)lane >= d) { c += oc; s += os; }
    }
    if (lane == 63) { s_wc[wave] = c; s_ws[wave] = s; }
    __syncthreads();
    uint32_t bc = 0;
    unsigned long long bs = 0;
    for (int w = 0; w < kWaves; w++)
        if (w < (int)wave) { bc += s_wc[w]; bs += s_ws[w]; }
    pre = bc + c - tot;
    pre_sum = bs + s - tsum;
    __syncthreads();
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

template <int kThreads, int kMaxFlank, int kInFlight>
__global__ __launch_bounds__(kThreads) void intron_stats_kernel(IntronTables T, uint32_t first_rec) {
    constexpr int kWaves = kThreads / 64;
    constexpr int kBinsPerThread = kBins / kThreads > 0 ? kBins / kThreads : 1;
    __shared__ uint32_t s_cnt[kBins];
    __shared__ uint32_t s_win[2][kMaxFlank];               // depths of the two flank windows
    __shared__ uint32_t s_med[4];                          // per window: the two order statistics its median reads
    __shared__ uint32_t s_wc[kWaves];
    __shared__ unsigned long long s_ws[kWaves];
    __shared__ uint32_t s_max, s_overflow, s_sel_bin, s_sel_cnt;
    __shared__ unsigned long long s_total, s_nz, s_acc;
    __shared__ uint32_t s_qval[kQueries];
    __shared__ unsigned long long s_qsum[kQueries];
    __shared__ unsigned long long s_qrank[kQueries];
    __shared__ uint32_t s_qneed[kQueries];
    __shared__ uint32_t s_prem[3];

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const IntronRec R = T.recs[first_rec + blockIdx.x];        // wave-uniform: scalar loads
    Ctx C;
    C.d0 = T.diff;
    C.d1 = T.diff + T.plane_stride;
    C.c0 = T.carry;
    C.c1 = T.carry + T.plane_stride / kSeg;
    const bool valid = R.valid != 0 && !(T.ablate & 1u);     // false: chromosome absent from the BAM header / not owned => depth 0 everywhere
    C.plane = 2;
    if (T.directionality != 0) C.plane = ((R.plus != 0) == (T.directionality > 0)) ? 1 : 0;
    const int plane = C.plane;

    // The counters of the row's tail, fetched now (uniform addresses: they land in scalar registers while the depth
    // pass runs).  They were finalised by the kernels before this one.
    uint32_t spl = 0, spr = 0, jl = 0, jr = 0, je = 0;
    if (R.valid != 0 && !(T.ablate & 2u)) {
        auto both = [&](const uint32_t* cnt, uint64_t n, int32_t idx) -> uint32_t {
            if (idx < 0) return 0u;
            return plane == 2 ? cnt[idx] + cnt[n + (uint64_t)idx] : cnt[(uint64_t)plane * n + (uint64_t)idx];
        };
        spl = both(T.sp_cnt, T.n_sp, R.sp_s);
        spr = both(T.sp_cnt, T.n_sp, R.sp_e);
        jl = both(T.left_cnt, T.n_ustart, R.us);
        jr = both(T.right_cnt, T.n_uend, R.ue);
        je = both(T.known_cnt, T.n_kj, R.kj);      // (Start,End) is a ref-sj junction: its reads were counted in place
    }

    const uint32_t b0 = R.blk_off, nb = R.n_blk;
    const unsigned long long N = R.measured;
    const uint64_t gS = R.g_start, gE = R.g_end;

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

    for (int k = tid; k < kBins; k += kThreads) s_cnt[k] = 0;
    if (tid == 0) { s_max = 0; s_overflow = 0; s_total = 0; s_nz = 0; s_acc = 0; }
    if (tid < 4) s_med[tid] = 0;
    __syncthreads();

    // ---------------- pass A: histogram (clipped) + window capture ----------------
    // (total, non-zero count and maximum are read off the histogram afterwards)
    uint32_t nwin[2] = {0, 0};
    if (valid) {
        bool ovf = false;
        if (!T.flank_use_measured) {
            // raw-coordinate windows: the engine keeps the buckets under them, contiguous in the plane
            for (int wi = 0; wi < 2; wi++) {
                if (w0[wi] >= w1[wi]) continue;
                const uint64_t len = w1[wi] - w0[wi];
                const uint64_t cs = plane_coord(T.pmap, w0[wi]);
                uint32_t* dst = s_win[wi];
                for_each_depth<kWaves>(C, cs, cs + len, [&](uint64_t x, uint32_t v, bool ok) { if (ok) dst[(uint32_t)(x - cs)] = v; });
                nwin[wi] = (uint32_t)len;
            }
        }
        for (uint32_t cb = 0; cb < nb; cb += kMaxBlk) {
            // lane b keeps measured block cb + b: plane start, length, its slice of the flattened segment list, and
            // the parts of the two flank windows that fall inside it (every wave builds the same registers)
            const uint32_t nbc = nb - cb < kMaxBlk ? nb - cb : kMaxBlk;
            const bool has = lane < nbc;
            const BlkRec br = T.blks[b0 + cb + (has ? lane : 0u)];
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
                Seg sg[kInFlight];
                uint32_t sb[kInFlight];
                uint64_t sk[kInFlight];
#pragma unroll
                for (int j = 0; j < kInFlight; j++) {
                    const uint32_t s = s0 + j * kWaves;
                    const uint32_t sc = s < S ? s : S - 1u;           // past the end: load the last segment again, unused
                    sb[j] = (uint32_t)__popcll(__ballot(d_seg_end <= sc));        // blocks that end before segment sc
                    sk[j] = lane_u64(d_bc, sb[j]) / kSeg + (sc - lane_u32(d_seg_start, sb[j]));
                    sg[j] = load_seg(C, sk[j], lane);
                }
#pragma unroll
                for (int j = 0; j < kInFlight; j++) {
                    if (s0 + j * kWaves >= S) continue;                // wave-uniform
                    const uint32_t b = sb[j];
                    const uint64_t bc = lane_u64(d_bc, b), be_c = bc + lane_u32(d_bl, b);
                    const uint32_t wn0 = lane_u32(d_wn[0], b), wn1 = lane_u32(d_wn[1], b);
                    const uint64_t ws0 = wn0 ? bc + lane_u32(d_wlo[0], b) : 0ull, we0 = ws0 + wn0;
                    const uint64_t ws1 = wn1 ? bc + lane_u32(d_wlo[1], b) : 0ull, we1 = ws1 + wn1;
                    const uint32_t at0 = lane_u32(d_wat[0], b), at1 = lane_u32(d_wat[1], b);
                    emit_seg(sg[j], sk[j], lane, [&](uint64_t seg0, uint64_t e0, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
                        const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)v0);
                        const bool inner = seg0 >= bc && seg0 + kSeg <= be_c &&                         // whole segment measured
                                           !(seg0 < we0 && seg0 + kSeg > ws0) && !(seg0 < we1 && seg0 + kSeg > ws1);
                        const bool flat = v0 == first && v1 == first && v2 == first && v3 == first;
                        if (inner && __ballot(flat) == ~0ull) {
                            // depth is piecewise constant: most segments hold one value 256 times -> one LDS atomic
                            if (lane == 0) { if (first < (uint32_t)kBins) atomicAdd(&s_cnt[first], (uint32_t)kSeg); else ovf = true; }
                            return;
                        }
                        const uint32_t vv[4] = {v0, v1, v2, v3};
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const uint64_t x = e0 + q;
                            const uint32_t v = vv[q];
                            const bool ok = x >= bc && x < be_c;
                            // wave-aggregated histogram update: lanes that share the first valid lane's value add once
                            const uint64_t act = __ballot(ok);
                            if (!act) continue;
                            const uint32_t lead = __shfl(v, __ffsll((long long)act) - 1);
                            const uint64_t same = __ballot(ok && v == lead);
                            if (ok) {
                                if (v == lead) {
                                    if (lane == (uint32_t)(__ffsll((long long)same) - 1)) {
                                        if (v < (uint32_t)kBins) atomicAdd(&s_cnt[v], (uint32_t)__popcll(same)); else ovf = true;
                                    }
                                } else {
                                    if (v < (uint32_t)kBins) atomicAdd(&s_cnt[v], 1u); else ovf = true;
                                }
                                if (x >= ws0 && x < we0) s_win[0][at0 + (uint32_t)(x - ws0)] = v;
                                if (x >= ws1 && x < we1) s_win[1][at1 + (uint32_t)(x - ws1)] = v;
                            }
                        }
                    });
                }
            }
        }
        if (ovf) s_overflow = 1;
    } else {
        // all-zero depth: N zeros in the histogram, all-zero windows of the right size
        if (tid == 0) s_cnt[0] = (uint32_t)N;
        for (int wi = 0; wi < 2; wi++) {
            uint32_t n = 0;
            if (w0[wi] < w1[wi]) {
                if (T.flank_use_measured) {
                    for (uint32_t b = b0; b < b0 + nb; b++) {
                        const uint64_t bs = T.blks[b].gstart, be = bs + T.blks[b].len;
                        const uint64_t s = bs > w0[wi] ? bs : w0[wi], e = be < w1[wi] ? be : w1[wi];
                        if (s < e) n += (uint32_t)(e - s);
                    }
                } else {
                    n = (uint32_t)(w1[wi] - w0[wi]);
                }
            }
            for (uint32_t k = tid; k < n; k += kThreads) s_win[wi][k] = 0;
            nwin[wi] = n;
        }
    }
    __syncthreads();
    const bool overflow = s_overflow != 0;
    if (overflow) {
        // depth beyond the single-level histogram somewhere: maximum, total and non-zero count by a pass of their own
        uint32_t vmax = 0;
        unsigned long long tsum = 0, nz = 0;
        for (uint32_t b = b0; b < b0 + nb; b++) {
            const uint64_t bc = T.blks[b].cstart;
            for_each_depth<kWaves>(C, bc, bc + T.blks[b].len, [&](uint64_t, uint32_t v, bool ok) {
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
        s_nz = N - s_cnt[0];
    }

    // ---------------- flank medians: rank of every window element by counting ----------------
    PctRanks fq[2];
    fq[0] = percentile_ranks(nwin[0], 50u, T.pct_interpolate != 0);
    fq[1] = percentile_ranks(nwin[1], 50u, T.pct_interpolate != 0);
    {
        const uint32_t n0 = nwin[0], n1 = nwin[1];
        for (uint32_t e = tid; e < n0 + n1; e += kThreads) {
            const int wi = e >= n0 ? 1 : 0;
            const uint32_t* w = s_win[wi];
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

    // ---------------- the rank queries ----------------
    // q0,q1: sums below ranks lo / hi of the trimmed mean; q2..q7: values for p25, p50, p75 (ranks k-1 and k)
    const unsigned long long skip = N * (unsigned long long)(100u - T.trim_centre_pct) / 200ull;
    const unsigned long long lo = skip, hi = N - skip;
    if (tid == 0) {
        s_qrank[0] = lo; s_qneed[0] = (N > 0 && lo > 0 && lo < N) ? 1u : 0u;
        s_qrank[1] = hi; s_qneed[1] = (N > 0 && hi > 0 && hi < N) ? 1u : 0u;
        const uint32_t pcts[3] = {25u, 50u, 75u};
        for (int p = 0; p < 3; p++) {
            const PctRanks q = percentile_ranks(N, pcts[p], T.pct_interpolate != 0);
            s_qrank[2 + 2 * p] = q.ra; s_qneed[2 + 2 * p] = q.need_a ? 1u : 0u;
            s_qrank[3 + 2 * p] = q.rb; s_qneed[3 + 2 * p] = q.need_b ? 1u : 0u;
            s_prem[p] = q.rem;
        }
    }
    if (tid < kQueries) { s_qval[tid] = 0; s_qsum[tid] = 0; }
    __syncthreads();

    if (N > 0 && !overflow) {
        // single level: bins are the depths themselves
        uint32_t c[kBinsPerThread];
        uint32_t tot = 0;
        unsigned long long tsum = 0;
#pragma unroll
        for (int j = 0; j < kBinsPerThread; j++) {
            c[j] = s_cnt[tid * kBinsPerThread + j];
            tot += c[j];
            tsum += (unsigned long long)c[j] * (unsigned long long)(tid * kBinsPerThread + j);
        }
        uint32_t pre;
        unsigned long long pre_sum;
        block_excl_scan<kWaves>(tot, tsum, pre, pre_sum, s_wc, s_ws);
        if (tid == kThreads - 1) s_total = pre_sum + tsum;       // sum of all depths
#pragma unroll
        for (int q = 0; q < kQueries; q++) {
            if (!s_qneed[q]) continue;
            const unsigned long long r = s_qrank[q];
            if (r >= pre && r < (unsigned long long)pre + tot) {
                unsigned long long run = pre, s = pre_sum;
#pragma unroll
                for (int j = 0; j < kBinsPerThread; j++) {
                    const unsigned long long bin = (unsigned long long)(tid * kBinsPerThread + j);
                    if (r < run + c[j]) { s_qval[q] = (uint32_t)bin; s_qsum[q] = s + bin * (r - run); break; }
                    run += c[j];
                    s += bin * c[j];
                }
            }
        }
        __syncthreads();
    } else if (N > 0) {
        // multi-level radix select, one descent per query; a sum query (q < 2) then adds the values below the selected one
        __syncthreads();
        int bits = 32 - __clz((int)s_max);
        int levels = (bits + kBinBits - 1) / kBinBits;
        if (levels < 1) levels = 1;
        for (int q = 0; q < kQueries; q++) {
            if (!s_qneed[q]) continue;                     // uniform across the block
            unsigned long long r = s_qrank[q];
            uint32_t prefix = 0;
            for (int lv = 0; lv < levels; lv++) {
                const int shift = kBinBits * (levels - 1 - lv);
                for (int k = tid; k < kBins; k += kThreads) s_cnt[k] = 0;
                __syncthreads();
                for (uint32_t b = b0; b < b0 + nb; b++) {
                    const uint64_t bc = T.blks[b].cstart;
                    for_each_depth<kWaves>(C, bc, bc + T.blks[b].len, [&](uint64_t, uint32_t v, bool ok) {
                        const bool in = ok && (lv == 0 || (v >> (shift + kBinBits)) == prefix);
                        if (in) atomicAdd(&s_cnt[(v >> shift) & (uint32_t)(kBins - 1)], 1u);
                    });
                }
                __syncthreads();
                uint32_t c[kBinsPerThread];
                uint32_t tot = 0;
#pragma unroll
                for (int j = 0; j < kBinsPerThread; j++) {
                    c[j] = s_cnt[tid * kBinsPerThread + j];
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
                __syncthreads();
                prefix = (prefix << kBinBits) | s_sel_bin;
                r -= s_sel_cnt;
                __syncthreads();
            }
            // prefix = the value at the rank; r = the rank's position among the copies of that value
            unsigned long long below = 0;
            if (q < 2) {
                unsigned long long part = 0;
                for (uint32_t b = b0; b < b0 + nb; b++) {
                    const uint64_t bc = T.blks[b].cstart;
                    for_each_depth<kWaves>(C, bc, bc + T.blks[b].len, [&](uint64_t, uint32_t v, bool ok) {
                        if (ok && v < prefix) part += v;
                    });
                }
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
                if (lane == 0 && part) atomicAdd(&s_acc, part);
                __syncthreads();
                below = s_acc;
                __syncthreads();
                if (tid == 0) s_acc = 0;
            }
            if (tid == 0) { s_qval[q] = prefix; s_qsum[q] = below + (unsigned long long)prefix * r; }
        }
        __syncthreads();
    } else {
        __syncthreads();
    }

    // ---------------- the row ----------------
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
                if (s_qneed[3 + 2 * p]) pv[p] = vk + ((double)s_prem[p] / 100.0) * ((double)s_qval[3 + 2 * p] - vk);
                else pv[p] = vk;
            }
        }
        if (R.valid != 0 && !(T.ablate & 2u) && R.kj < 0) {
            // (Start,End) is not in ref-sj: look it up among the novel junctions
            const uint64_t span = gE - gS;
            if (span < (1ull << kJuncSpanBits) && gS < kJuncMaxGlobal && T.n_unique > 0) {
                const uint64_t key0 = (gS << kJuncLeftShift) | (span << 1);
                uint64_t lo_i = 0, hi_i = T.n_unique;
                while (lo_i < hi_i) {
                    const uint64_t mid = lo_i + ((hi_i - lo_i) >> 1);
                    if (T.ukeys[mid] < key0) lo_i = mid + 1; else hi_i = mid;
                }
                for (uint64_t u = lo_i; u < T.n_unique && u < lo_i + 2 && (T.ukeys[u] >> 1) == (key0 >> 1); u++) {
                    const int d = (int)(T.ukeys[u] & 1ull);
                    if (plane == 2 || plane == d) je += T.ucount[u];
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
        else if (maxflank > depth * T.warn_nonuniform_rel + T.warn_nonuniform_abs) warn = 4;
        else warn = 0;
        const uint64_t n = T.n_intron;
        const uint64_t i = R.row;
        T.out_f64[0 * n + i] = coverage;
        T.out_f64[1 * n + i] = depth;
        T.out_f64[2 * n + i] = pv[0];
        T.out_f64[3 * n + i] = pv[1];
        T.out_f64[4 * n + i] = pv[2];
        T.out_f64[5 * n + i] = flank0;
        T.out_f64[6 * n + i] = flank1;
        T.out_f64[7 * n + i] = irratio;
        T.out_u32[0 * n + i] = spl;
        T.out_u32[1 * n + i] = spr;
        T.out_u32[2 * n + i] = jl;
        T.out_u32[3 * n + i] = jr;
        T.out_u32[4 * n + i] = je;
        T.out_u32[5 * n + i] = warn;
    }
}

}   // namespace

// T.recs is sorted by size class (kernels.h: StatsClasses), longest intron first inside a class; every class is one
// launch with its own workgroup size.  The small workgroups keep their flank windows in a 64-entry LDS buffer: the
// host puts every intron into the 256-thread class when flank_len is larger.
void launch_intron_stats(const IntronTables& T, const StatsClasses& K, hipStream_t s) {
    if (K.count[0]) hipLaunchKernelGGL((intron_stats_kernel<1024, kStatsMaxFlank, 4>), dim3(K.count[0]), dim3(1024), 0, s, T, K.first[0]);
    if (K.count[1]) hipLaunchKernelGGL((intron_stats_kernel<256, kStatsMaxFlank, 4>), dim3(K.count[1]), dim3(256), 0, s, T, K.first[1]);
    if (K.count[2]) hipLaunchKernelGGL((intron_stats_kernel<64, kStatsSmallFlank, 4>), dim3(K.count[2]), dim3(64), 0, s, T, K.first[2]);
}

}   // namespace dev
}   // namespace irf
