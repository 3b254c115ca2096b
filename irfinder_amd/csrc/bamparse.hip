// bamparse.hip — inflated BAM stream -> structure-of-arrays fragment batch, on the device (gfx950).
//
// Replaces ⟨upstream⟩ BAM2blocks::processAll / processPair / processSingle / cigar2block (BAM2blocks.cpp) for input
// that is already in HBM (SURVEY §8 f N1); the semantics are those of the host reader (host/bam_reader.cpp), which the
// parity tests compare against column by column.  Upstream's loop is sequential twice over: a record is found only
// through the block_size of its predecessor, and whether a record is a single read, the first or the second mate of a
// pair depends on the records before it.  Both chains are resolved exactly, in parallel:
//   1. record starts: speculate per 4 KiB segment, verify against the predecessor's walk (dreader_kernels.h: ChainBufs);
//   2. pairing: "the NEXT record is the mate" is a two-state automaton (free / waiting for the mate); a record without
//      the paired flag always leaves it free, a paired one toggles it, so the state before record i is the parity of
//      the run of paired records that ends at i - 1 — a running maximum of "last unpaired index", per tile + carry;
//   3. fragments: every record that completes a fragment computes its block count, two exclusive scans place the
//      fragments in record order (so the batch equals the host reader's), a second pass writes them.
#include "dreader_kernels.h"

namespace irf {
namespace dev {

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// could a record start at o?  (speculation only: verified afterwards)
__device__ bool plausible_record(const uint8_t* buf, uint64_t o, uint64_t total, int32_t n_ref, uint32_t* block_size) {
    if (o + 36 > total) return false;
    const uint8_t* p = buf + o;
    const uint32_t bs = ld32u(p);
    if (bs < 32 || bs > kMaxBamBlockSize) return false;
    const int32_t ref_id = (int32_t)ld32u(p + 4), pos = (int32_t)ld32u(p + 8);
    if (ref_id < -1 || ref_id >= n_ref || pos < -1) return false;
    const uint32_t w = ld32u(p + 12), w2 = ld32u(p + 16);
    const uint32_t l_read_name = w & 0xffu, n_cigar = w2 & 0xffffu;
    const int32_t l_seq = (int32_t)ld32u(p + 20);
    const int32_t next_ref = (int32_t)ld32u(p + 24), next_pos = (int32_t)ld32u(p + 28);
    if (l_read_name == 0 || l_seq < 0 || next_ref < -1 || next_ref >= n_ref || next_pos < -1) return false;
    const uint64_t need = 32ull + l_read_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (need > bs) return false;
    const uint64_t nul = o + 36 + l_read_name - 1;
    if (nul < total && buf[nul] != 0) return false;
    *block_size = bs;
    return true;
}

// walks complete records from `from` while they start before seg_end: returns the first start at or after seg_end
// (or the start of the partial record at the end of the buffer), kNoRecord if the chain breaks
__device__ uint64_t walk_segment(const uint8_t* buf, uint64_t from, uint64_t seg_end, uint64_t total, uint32_t* count) {
    uint32_t n = 0;
    uint64_t o = from;
    if (o != kNoRecord) {
        while (o < seg_end) {
            if (o + 4 > total) break;
            const uint32_t bs = ld32u(buf + o);
            if (bs < 32 || bs > kMaxBamBlockSize) { o = kNoRecord; break; }
            if (o + 4 + (uint64_t)bs > total) break;
            n++;
            o += 4 + (uint64_t)bs;
        }
    }
    *count = n;
    return o;
}

__device__ __forceinline__ uint64_t segment_end(const ChainBufs& C, uint32_t i) {
    return i + 1 == C.n_seg ? ~0ull - 1 : C.start + (uint64_t)(i + 1) * kChainSegment;
}

__global__ __launch_bounds__(kThreads) void chain_guess_kernel(ChainBufs C) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= C.n_seg) return;
    const uint64_t beg = C.start + (uint64_t)i * kChainSegment;
    const uint64_t lim = min(beg + (uint64_t)kChainSegment, C.total);
    uint64_t g = kNoRecord;
    if (i == 0) {
        g = C.start;
    } else {
        for (uint64_t o = beg; o < lim; o++) {
            uint32_t bs;
            if (!plausible_record(C.buf, o, C.total, C.n_ref, &bs)) continue;
            uint64_t o2 = o + 4 + (uint64_t)bs;
            bool ok = true;
            for (int k = 0; k < 2 && ok; k++) {
                if (o2 + 36 > C.total) break;              // cannot be checked any further
                uint32_t bs2 = 0;
                ok = plausible_record(C.buf, o2, C.total, C.n_ref, &bs2);
                o2 += 4 + (uint64_t)bs2;
            }
            if (ok) { g = o; break; }
        }
    }
    uint32_t n;
    const uint64_t e = walk_segment(C.buf, g, segment_end(C, i), C.total, &n);
    C.guess[i] = g;
    C.exit_[i] = e;
    C.count[i] = n;
}

// guess[i + 1] must be what segment i's walk arrives at; otherwise that walk is carried on until a guess is confirmed.
// Every thread re-walks its own segment from the guess it reads (exit_ / count may have been left behind by a walker
// that lost a race), so after a launch the guesses of a prefix one segment longer are final: threads inside the prefix
// write nothing, walkers only ever write segments ahead of them.  A launch that writes nothing has checked everything.
// Only a thread whose own guess is what its predecessor arrived at (as of the previous launch) carries its walk on: a
// wrongly guessed start is a chain through arbitrary bytes, whose "block sizes" would send the walker across thousands
// of segments, overwriting good guesses on the way.  Segment 0 is always confirmed, and a launch that changes nothing
// confirms segment i + 1 from segment i, so neither the induction nor the end condition changes.
__global__ __launch_bounds__(kThreads) void chain_verify_kernel(ChainBufs C) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= C.n_seg) return;
    bool touched = false;
    uint32_t n;
    const uint64_t g = C.guess[i];
    const bool confirmed = i == 0 || C.exit_[i - 1] == g;
    uint64_t cur = walk_segment(C.buf, g, segment_end(C, i), C.total, &n);
    if (C.exit_[i] != cur || C.count[i] != n) {
        C.exit_[i] = cur;
        C.count[i] = n;
        touched = true;
    }
    uint32_t j = i + 1;
    // (a broken chain is not carried on: the break is reported by chain_emit, what follows it is never used)
    while (confirmed && j < C.n_seg && cur != kNoRecord && C.guess[j] != cur) {
        const uint64_t e = walk_segment(C.buf, cur, segment_end(C, j), C.total, &n);
        C.guess[j] = cur;
        C.exit_[j] = e;
        C.count[j] = n;
        cur = e;
        j++;
        touched = true;
    }
    if (touched) *C.changed = 1u;
}

__global__ __launch_bounds__(kThreads) void chain_emit_kernel(ChainBufs C, const uint32_t* __restrict__ seg_base, RecordIndex R) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= C.n_seg) return;
    uint64_t o = C.guess[i];
    if (o == kNoRecord) return;
    const uint64_t seg_end = segment_end(C, i);
    uint32_t idx = seg_base[i];
    while (o < seg_end) {
        if (o + 4 > C.total) break;
        const uint32_t bs = ld32u(C.buf + o);
        if (bs < 32 || bs > kMaxBamBlockSize) {
            atomicMin(R.err, (unsigned long long)(o << 2 | 1ull));
            break;
        }
        if (o + 4 + (uint64_t)bs > C.total) break;
        R.rec_off[idx] = o;
        R.rec_nf[idx] = ld32u(C.buf + o + 16);
        idx++;
        o += 4 + (uint64_t)bs;
    }
}

// ---- pairing state ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_max(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, (uint32_t)__shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d);
        if ((int)lane >= d) v = max(v, o);
    }
    return v;
}

constexpr uint32_t kPerThread = kPairTile / kThreads;     // 4 consecutive records per thread

__global__ __launch_bounds__(kThreads) void pair_tile_last_kernel(const uint32_t* __restrict__ rec_nf, uint32_t n_rec,
                                                                  uint32_t* __restrict__ tile_last) {
    __shared__ uint32_t s_w[kThreads / 64];
    const uint32_t i0 = blockIdx.x * kPairTile + threadIdx.x * kPerThread;
    uint32_t m = 0;
    for (uint32_t k = 0; k < kPerThread; k++) {
        const uint32_t i = i0 + k;
        if (i < n_rec && !((rec_nf[i] >> 16) & 1u)) m = i + 1;
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = m;
    wg_barrier();
    if (threadIdx.x == 0) tile_last[blockIdx.x] = max(max(s_w[0], s_w[1]), max(s_w[2], s_w[3]));
}

// one workgroup: tile_carry[t] = max over tiles before t
__global__ __launch_bounds__(kThreads) void pair_tile_scan_kernel(const uint32_t* __restrict__ tile_last, uint32_t n_tiles,
                                                                  uint32_t* __restrict__ tile_carry) {
    __shared__ uint32_t s_w[kThreads / 64];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < n_tiles; t0 += kThreads) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < n_tiles ? tile_last[t] : 0u;
        const uint32_t inc = wave_inclusive_max(v, lane);
        if (lane == 63) s_w[wave] = inc;
        wg_barrier();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wave; w++) before = max(before, s_w[w]);
        uint32_t excl = __shfl_up(inc, 1);
        if (lane == 0) excl = 0;
        if (t < n_tiles) tile_carry[t] = max(before, excl);
        carry = max(max(carry, max(s_w[0], s_w[1])), max(s_w[2], s_w[3]));
        wg_barrier();
    }
}

// ---- fragment assembly ----------------------------------------------------------------------------------------------
struct Rec {
    int32_t ref_id, pos;
    uint32_t flag, n_cigar;
    int32_t l_seq;
    bool overrun;
    const uint8_t* cigar;
};

__device__ __forceinline__ Rec load_rec(const uint8_t* buf, uint64_t off) {
    const uint8_t* p = buf + off;
    Rec r;
    const uint32_t bs = ld32u(p);
    r.ref_id = (int32_t)ld32u(p + 4);
    r.pos = (int32_t)ld32u(p + 8);
    const uint32_t l_read_name = p[12];
    const uint32_t nf = ld32u(p + 16);
    r.n_cigar = nf & 0xffffu;
    r.flag = nf >> 16;
    r.l_seq = (int32_t)ld32u(p + 20);
    r.overrun = 32ull + l_read_name + 4ull * r.n_cigar > bs;
    r.cigar = p + 36 + l_read_name;
    return r;
}

// ⟨upstream⟩ BAM2blocks::cigar2block: M, D, =, X extend the open block; N closes it; I, S, H, P are ignored.
// f(index, start, len) for every block; returns the reference end of the read.
template <class F>
__device__ __forceinline__ uint32_t for_each_block(const Rec& r, uint32_t* n_blocks, F f) {
    uint32_t origin = (uint32_t)r.pos, run = 0, nb = 0;
    for (uint32_t i = 0; i < r.n_cigar; i++) {
        const uint32_t w = ld32u(r.cigar + 4 * i);
        const uint32_t op = w & 0xfu, l = w >> 4;
        if (op == 0 || op == 2 || op == 7 || op == 8) {
            run += l;
        } else if (op == 3) {
            if (run) { f(nb, origin, run); nb++; }
            origin += run + l;
            run = 0;
        }
    }
    if (run) { f(nb, origin, run); nb++; }
    *n_blocks = nb;
    return origin + run;
}

struct Unit {              // what one closing record produces
    uint32_t n_frag, n_blk;
};

struct Out {
    uint32_t f, b;         // fragment index, first block index
};

// One single read, or one pair (x arrived first): the checks and the assembly of host/bam_reader.cpp (on_record,
// pair_or_drop, emit_single, emit_pair).  WRITE = false: counts and reader statistics; WRITE = true: the columns.
template <bool WRITE>
__device__ Unit process_unit(const AssembleArgs& A, bool is_pair, const Rec& x, const Rec& y, Out o, uint32_t* st /*[RS_COUNT] per thread*/) {
    Unit u{0, 0};
    auto emit_header = [&](uint32_t chr, bool dir, uint32_t fs, uint32_t fe, bool two) {
        A.frag_chr[o.f] = chr;
        A.frag_flags[o.f] = (uint8_t)((dir ? 1u : 0u) | (two ? 2u : 0u));
        A.frag_start[o.f] = fs;
        A.frag_end[o.f] = fe;
        A.frag_blk_off[o.f] = o.b;
    };
    auto put_block = [&](uint32_t k, uint32_t s, uint32_t l, uint32_t read) {
        A.blk_start[o.b + k] = s;
        A.blk_len[o.b + k] = l;
        A.blk_read[o.b + k] = (uint8_t)read;
    };
    if (!is_pair) {
        const Rec& r = x;
        if (r.flag & A.skip_flag_mask) { if (!WRITE) st[RS_SKIPPED]++; return u; }
        if (r.ref_id < 0 || r.ref_id >= A.n_ref || r.pos < 0) { if (!WRITE) st[RS_ERROR]++; return u; }
        uint32_t nb;
        const uint32_t end = for_each_block(r, &nb, [&](uint32_t k, uint32_t s, uint32_t l) { if (WRITE) put_block(k, s, l, 0); });
        if (nb == 0) { if (!WRITE) st[RS_ERROR]++; return u; }
        if (WRITE) emit_header((uint32_t)r.ref_id, !(r.flag & 0x10u), (uint32_t)r.pos, end, false);
        else st[RS_SINGLES]++;
        u.n_frag = 1; u.n_blk = nb;
        return u;
    }
    if ((x.flag | y.flag) & A.skip_flag_mask) { if (!WRITE) st[RS_SKIPPED] += 2; return u; }
    if (x.ref_id != y.ref_id || x.ref_id < 0 || x.ref_id >= A.n_ref || x.pos < 0 || y.pos < 0) { if (!WRITE) st[RS_ERROR] += 2; return u; }
    const bool x_left = x.pos <= y.pos;
    const Rec& a = x_left ? x : y;
    const Rec& b = x_left ? y : x;
    const uint32_t pos1 = (uint32_t)a.pos, pos2 = (uint32_t)b.pos;
    // first pass over both cigars: block counts, ends, the read-1 block that holds pos2, the first block of read 2
    uint32_t nb1, nb2, hit = 0xffffffffu, hit_start = 0, first2_len = 0;
    const uint32_t e1 = for_each_block(a, &nb1, [&](uint32_t k, uint32_t s, uint32_t l) {
        if (hit == 0xffffffffu && s <= pos2 && pos2 <= s + l) { hit = k; hit_start = s; }
    });
    const uint32_t e2 = for_each_block(b, &nb2, [&](uint32_t k, uint32_t s, uint32_t l) { if (k == 0) first2_len = l; });
    if (nb1 == 0 || nb2 == 0) { if (!WRITE) st[RS_ERROR] += 2; return u; }
    const bool dir = (a.flag & 0x40u) ? !(a.flag & 0x10u) : !(a.flag & 0x20u);
    const uint32_t chr = (uint32_t)a.ref_id;
    const bool is_long = A.pair_long_inclusive ? (e1 <= pos2) : (e1 < pos2);
    if (!WRITE) st[RS_PAIRS]++;
    u.n_frag = 1;
    bool gap_pair = false;
    if (!is_long) {
        const bool is_short = A.pair_short_inclusive ? (e1 >= e2) : (e1 > e2);
        if (is_short) {
            u.n_blk = nb1;
            if (WRITE) {
                uint32_t t;
                for_each_block(a, &t, [&](uint32_t k, uint32_t s, uint32_t l) { put_block(k, s, l, 0); });
                emit_header(chr, dir, pos1, e1, false);
            } else st[RS_SHORT]++;
            return u;
        }
        if (hit != 0xffffffffu) {
            u.n_blk = hit + 1 + (nb2 - 1);
            if (WRITE) {
                uint32_t t;
                for_each_block(a, &t, [&](uint32_t k, uint32_t s, uint32_t l) {
                    if (k < hit) put_block(k, s, l, 0);
                    else if (k == hit) put_block(k, s, pos2 - hit_start + first2_len, 0);
                });
                for_each_block(b, &t, [&](uint32_t k, uint32_t s, uint32_t l) { if (k) put_block(hit + k, s, l, 0); });
                emit_header(chr, dir, pos1, max(e1, e2), false);
            } else st[RS_INTERSECT]++;
            return u;
        }
        // pos2 in an N gap of read 1 (OQ-3): two separate reads; its class in the summary is a switch
        gap_pair = true;
    }
    u.n_blk = nb1 + nb2;
    if (WRITE) {
        uint32_t t;
        for_each_block(a, &t, [&](uint32_t k, uint32_t s, uint32_t l) { put_block(k, s, l, 0); });
        for_each_block(b, &t, [&](uint32_t k, uint32_t s, uint32_t l) { put_block(nb1 + k, s, l, 1); });
        emit_header(chr, dir, pos1, max(e1, e2), true);
    } else st[gap_pair && A.gap_pair_is_intersect ? RS_INTERSECT : RS_LONG]++;
    return u;
}

// (index of the last unpaired record before i) + 1 for the thread's first record; 0 = none
__device__ __forceinline__ uint32_t last_unpaired_before(const AssembleArgs& A, uint32_t i0, uint32_t* s_w) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t m = 0;
    for (uint32_t k = 0; k < kPerThread; k++) {
        const uint32_t i = i0 + k;
        if (i < A.n_rec && !((A.R.rec_nf[i] >> 16) & 1u)) m = i + 1;
    }
    const uint32_t inc = wave_inclusive_max(m, lane);
    if (lane == 63) s_w[wave] = inc;
    wg_barrier();
    uint32_t before = A.tile_carry[blockIdx.x];
    for (uint32_t w = 0; w < wave; w++) before = max(before, s_w[w]);
    uint32_t excl = __shfl_up(inc, 1);
    if (lane == 0) excl = 0;
    return max(before, excl);
}

template <bool WRITE>
__global__ __launch_bounds__(kThreads) void assemble_kernel(AssembleArgs A) {
    __shared__ uint32_t s_w[kThreads / 64];
    __shared__ unsigned long long s_stats[RS_COUNT];
    if (!WRITE && threadIdx.x < RS_COUNT) s_stats[threadIdx.x] = 0;
    const uint32_t i0 = blockIdx.x * kPairTile + threadIdx.x * kPerThread;
    uint32_t last = 0;
    if (A.mate) wg_barrier();
    else last = last_unpaired_before(A, i0, s_w);         // (barrier inside: s_stats is zeroed for everyone)
    uint32_t st[RS_COUNT];
#pragma unroll
    for (int k = 0; k < RS_COUNT; k++) st[k] = 0;
    for (uint32_t k = 0; k < kPerThread; k++) {
        const uint32_t i = i0 + k;
        if (i >= A.n_rec) break;
        if (A.mate) {
            // name-joined pairing: the join kernels decided who completes what (records and nucleotides were counted
            // when the records arrived)
            const uint32_t m = A.mate[i];
            Unit u{0, 0};
            if (m < A.n_rec || m == JOIN_SINGLE) {
                Out o{0, 0};
                if (WRITE) { o.f = A.fr_cnt[i]; o.b = A.bl_cnt[i]; }
                const Rec y = load_rec(A.buf, A.R.rec_off[i]);
                if (m < A.n_rec) {
                    const Rec x = load_rec(A.buf, A.R.rec_off[m]);
                    if (!x.overrun && !y.overrun) u = process_unit<WRITE>(A, true, x, y, o, st);
                } else if (!y.overrun) {
                    u = process_unit<WRITE>(A, false, y, y, o, st);
                }
            }
            if (!WRITE) { A.fr_cnt[i] = u.n_frag; A.bl_cnt[i] = u.n_blk; }
            continue;
        }
        const uint32_t nf = A.R.rec_nf[i];
        const bool paired = (nf >> 16) & 1u;
        const bool waiting = ((i - last) & 1u) != 0;      // an odd run of paired records precedes: record i is the mate
        if (!paired) last = i + 1;
        Unit u{0, 0};
        if (!waiting && paired) {
            // opens a pair: the next record is its mate.  The last record of the buffer stays for the next feed.
            if (i + 1 == A.n_rec) {
                if (!WRITE) *A.dangling = 1u;
            }
        } else {
            Out o{0, 0};
            if (WRITE) { o.f = A.fr_cnt[i]; o.b = A.bl_cnt[i]; }
            const Rec y = load_rec(A.buf, A.R.rec_off[i]);
            if (waiting) {
                const Rec x = load_rec(A.buf, A.R.rec_off[i - 1]);
                if (!WRITE) {
                    st[RS_RECORDS] += 2;
                    if (x.l_seq > 0) st[RS_NUCLEOTIDES] += (uint32_t)x.l_seq;
                    if (y.l_seq > 0) st[RS_NUCLEOTIDES] += (uint32_t)y.l_seq;
                    if (x.overrun) atomicMin(A.R.err, (unsigned long long)(A.R.rec_off[i - 1] << 2 | 2ull));
                    if (y.overrun) atomicMin(A.R.err, (unsigned long long)(A.R.rec_off[i] << 2 | 2ull));
                }
                if (!x.overrun && !y.overrun) u = process_unit<WRITE>(A, true, x, y, o, st);
            } else {
                if (!WRITE) {
                    st[RS_RECORDS] += 1;
                    if (y.l_seq > 0) st[RS_NUCLEOTIDES] += (uint32_t)y.l_seq;
                    if (y.overrun) atomicMin(A.R.err, (unsigned long long)(A.R.rec_off[i] << 2 | 2ull));
                }
                if (!y.overrun) u = process_unit<WRITE>(A, false, y, y, o, st);
            }
        }
        if (!WRITE) {
            A.fr_cnt[i] = u.n_frag;
            A.bl_cnt[i] = u.n_blk;
        }
    }
    if (!WRITE) {
#pragma unroll
        for (int k = 0; k < RS_COUNT; k++) {
            unsigned long long v = st[k];
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
            if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_stats[k], v);
        }
        wg_barrier();
        if (threadIdx.x < RS_COUNT && s_stats[threadIdx.x]) atomicAdd(&A.stats[threadIdx.x], s_stats[threadIdx.x]);
    }
}

// ---- name-joined pairing -----------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long name_hash(const uint8_t* name, uint32_t n) {
    unsigned long long h = 0x9E3779B97F4A7C15ull ^ n;
    for (uint32_t k = 0; k < n; k++) {
        h = (h ^ name[k]) * 0x100000001B3ull;
        h ^= h >> 29;
    }
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h == kJoinNoKey ? h - 1ull : h;
}

__global__ __launch_bounds__(kThreads) void join_keys_kernel(JoinArgs J) {
    __shared__ unsigned long long s_stats[2];
    if (threadIdx.x < 2) s_stats[threadIdx.x] = 0;
    wg_barrier();
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    unsigned long long nrec = 0, nnuc = 0;
    if (i < J.n_rec) {
        const uint64_t off = J.R.rec_off[i];
        const Rec r = load_rec(J.buf, off);
        if (r.overrun) atomicMin(J.R.err, (unsigned long long)(off << 2 | 2ull));
        if (i >= J.n_old) { nrec = 1; if (r.l_seq > 0) nnuc = (unsigned long long)r.l_seq; }
        const uint32_t l_name = J.buf[off + 12];
        const bool paired = (r.flag & 1u) != 0 && !r.overrun;
        J.keys[i] = paired ? name_hash(J.buf + off + 36, l_name ? l_name - 1u : 0u) : kJoinNoKey;
        J.vals[i] = i;
        J.mate[i] = JOIN_SINGLE;       // paired records are overwritten by their group's walk
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { nrec += __shfl_xor(nrec, d); nnuc += __shfl_xor(nnuc, d); }
    if ((threadIdx.x & 63) == 0) { if (nrec) atomicAdd(&s_stats[0], nrec); if (nnuc) atomicAdd(&s_stats[1], nnuc); }
    wg_barrier();
    if (threadIdx.x == 0) {
        if (s_stats[0]) atomicAdd(&J.stats[RS_RECORDS], s_stats[0]);
        if (s_stats[1]) atomicAdd(&J.stats[RS_NUCLEOTIDES], s_stats[1]);
    }
}

__device__ __forceinline__ bool same_name(const uint8_t* buf, uint64_t a, uint64_t b) {
    const uint32_t la = buf[a + 12], lb = buf[b + 12];
    if (la != lb) return false;
    for (uint32_t k = 0; k + 1 < la; k++)
        if (buf[a + 36 + k] != buf[b + 36 + k]) return false;
    return true;
}

// one thread per sorted position; the head of a group of equal hashes walks the group (arrival order: the sort is stable)
__global__ __launch_bounds__(kThreads) void join_groups_kernel(JoinArgs J) {
    const uint32_t p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= J.n_rec) return;
    const unsigned long long key = J.keys[p];
    if (key == kJoinNoKey || (p > 0 && J.keys[p - 1] == key)) return;
    uint32_t waiting = 0xffffffffu, displaced = 0;
    for (uint32_t q = p; q < J.n_rec && J.keys[q] == key; q++) {
        const uint32_t r = J.vals[q];
        if (waiting == 0xffffffffu) { waiting = r; continue; }
        if (!same_name(J.buf, J.R.rec_off[waiting], J.R.rec_off[r])) { *J.collision = 1u; continue; }
        const uint32_t fw = J.R.rec_nf[waiting] >> 16, fr = J.R.rec_nf[r] >> 16;
        if (((fw ^ fr) & 0xc0u) == 0u) {          // the same mate again: the waiting record is displaced
            J.mate[waiting] = JOIN_DEAD;
            displaced++;
            waiting = r;
        } else {
            J.mate[waiting] = JOIN_FIRST;
            J.mate[r] = waiting;
            waiting = 0xffffffffu;
        }
    }
    if (waiting != 0xffffffffu) J.mate[waiting] = JOIN_WAITING;
    if (displaced) atomicAdd(&J.stats[RS_ERROR], (unsigned long long)displaced);
}

__global__ __launch_bounds__(kThreads) void join_sizes_kernel(JoinArgs J) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    uint32_t sz = 0;
    if (i < J.n_rec && J.mate[i] == JOIN_WAITING) sz = 4u + ld32u(J.buf + J.R.rec_off[i]);
    if (i <= J.n_rec) J.sizes[i] = sz;
    const uint64_t w = __ballot(sz != 0u);
    if (w && (threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)w) - 1)) atomicAdd(J.n_waiting, (uint32_t)__popcll(w));
}

// one wave per record: the waiting records, whole, in arrival order
__global__ __launch_bounds__(kThreads) void join_carry_kernel(JoinArgs J, const uint32_t* __restrict__ offsets, uint8_t* __restrict__ dst) {
    const uint32_t i = blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (i >= J.n_rec || J.mate[i] != JOIN_WAITING) return;
    const uint64_t src = J.R.rec_off[i];
    const uint32_t n = 4u + ld32u(J.buf + src);
    uint8_t* d = dst + offsets[i];
    for (uint32_t k = lane; k < n; k += 64u) d[k] = J.buf[src + k];
}

__global__ void close_offsets_kernel(uint32_t* frag_blk_off, uint32_t n_frag, uint32_t n_blk) { frag_blk_off[n_frag] = n_blk; }

}   // namespace

void launch_chain_guess(const ChainBufs& C, hipStream_t s) {
    if (!C.n_seg) return;
    hipLaunchKernelGGL(chain_guess_kernel, dim3((C.n_seg + kThreads - 1) / kThreads), dim3(kThreads), 0, s, C);
}

void launch_chain_verify(const ChainBufs& C, hipStream_t s) {
    if (!C.n_seg) return;
    hipLaunchKernelGGL(chain_verify_kernel, dim3((C.n_seg + kThreads - 1) / kThreads), dim3(kThreads), 0, s, C);
}

void launch_chain_emit(const ChainBufs& C, const uint32_t* seg_base, const RecordIndex& R, hipStream_t s) {
    if (!C.n_seg) return;
    hipLaunchKernelGGL(chain_emit_kernel, dim3((C.n_seg + kThreads - 1) / kThreads), dim3(kThreads), 0, s, C, seg_base, R);
}

void launch_join_keys(const JoinArgs& J, hipStream_t s) {
    if (J.n_rec) hipLaunchKernelGGL(join_keys_kernel, dim3((J.n_rec + kThreads - 1) / kThreads), dim3(kThreads), 0, s, J);
}
void launch_join_groups(const JoinArgs& J, hipStream_t s) {
    if (J.n_rec) hipLaunchKernelGGL(join_groups_kernel, dim3((J.n_rec + kThreads - 1) / kThreads), dim3(kThreads), 0, s, J);
}
void launch_join_sizes(const JoinArgs& J, hipStream_t s) {
    hipLaunchKernelGGL(join_sizes_kernel, dim3((J.n_rec + 1 + kThreads - 1) / kThreads), dim3(kThreads), 0, s, J);
}
void launch_join_carry(const JoinArgs& J, const uint32_t* offsets, uint8_t* dst, hipStream_t s) {
    if (J.n_rec) hipLaunchKernelGGL(join_carry_kernel, dim3((J.n_rec + kThreads / 64 - 1) / (kThreads / 64)), dim3(kThreads), 0, s, J, offsets, dst);
}

uint32_t pair_tiles(uint32_t n_rec) { return (n_rec + kPairTile - 1) / kPairTile; }

void launch_pair_state(const uint32_t* rec_nf, uint32_t n_rec, uint32_t* tile_last, uint32_t* tile_carry, hipStream_t s) {
    if (!n_rec) return;
    const uint32_t nt = pair_tiles(n_rec);
    hipLaunchKernelGGL(pair_tile_last_kernel, dim3(nt), dim3(kThreads), 0, s, rec_nf, n_rec, tile_last);
    hipLaunchKernelGGL(pair_tile_scan_kernel, dim3(1), dim3(kThreads), 0, s, (const uint32_t*)tile_last, nt, tile_carry);
}

void launch_assemble_count(const AssembleArgs& A, hipStream_t s) {
    if (!A.n_rec) return;
    hipLaunchKernelGGL(assemble_kernel<false>, dim3(pair_tiles(A.n_rec)), dim3(kThreads), 0, s, A);
}

void launch_assemble_write(const AssembleArgs& A, uint32_t n_frag, uint32_t n_blk, hipStream_t s) {
    if (A.n_rec) hipLaunchKernelGGL(assemble_kernel<true>, dim3(pair_tiles(A.n_rec)), dim3(kThreads), 0, s, A);
    hipLaunchKernelGGL(close_offsets_kernel, dim3(1), dim3(1), 0, s, A.frag_blk_off, n_frag, n_blk);
}

}   // namespace dev
}   // namespace irf
