// bgzf.hip — BGZF members -> inflated BAM stream, on the device (gfx950).
//
// Stands where ⟨upstream⟩ bin/IRFinder runs `gzip -cd` in front of the engine (SURVEY §3.1, §8 f N1).  BGZF (SAM
// spec §4.1) is a concatenation of independent gzip members of at most 64 KiB of inflated data each, so the file is
// block-parallel: one LANE inflates one member (RFC 1951: stored, fixed and dynamic Huffman blocks, written here from
// the RFC — zlib stays on the host, for the host reader only).
//
// A lane's Huffman state lives in LDS: canonical decoding needs, per code length L, the left-justified upper limit of
// the codes of that length and the offset of their symbols in the length-sorted symbol list — 15 + 15 + 288 entries for
// literal/length, 15 + 15 + 32 for distances = 380 uint16 per lane, interleaved by lane (element j of lane l at
// [j][l]), 47.5 KiB per 64-lane workgroup, three workgroups per CU.  A code is decoded by a 4-step binary search over the
// limits (15 peeked bits, bit-reversed: DEFLATE packs Huffman codes most-significant bit first into a
// least-significant-bit-first stream).  The code-length alphabet of a dynamic block (19 symbols, <= 7 bits) is decoded
// from registers; the code lengths are read TWICE (count per length, then place the symbols), which saves a 320-entry
// per-lane length array.
//
// Output: literal bytes collect in a register and leave as dwords; a match of up to 32 bytes is a deferred copy (loads
// now, stores behind the next symbol's decode; one that overlaps its own output at a distance below 8 is stored as the
// period the first 8 loaded bytes hold), anything else is copied in place.  The CRC-32 of every member is checked by a second kernel (one wave per member, 64 slices
// combined with x^(8n) mod P), so a corrupt file is refused exactly where the host reader refuses it.
#include "dreader_kernels.h"

namespace irf {
namespace dev {

namespace {

#ifndef IRF_LITERAL_RUN
#define IRF_LITERAL_RUN 1          // a literal behind a symbol is taken in the same iteration (0 or 1: it refills from the OLD input window)
#endif

constexpr int LIT_SYM = 0, DIST_SYM = 288, LIT_LIM = 320, LIT_BASE = 335, DIST_LIM = 350, DIST_BASE = 365, TAB_U16 = 380;
constexpr int kLanes = 64;

__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ __forceinline__ void st32u(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

struct Bits {
    const uint8_t* p;    // next input byte to load
    uint64_t bb;         // bit buffer, next bit = bit 0
    uint32_t bc;         // valid bits in bb
    uint32_t nxt;        // the 4 bytes before p, loaded one refill ahead: their latency is hidden behind the decoding of
                         // the bits in bb (with 64 streams per wave some lane refills in every iteration)
    __device__ __forceinline__ void start(const uint8_t* in0) {
        bb = 0; bc = 0;
        nxt = ld32u(in0);
        p = in0 + 4;
    }
    __device__ __forceinline__ void refill() {           // afterwards bc >= 33
        if (bc <= 32) {
            bb |= (uint64_t)nxt << bc;
            bc += 32;
            nxt = ld32u(p);
            p += 4;
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)bb & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(uint32_t n) { bb >>= n; bc -= n; }
    __device__ __forceinline__ uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
};

// The bit buffer of the symbol loop: the next four input words (q0 .. q3 = the 16 bytes at p) are in registers, so a
// refill is register work.  An iteration refills at most three times (15 + 5 bits, refill, 15 + 13 bits, refill, 15 bits
// of the literal behind the symbol, refill) and every lane reloads all four words with ONE 16-byte load per iteration (its
// own address: 64 cache lines per instruction, once per iteration).
struct FastBits {
    const uint8_t* p;    // address of q0
    uint64_t bb;
    uint32_t bc, q0, q1, q2, q3;
    __device__ __forceinline__ void from(const Bits& s) {
        p = s.p - 4; bb = s.bb; bc = s.bc; q0 = s.nxt;
        q1 = ld32u(s.p); q2 = ld32u(s.p + 4); q3 = ld32u(s.p + 8);
        asm volatile("" ::"v"(q0), "v"(q1), "v"(q2), "v"(q3));      // waited for here: nothing is pending when the symbol loop is entered
    }
    __device__ __forceinline__ void to(Bits& s) const {
        s.p = p + 4; s.bb = bb; s.bc = bc; s.nxt = q0;
    }
    __device__ __forceinline__ void refill() {
        if (bc <= 32) {
            bb |= (uint64_t)q0 << bc;
            bc += 32;
            q0 = q1; q1 = q2; q2 = q3;
            p += 4;
        }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) const { return (uint32_t)bb & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(uint32_t n) { bb >>= n; bc -= n; }
    __device__ __forceinline__ uint32_t take(uint32_t n) { const uint32_t v = peek(n); drop(n); return v; }
};

#define TAB(j) s_tab[(j) * kLanes + lane]

// The 15 limits and the 15 bases of a code, in registers: the search over the limits is the dependent chain of every
// symbol, and with one wave per SIMD (the tables of 64 lanes fill a third of the LDS) nothing hides an LDS round trip.
struct CodeRegs {
    uint32_t lim[15], base[15];
};

__device__ __forceinline__ void load_code_regs(const uint16_t* s_tab, uint32_t lane, int LIM, int BASE, CodeRegs& c) {
    // (written out: the indices must be constants before the first scalar-replacement pass, or the selects below turn
    // into indexed loads from a stack copy)
#define IRF_LC(k) c.lim[k] = TAB(LIM + k); c.base[k] = TAB(BASE + k);
    IRF_LC(0) IRF_LC(1) IRF_LC(2) IRF_LC(3) IRF_LC(4) IRF_LC(5) IRF_LC(6) IRF_LC(7)
    IRF_LC(8) IRF_LC(9) IRF_LC(10) IRF_LC(11) IRF_LC(12) IRF_LC(13) IRF_LC(14)
#undef IRF_LC
}

// canonical Huffman: smallest length whose left-justified limit exceeds the next 15 bits (binary search by selects)
template <class B>
__device__ __forceinline__ int decode_sym(const uint16_t* s_tab, uint32_t lane, const CodeRegs& c, int SYM, B& b) {
    const uint32_t v = __brev(b.peek(15)) >> 17;
    // (values first: `x ? c.lim[i] : c.lim[j]` would select an address and load through it)
    const uint32_t l0 = c.lim[0], l1 = c.lim[1], l2 = c.lim[2], l3 = c.lim[3], l4 = c.lim[4], l5 = c.lim[5], l6 = c.lim[6], l7 = c.lim[7],
                   l8 = c.lim[8], l9 = c.lim[9], l10 = c.lim[10], l11 = c.lim[11], l12 = c.lim[12], l13 = c.lim[13], l14 = c.lim[14];
    const uint32_t a0 = c.base[0], a1 = c.base[1], a2 = c.base[2], a3 = c.base[3], a4 = c.base[4], a5 = c.base[5], a6 = c.base[6], a7 = c.base[7],
                   a8 = c.base[8], a9 = c.base[9], a10 = c.base[10], a11 = c.base[11], a12 = c.base[12], a13 = c.base[13], a14 = c.base[14];
    const bool b8 = l7 <= v;
    const bool b4 = (b8 ? l11 : l3) <= v;
    const bool b2 = (b8 ? (b4 ? l13 : l9) : (b4 ? l5 : l1)) <= v;
    const uint32_t e0 = b2 ? l2 : l0, e1 = b2 ? l6 : l4, e2 = b2 ? l10 : l8, e3 = b2 ? l14 : l12;
    const bool b1 = (b8 ? (b4 ? e3 : e2) : (b4 ? e1 : e0)) <= v;
    const uint32_t pos = (b8 ? 8u : 0u) + (b4 ? 4u : 0u) + (b2 ? 2u : 0u) + (b1 ? 1u : 0u);
    if (pos >= 15) return -1;
    // base[pos] by the same bits
    const uint32_t f0 = b1 ? a1 : a0, f1 = b1 ? a3 : a2, f2 = b1 ? a5 : a4, f3 = b1 ? a7 : a6;
    const uint32_t f4 = b1 ? a9 : a8, f5 = b1 ? a11 : a10, f6 = b1 ? a13 : a12, f7 = a14;
    const uint32_t g0 = b2 ? f1 : f0, g1 = b2 ? f3 : f2, g2 = b2 ? f5 : f4, g3 = b2 ? f7 : f6;
    const uint32_t base = b8 ? (b4 ? g3 : g2) : (b4 ? g1 : g0);
    const uint32_t idx = (uint16_t)(base + (v >> (14 - pos)));
    b.drop(pos + 1);
    return TAB(SYM + idx);
}

// counts per length (in the LIM slots) -> limits; BASE slots = first free symbol slot of each length (for placing).
// Returns 0, or -1 for an over-subscribed set or an incomplete one (zlib accepts an incomplete set only when its longest
// code has one bit).
__device__ int limits_from_counts(uint16_t* s_tab, uint32_t lane, int LIM, int BASE) {
    uint32_t first = 0, offs = 0, maxlen = 0;
    for (uint32_t L = 1; L <= 15; L++) {
        const uint32_t c = TAB(LIM + L - 1);
        if (first + c > (1u << L)) return -1;
        if (c) maxlen = L;
        TAB(BASE + L - 1) = (uint16_t)offs;
        TAB(LIM + L - 1) = (uint16_t)((first + c) << (15 - L));
        offs += c;
        first = (first + c) << 1;
    }
    // first == 2 * (codes used at length 15); complete iff that is 2^16
    if (first != (1u << 16) && maxlen > 1) return -1;
    return 0;
}

// BASE slots -> (offset of the length's symbols) - (first code of the length), the constant decode_sym adds to a code
__device__ void bases_from_limits(uint16_t* s_tab, uint32_t lane, int LIM, int BASE) {
    uint32_t first = 0, offs = 0;
    for (uint32_t L = 1; L <= 15; L++) {
        const uint32_t end = (uint32_t)TAB(LIM + L - 1) >> (15 - L);
        TAB(BASE + L - 1) = (uint16_t)(offs - first);
        offs += end - first;
        first = end << 1;
    }
}

__device__ __forceinline__ void zero_counts(uint16_t* s_tab, uint32_t lane, int LIM) {
    for (int L = 0; L < 15; L++) TAB(LIM + L) = 0;
}

__device__ __forceinline__ void place_sym(uint16_t* s_tab, uint32_t lane, int BASE, int SYM, uint32_t len, uint32_t sym) {
    const uint32_t k = TAB(BASE + len - 1);
    TAB(BASE + len - 1) = (uint16_t)(k + 1);
    TAB(SYM + k) = (uint16_t)sym;
}

__device__ __forceinline__ uint32_t fixed_lit_len(uint32_t s) { return s < 144 ? 8u : s < 256 ? 9u : s < 280 ? 7u : 8u; }

__device__ void build_fixed(uint16_t* s_tab, uint32_t lane) {
    zero_counts(s_tab, lane, LIT_LIM);
    zero_counts(s_tab, lane, DIST_LIM);
    TAB(LIT_LIM + 6) = 24; TAB(LIT_LIM + 7) = 152; TAB(LIT_LIM + 8) = 112;
    TAB(DIST_LIM + 4) = 32;
    limits_from_counts(s_tab, lane, LIT_LIM, LIT_BASE);
    limits_from_counts(s_tab, lane, DIST_LIM, DIST_BASE);
    for (uint32_t s = 0; s < 288; s++) place_sym(s_tab, lane, LIT_BASE, LIT_SYM, fixed_lit_len(s), s);
    for (uint32_t s = 0; s < 32; s++) place_sym(s_tab, lane, DIST_BASE, DIST_SYM, 5, s);
    bases_from_limits(s_tab, lane, LIT_LIM, LIT_BASE);
    bases_from_limits(s_tab, lane, DIST_LIM, DIST_BASE);
}

// the code-length alphabet, in registers: counts per length (5 bits each), symbols sorted by (length, symbol) (5 bits each)
struct ClCode { uint64_t counts, sym_lo, sym_hi; };

__device__ __forceinline__ int cl_decode(const ClCode& c, Bits& b) {
    const uint32_t v = __brev(b.peek(7)) >> 25;
    uint32_t first = 0, index = 0;
#pragma unroll
    for (uint32_t L = 1; L <= 7; L++) {
        const uint32_t n = (uint32_t)(c.counts >> (5 * (L - 1))) & 31u;
        const uint32_t code = v >> (7 - L);
        if (code - first < n) {
            const uint32_t k = index + code - first;
            b.drop(L);
            return (int)(k < 12 ? (c.sym_lo >> (5 * k)) & 31u : (c.sym_hi >> (5 * (k - 12))) & 31u);
        }
        index += n;
        first = (first + n) << 1;
    }
    return -1;
}

// reads the hlit + hdist code lengths of a dynamic block: PLACE = false counts them per length (LIM slots),
// PLACE = true puts every symbol into its slot of the length-sorted lists.  Returns 0 / -1.
template <bool PLACE>
__device__ int read_code_lengths(uint16_t* s_tab, uint32_t lane, Bits& b, const ClCode& cl, uint32_t hlit, uint32_t hdist,
                                 bool* has_eob) {
    const uint32_t total = hlit + hdist;
    uint32_t s = 0, prev = 0;
    while (s < total) {
        b.refill();
        const int sym = cl_decode(cl, b);
        if (sym < 0) return -1;
        uint32_t rep = 1, val = (uint32_t)sym;
        if (sym == 16) {
            if (s == 0) return -1;
            val = prev;
            rep = 3 + b.take(2);
        } else if (sym == 17) {
            val = 0;
            rep = 3 + b.take(3);
        } else if (sym == 18) {
            val = 0;
            rep = 11 + b.take(7);
        }
        if (s + rep > total) return -1;
        if (val) {
            for (uint32_t r = 0; r < rep; r++) {
                const uint32_t t = s + r;
                const bool lit = t < hlit;
                if (PLACE) {
                    if (lit) place_sym(s_tab, lane, LIT_BASE, LIT_SYM, val, t);
                    else place_sym(s_tab, lane, DIST_BASE, DIST_SYM, val, t - hlit);
                } else {
                    const int slot = (lit ? LIT_LIM : DIST_LIM) + (int)val - 1;
                    TAB(slot) = (uint16_t)(TAB(slot) + 1);
                    if (t == 256) *has_eob = true;
                }
            }
        }
        s += rep;
        prev = val;
    }
    return 0;
}

// order in which the code-length code lengths are stored (RFC 1951 §3.2.7), 5 bits each
constexpr uint64_t kClOrderLo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 |
                                10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
constexpr uint64_t kClOrderHi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;

__device__ int build_dynamic(uint16_t* s_tab, uint32_t lane, Bits& b) {
    b.refill();
    const uint32_t hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return -1;
    uint64_t cll = 0;     // 3 bits per code-length symbol
    for (uint32_t i = 0; i < hclen; i++) {
        b.refill();
        const uint32_t sym = (uint32_t)(i < 12 ? kClOrderLo >> (5 * i) : kClOrderHi >> (5 * (i - 12))) & 31u;
        cll |= (uint64_t)b.take(3) << (3 * sym);
    }
    ClCode cl{0, 0, 0};
    {
        uint32_t n = 0;
        int left = 1;
        for (uint32_t L = 1; L <= 7; L++) {
            uint32_t c = 0;
            for (uint32_t sym = 0; sym < 19; sym++) {
                if (((cll >> (3 * sym)) & 7u) == L) {
                    if (n < 12) cl.sym_lo |= (uint64_t)sym << (5 * n);
                    else cl.sym_hi |= (uint64_t)sym << (5 * (n - 12));
                    n++;
                    c++;
                }
            }
            cl.counts |= (uint64_t)c << (5 * (L - 1));
            left = (left << 1) - (int)c;
            if (left < 0) return -1;
        }
        if (left > 0) return -1;          // zlib: an incomplete code-length code is an error
    }
    zero_counts(s_tab, lane, LIT_LIM);
    zero_counts(s_tab, lane, DIST_LIM);
    bool has_eob = false;
    const Bits at_lengths = b;
    if (read_code_lengths<false>(s_tab, lane, b, cl, hlit, hdist, &has_eob) != 0) return -1;
    if (!has_eob) return -1;
    if (limits_from_counts(s_tab, lane, LIT_LIM, LIT_BASE) != 0) return -1;
    if (limits_from_counts(s_tab, lane, DIST_LIM, DIST_BASE) != 0) return -1;
    b = at_lengths;
    read_code_lengths<true>(s_tab, lane, b, cl, hlit, hdist, &has_eob);
    bases_from_limits(s_tab, lane, LIT_LIM, LIT_BASE);
    bases_from_limits(s_tab, lane, DIST_LIM, DIST_BASE);
    return 0;
}

__global__ __launch_bounds__(kLanes) void bgzf_inflate_kernel(const uint8_t* __restrict__ in, const BgzfMember* __restrict__ members,
                                                              uint32_t n_members, uint8_t* out, uint32_t* __restrict__ status) {
    __shared__ uint16_t s_tab[TAB_U16 * kLanes];
    const uint32_t lane = threadIdx.x;
    const uint32_t mi = blockIdx.x * kLanes + lane;
    if (mi >= n_members) return;
#ifndef IRF_INFLATE_EXP
#define IRF_INFLATE_EXP 0                  // diagnostic builds only (tools/inflate_exp.sh): 1 no match copies, 2 no literal stores, 4 cached input, 8 no slow-path copies, 16 no deferred copies
#endif
    BgzfMember m = members[mi];
    if (IRF_INFLATE_EXP & 4) {
        const BgzfMember a = members[mi % 128u];
        if (a.isize == m.isize) { m.in_off = a.in_off; m.clen = a.clen; }
    }
    if (m.isize == 0) {                    // the 28-byte EOF marker and other empty members (the host reader skips them too)
        status[mi] = BGZF_OK;
        return;
    }
    const uint8_t* const in0 = in + m.in_off;
    const uint8_t* const in_lim = in0 + m.clen + 8;      // a valid stream never has its next input word beyond this (16 more bytes are read)
    Bits b;
    b.start(in0);
    uint8_t* const obase = out + m.out_off;
    const uint32_t isize = m.isize;
    uint32_t op = 0, pend = 0, pn = 0;
    bool bad = false, last = false;

    auto consumed_bits = [&]() -> uint64_t { return (uint64_t)(b.p - in0 - 4) * 8ull - b.bc; };
    auto put = [&](uint32_t byte) {
        uint8_t* a = obase + op;
        if (IRF_INFLATE_EXP & 2) { op++; return; }
        pend |= byte << (8 * pn);
        if (++pn == 4) {
            st32u(a - 3, pend);                 // (any alignment)
            pend = 0;
            pn = 0;
        }
        op++;
    };
    // in front of a match of >= 3 bytes that fits the member: the pending literals leave as ONE dword; the zero bytes
    // behind them lie inside the match's destination, which is written afterwards
    auto flush_before_match = [&]() {
        if (pn) st32u(obase + op - pn, pend);
        pend = 0;
        pn = 0;
    };
    auto flush = [&]() {
        uint8_t* a = obase + op - pn;
        for (uint32_t k = 0; k < pn; k++) a[k] = (uint8_t)(pend >> (8 * k));
        pend = 0;
        pn = 0;
    };

    while (!bad && !last) {
        if (consumed_bits() > (uint64_t)m.clen * 8ull) { bad = true; break; }
        b.refill();
        last = b.take(1) != 0;
        const uint32_t btype = b.take(2);
        if (btype == 0) {
            b.drop(b.bc & 7u);
            b.refill();
            const uint32_t len = b.take(16), nlen = b.take(16);
            if ((len ^ nlen) != 0xffffu || op + len > isize || consumed_bits() + 8ull * len > (uint64_t)m.clen * 8ull) { bad = true; break; }
            for (uint32_t k = 0; k < len; k++) {
                b.refill();
                put(b.take(8));
            }
            continue;
        }
        if (btype == 3) { bad = true; break; }
        if (btype == 1) build_fixed(s_tab, lane);
        else if (build_dynamic(s_tab, lane, b) != 0) { bad = true; break; }
        CodeRegs lit, dcode;
        load_code_regs(s_tab, lane, LIT_LIM, LIT_BASE, lit);
        load_code_regs(s_tab, lane, DIST_LIM, DIST_BASE, dcode);
        // One literal/length decode site for every lane, at the bottom of the loop, one step ahead of its use (sym).
        // After a short match the loads of the copy are issued first, then the next symbol is decoded (bit buffer + LDS
        // only), and only then are the loaded bytes stored: the copy's round trip to L2 runs under the decoding.  The
        // eight loads are issued by every lane in every iteration (lanes without a deferred copy read the start of
        // their member), together with the load of the next four input words: an iteration waits for memory ONCE — behind
        // the decode AND the literal that may follow the symbol, ~150 instructions after the loads — for everything it and
        // its predecessor issued (on gfx9 stores count in vmcnt like loads, so any earlier wait for a load would also be
        // a wait for the acknowledgement of the stores before it).
        FastBits fb;
        fb.from(b);
        fb.refill();
        int sym = decode_sym(s_tab, lane, lit, LIT_SYM, fb);
        while (true) {
            if (fb.p > in_lim) { bad = true; break; }
            if (sym < 0) { bad = true; break; }
            const uint8_t* dsrc = out;                      // (lanes without a deferred copy all read the same line: one tag lookup)
            uint8_t* ddst = obase;
            uint32_t need = 0;                              // bytes of a deferred copy (whole 8-byte steps)
            uint32_t pat_d = 0;                             // its period, if it overlaps its own output (dist < 8)
            if (sym < 256) {
                if (op >= isize) { bad = true; break; }
                put((uint32_t)sym);
            } else {
                if (sym == 256) break;
                sym -= 257;
                if (sym >= 29) { bad = true; break; }
                uint32_t len;
                if (sym < 8) len = 3u + (uint32_t)sym;
                else if (sym == 28) len = 258u;
                else {
                    const uint32_t e = ((uint32_t)sym - 4u) >> 2;
                    len = 3u + ((4u + ((uint32_t)sym & 3u)) << e) + fb.take(e);
                }
                fb.refill();
                const int ds = decode_sym(s_tab, lane, dcode, DIST_SYM, fb);
                if (ds < 0 || ds >= 30) { bad = true; break; }
                uint32_t dist;
                if (ds < 4) dist = 1u + (uint32_t)ds;
                else {
                    const uint32_t e = ((uint32_t)ds - 2u) >> 1;
                    dist = 1u + ((2u + ((uint32_t)ds & 1u)) << e) + fb.take(e);
                }
                if (dist > op || op + len > isize) { bad = true; break; }
                flush_before_match();
                uint8_t* dst = obase + op;
                const uint8_t* src = dst - dist;
                // Whole 8-byte steps may write up to 8 bytes past the end of the match as long as they stay inside the
                // member: the output that follows overwrites them (a member that ends early is refused: op != isize).
                const bool may_overrun = op + len + 8u <= isize;
                op += len;
                const uint32_t whole = (len + 7u) & ~7u;
                if (IRF_INFLATE_EXP & 1) {
                } else if (may_overrun && whole <= 32u && (dist >= len || dist < 8u)) {
                    // (dist >= len is enough: the loads may reach into the match's own, not yet written destination, but what
                    //  they read there lands at offsets >= dist >= len, in the overrun.  dist < 8: the first 8 loaded bytes
                    //  hold the period, the stored words are built from it below)
                    if (!(IRF_INFLATE_EXP & 16)) { dsrc = src; ddst = dst; need = whole; pat_d = dist < len ? dist : 0u; }
                } else if (IRF_INFLATE_EXP & 8) {
                } else {
                    if (dist >= 8) {
                        if (dist >= 32) {
                            while (len >= 32) {          // one round trip per 32 bytes
                                uint32_t w[8];
#pragma unroll
                                for (int k = 0; k < 8; k++) w[k] = ld32u(src + 4 * k);
#pragma unroll
                                for (int k = 0; k < 8; k++) st32u(dst + 4 * k, w[k]);
                                src += 32; dst += 32; len -= 32;
                            }
                        }
                        while (len >= 8) {
                            const uint32_t w0 = ld32u(src), w1 = ld32u(src + 4);
                            st32u(dst, w0);
                            st32u(dst + 4, w1);
                            src += 8; dst += 8; len -= 8;
                        }
                        if (len && may_overrun) {
                            const uint32_t w0 = ld32u(src), w1 = ld32u(src + 4);
                            st32u(dst, w0);
                            st32u(dst + 4, w1);
                            len = 0;
                        }
                    } else if (may_overrun) {
                        // the match overlaps its own output: a pattern of period dist, built in registers from one load
                        uint64_t pat = (uint64_t)ld32u(src) | (uint64_t)ld32u(src + 4) << 32;
                        pat &= ~0ull >> (64u - 8u * dist);
                        for (uint32_t sh = dist; sh < 8; sh <<= 1) pat |= pat << (8u * sh);
                        const uint32_t step = dist == 3 ? 6u : dist == 5 ? 5u : dist == 6 ? 6u : dist == 7 ? 7u : 8u;   // multiple of dist
                        while (true) {
                            st32u(dst, (uint32_t)pat);
                            st32u(dst + 4, (uint32_t)(pat >> 32));
                            if (len <= step) break;
                            dst += step; len -= step;
                        }
                        len = 0;
                    }
                    if (dist >= 4) {
                        while (len >= 4) {
                            st32u(dst, ld32u(src));
                            src += 4; dst += 4; len -= 4;
                        }
                    }
                    while (len) {
                        *dst++ = *src++;
                        len--;
                    }
                }
            }
            fb.refill();
            uint32_t w[8];
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = (IRF_INFLATE_EXP & 1) ? 0u : ld32u(dsrc + 4 * k);
            const uint8_t* const hsrc = need > 16u ? dsrc + 16 : out;      // most matches are short: their upper half reads the shared line
#pragma unroll
            for (int k = 0; k < 4; k++) w[4 + k] = (IRF_INFLATE_EXP & 1) ? 0u : ld32u(hsrc + 4 * k);
            // every lane reloads its window (the 16 bytes at p) in every iteration; the loaded words are taken over BEHIND the
            // literal step below, which still refills from the old window (it holds at least one word: an iteration starts
            // with three or four and has taken at most two) — so the one wait of the iteration, for the window and the copy
            // sources, comes a symbol later, when an L2 round trip is over.  A lane with a deferred copy has no pending
            // literal bytes (flush_before_match), so the literal below stores nothing in front of the copy's stores.
            static_assert(IRF_LITERAL_RUN <= 1, "the old window has one word left for the literal step");
            uint4 nw;
            __builtin_memcpy(&nw, fb.p, 16);
            const uint8_t* const p_loaded = fb.p;
            sym = decode_sym(s_tab, lane, lit, LIT_SYM, fb);
#pragma unroll
            for (int extra = 0; extra < IRF_LITERAL_RUN; extra++) {
                if ((uint32_t)sym < 256u && op < isize) {
                    put((uint32_t)sym);
                    fb.refill();
                    sym = decode_sym(s_tab, lane, lit, LIT_SYM, fb);
                }
            }
            asm volatile("" ::"v"(nw.x), "v"(nw.y), "v"(nw.z), "v"(nw.w));      // the one wait
            const bool shifted = fb.p != p_loaded;          // (by one word)
            fb.q0 = shifted ? nw.y : nw.x;
            fb.q1 = shifted ? nw.z : nw.y;
            fb.q2 = shifted ? nw.w : nw.z;
            fb.q3 = nw.w;
            if (pat_d) {
                // word j of the destination = the period starting at phase 8j mod dist
                uint64_t pat = (uint64_t)w[0] | (uint64_t)w[1] << 32;
                pat &= ~0ull >> (64u - 8u * pat_d);
                for (uint32_t sh = pat_d; sh < 8; sh <<= 1) pat |= pat << (8u * sh);
                const uint32_t r1 = (0x12302000u >> (4u * pat_d)) & 7u;        // 8 mod dist, dist = 1 .. 7
                uint32_t r = 0;
#pragma unroll
                for (int j = 1; j < 4; j++) {
                    r += r1;
                    if (r >= pat_d) r -= pat_d;
                    const uint64_t pj = r ? (pat >> (8u * r)) | (pat << (8u * (pat_d - r))) : pat;
                    w[2 * j] = (uint32_t)pj; w[2 * j + 1] = (uint32_t)(pj >> 32);
                }
                w[0] = (uint32_t)pat; w[1] = (uint32_t)(pat >> 32);
            }
            if (need) {
                st32u(ddst, w[0]); st32u(ddst + 4, w[1]);
                if (need > 8u) { st32u(ddst + 8, w[2]); st32u(ddst + 12, w[3]); }
                if (need > 16u) { st32u(ddst + 16, w[4]); st32u(ddst + 20, w[5]); }
                if (need > 24u) { st32u(ddst + 24, w[6]); st32u(ddst + 28, w[7]); }
            }
        }
        fb.to(b);
    }
    flush();
    if (!bad && (op != isize || consumed_bits() > (uint64_t)m.clen * 8ull)) bad = true;
    status[mi] = bad ? BGZF_BAD_DEFLATE : BGZF_OK;
}

// ---- CRC-32 -------------------------------------------------------------------------------------------------------
constexpr uint32_t kCrcPoly = 0xEDB88320u;
struct CrcPowers { uint32_t x2n[32]; };       // x^(2^k) mod P, reflected (bit 31 = x^0)

__host__ __device__ inline uint32_t crc_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & (0x80000000u >> i)) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ kCrcPoly : b >> 1;
    }
    return p;
}

// One wave per member: lane l takes the l-th of 64 equal slices; crc(A||B) = crc(A) * x^(8|B|) + crc(B) (mod P), so the
// member's CRC is the XOR over lanes of crc(slice) * x^(8 * bytes behind the slice).  A lane reads its slice 64 bytes at
// a time (four 16-byte loads issued together: a cache line is fetched once or twice, not once per dword) and folds a
// dword per step with four independent table lookups (slicing by 4) instead of four dependent ones.
__global__ __launch_bounds__(256) void bgzf_crc_kernel(const uint8_t* __restrict__ out, const BgzfMember* __restrict__ members,
                                                       uint32_t n_members, uint32_t* status, CrcPowers pw) {
    __shared__ uint32_t s_t[4][256];      // s_t[k][b]: CRC of byte b followed by k zero bytes
    {
        uint32_t c = threadIdx.x;
        for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ kCrcPoly : c >> 1;
        s_t[0][threadIdx.x] = c;
        wg_barrier();
        for (int k = 1; k < 4; k++) {
            c = (c >> 8) ^ s_t[0][c & 0xffu];
            s_t[k][threadIdx.x] = c;
        }
    }
    wg_barrier();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t mi = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (mi >= n_members) return;
    if (status[mi] != BGZF_OK) return;
    const BgzfMember m = members[mi];
    const uint32_t S = (m.isize + 63u) / 64u;
    const uint32_t beg = min(lane * S, m.isize), end = min(beg + S, m.isize);
    const uint8_t* p = out + m.out_off + beg;
    uint32_t n = end - beg, c = 0xffffffffu;
    auto fold = [&](uint32_t w) {
        c ^= w;
        c = s_t[3][c & 0xffu] ^ s_t[2][(c >> 8) & 0xffu] ^ s_t[1][(c >> 16) & 0xffu] ^ s_t[0][c >> 24];
    };
    while (n >= 64) {
        uint4 q[4];
        __builtin_memcpy(q, p, 64);
#pragma unroll
        for (int k = 0; k < 4; k++) { fold(q[k].x); fold(q[k].y); fold(q[k].z); fold(q[k].w); }
        p += 64; n -= 64;
    }
    while (n >= 4) {
        fold(ld32u(p));
        p += 4; n -= 4;
    }
    while (n) {
        c = s_t[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
        n--;
    }
    c = ~c;
    uint32_t contrib = 0;
    if (end > beg) {
        const uint32_t tail = m.isize - end;
        uint32_t pwr = 0x80000000u;
        for (int k = 0; k < 17; k++)
            if ((tail >> k) & 1u) pwr = crc_mulmod(pw.x2n[k + 3], pwr);
        contrib = crc_mulmod(pwr, c);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) contrib ^= __shfl_xor(contrib, d);
    if (lane == 0 && contrib != m.crc) status[mi] = BGZF_BAD_CRC;
}

__global__ __launch_bounds__(256) void first_bad_kernel(const uint32_t* __restrict__ status, uint32_t n, uint32_t* first_bad) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n && status[i] != BGZF_OK) atomicMin(first_bad, i);
}

CrcPowers crc_powers() {
    CrcPowers pw;
    pw.x2n[0] = 0x40000000u;
    for (int k = 1; k < 32; k++) pw.x2n[k] = crc_mulmod(pw.x2n[k - 1], pw.x2n[k - 1]);
    return pw;
}

}   // namespace

void launch_bgzf_inflate(const uint8_t* in, const BgzfMember* members, uint32_t n_members, uint8_t* out, uint32_t* status,
                         hipStream_t s) {
    if (!n_members) return;
    hipLaunchKernelGGL(bgzf_inflate_kernel, dim3((n_members + kLanes - 1) / kLanes), dim3(kLanes), 0, s, in, members, n_members, out,
                       status);
}

uint32_t bgzf_inflate_slots() {
    int dev_id = 0, cus = 0, wgs = 0;
    if (hipGetDevice(&dev_id) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&wgs, bgzf_inflate_kernel, kLanes, 0) != hipSuccess || cus <= 0 || wgs <= 0)
        return 0;
    return (uint32_t)cus * (uint32_t)wgs * (uint32_t)kLanes;
}

void launch_bgzf_crc(const uint8_t* out, const BgzfMember* members, uint32_t n_members, uint32_t* status, hipStream_t s) {
    if (!n_members) return;
    static const CrcPowers pw = crc_powers();
    hipLaunchKernelGGL(bgzf_crc_kernel, dim3((n_members + 3) / 4), dim3(256), 0, s, out, members, n_members, status, pw);
}

void launch_first_bad(const uint32_t* status, uint32_t n_members, uint32_t* first_bad, hipStream_t s) {
    if (!n_members) return;
    hipLaunchKernelGGL(first_bad_kernel, dim3((n_members + 255) / 256), dim3(256), 0, s, status, n_members, first_bad);
}

}   // namespace dev
}   // namespace irf
