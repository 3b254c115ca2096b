/*
 * reads.cpp — seeded synthetic paired-end alignment generator (input synthesis
 * for tests and bench; SURVEY.md §8 d).  Not on the counting path.
 *
 * Emits raw BAM alignment records (SAM spec §4.2) for read pairs sampled from a
 * transcript model: transcript ~ abundance, optional retained intron, fragment
 * length ~ N(mean, sd), mates from opposite ends projected through the exon
 * structure to M/N CIGARs, a few D/I edits and soft clips, intergenic noise.
 * Pair i depends only on (seed, i): any range of pairs can be generated
 * independently (chunked, multi-threaded or sharded) with identical bytes.
 */
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

extern "C" {

typedef struct synth_model {
    uint32_t n_chr;
    uint32_t n_tx;
    const uint32_t* chr_len;
    const double* chr_cdf;        /* n_chr, cumulative length fraction (noise reads)  */
    const int32_t* tx_chr;
    const int8_t* tx_strand;      /* 1 '+', 0 '-' */
    const int64_t* tx_ex_off;     /* n_tx + 1 */
    const uint32_t* ex_start;
    const uint32_t* ex_end;
    const double* tx_cdf;         /* n_tx, cumulative abundance in (0,1] */
    const double* tx_ir_prob;
    uint32_t read_len;
    uint32_t stranded;            /* 0 unstranded (coin flip), 1 dUTP (read 1 antisense) */
    double frag_mean, frag_sd;
    double noise_frac, indel_frac, clip_frac, single_frac;
    uint64_t seed;
    uint32_t with_seq;            /* 1: write l_seq + seq/qual bytes (realistic record size) */
    uint32_t reserved;
} synth_model;

}   // extern "C"

namespace {

struct Rng {
    uint64_t s;
    Rng(uint64_t seed, uint64_t idx) {
        s = seed * 0x9E3779B97F4A7C15ull + idx * 0xD1B54A32D192ED03ull + 0x632BE59BD9B4E019ull;
        next(); next();
    }
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return n ? next() % n : 0; }
    double normal() {
        double u1 = uni(), u2 = uni();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
};

const int MAXB = 80;

struct Mol { int n; uint32_t s[MAXB], e[MAXB]; uint64_t len; };

struct Aln { uint32_t pos; int ncig; uint32_t cig[2 * MAXB + 8]; uint32_t qlen; };

/* project molecule coordinates [a,b) to the genome */
void project(const Mol& m, uint64_t a, uint64_t b, Aln& out) {
    out.ncig = 0; out.qlen = 0; out.pos = 0;
    uint64_t cum = 0;
    bool started = false;
    uint32_t prev_end = 0;
    for (int k = 0; k < m.n && a < b; k++) {
        uint64_t bl = m.e[k] - m.s[k];
        if (a < cum + bl) {
            uint64_t o = a - cum;
            uint64_t take = bl - o;
            if (take > b - a) take = b - a;
            uint32_t gs = m.s[k] + (uint32_t)o;
            if (!started) { out.pos = gs; started = true; }
            else out.cig[out.ncig++] = ((gs - prev_end) << 4) | 3u;      /* N */
            out.cig[out.ncig++] = ((uint32_t)take << 4) | 0u;            /* M */
            out.qlen += (uint32_t)take;
            prev_end = gs + (uint32_t)take;
            a += take;
        }
        cum += bl;
    }
}

/* cosmetic edits that exercise cigar2block: D / I inside a long M, soft clips */
void edit(Aln& a, Rng& r, double indel_frac, double clip_frac, bool left_mate) {
    if (r.uni() < indel_frac) {
        for (int i = 0; i < a.ncig; i++) {
            uint32_t op = a.cig[i] & 0xf, l = a.cig[i] >> 4;
            if (op == 0 && l >= 24) {
                bool del = r.next() & 1;
                uint32_t x = l / 2;
                uint32_t ins[3];
                if (del) { ins[0] = (x << 4); ins[1] = (2u << 4) | 2u; ins[2] = ((l - x - 2) << 4); a.qlen -= 2; }
                else     { ins[0] = (x << 4); ins[1] = (3u << 4) | 1u; ins[2] = ((l - x) << 4); a.qlen += 3; }
                memmove(a.cig + i + 3, a.cig + i + 1, sizeof(uint32_t) * (size_t)(a.ncig - i - 1));
                a.cig[i] = ins[0]; a.cig[i + 1] = ins[1]; a.cig[i + 2] = ins[2];
                a.ncig += 2;
                break;
            }
        }
    }
    if (r.uni() < clip_frac) {
        uint32_t c = 3 + (uint32_t)r.below(6);
        if (left_mate) {
            if ((a.cig[0] & 0xf) == 0 && (a.cig[0] >> 4) > c + 10) {
                a.cig[0] -= c << 4;
                a.pos += c;
                memmove(a.cig + 1, a.cig, sizeof(uint32_t) * (size_t)a.ncig);
                a.cig[0] = (c << 4) | 4u;
                a.ncig++;
            }
        } else {
            int z = a.ncig - 1;
            if ((a.cig[z] & 0xf) == 0 && (a.cig[z] >> 4) > c + 10) {
                a.cig[z] -= c << 4;
                a.cig[a.ncig++] = (c << 4) | 4u;
            }
        }
    }
}

struct Writer {
    uint8_t* out; size_t cap; size_t n;
    void put(const void* p, size_t k) {
        if (out && n + k <= cap) memcpy(out + n, p, k);
        n += k;
    }
    void fill(uint8_t v, size_t k) {
        if (out && n + k <= cap) memset(out + n, v, k);
        n += k;
    }
};

void write_record(Writer& w, int32_t ref, const Aln& a, uint16_t flag, int32_t mref, int32_t mpos, uint64_t pair_idx,
                  uint32_t with_seq) {
    char name[24];
    int ln = snprintf(name, sizeof name, "p%llx", (unsigned long long)pair_idx) + 1;
    int32_t l_seq = with_seq ? (int32_t)a.qlen : 0;
    int32_t block_size = 32 + ln + 4 * a.ncig + (l_seq + 1) / 2 + l_seq;
    int32_t pos = (int32_t)a.pos;
    uint8_t l_read_name = (uint8_t)ln, mapq = 255;
    uint16_t bin = 0, n_cigar = (uint16_t)a.ncig;
    int32_t tlen = 0;
    w.put(&block_size, 4); w.put(&ref, 4); w.put(&pos, 4); w.put(&l_read_name, 1); w.put(&mapq, 1); w.put(&bin, 2);
    w.put(&n_cigar, 2); w.put(&flag, 2); w.put(&l_seq, 4); w.put(&mref, 4); w.put(&mpos, 4); w.put(&tlen, 4);
    w.put(name, (size_t)ln);
    w.put(a.cig, 4 * (size_t)a.ncig);
    if (l_seq && with_seq == 2) {
        /* random bases (2 bits of entropy per base) and binned qualities: compresses about as a real RNA-seq BAM does */
        Rng r(0x5eed, pair_idx * 2 + ((flag & 0x80) ? 1 : 0));
        static const uint8_t nib[4] = {1, 2, 4, 8};
        static const uint8_t qbin[8] = {37, 37, 37, 37, 37, 23, 12, 2};
        uint8_t tmp[1024];
        size_t nb = (size_t)(l_seq + 1) / 2;
        for (size_t i = 0; i < nb; i += sizeof tmp) {
            size_t k = nb - i < sizeof tmp ? nb - i : sizeof tmp;
            for (size_t j = 0; j < k; j++) { uint64_t x = r.next(); tmp[j] = (uint8_t)(nib[x & 3] << 4 | nib[(x >> 2) & 3]); }
            w.put(tmp, k);
        }
        for (size_t i = 0; i < (size_t)l_seq; i += sizeof tmp) {
            size_t k = (size_t)l_seq - i < sizeof tmp ? (size_t)l_seq - i : sizeof tmp;
            uint64_t x = 0;
            for (size_t j = 0; j < k; j++) { if ((j & 15) == 0) x = r.next(); tmp[j] = qbin[x & 7]; x >>= 3; }
            w.put(tmp, k);
        }
    } else if (l_seq) { w.fill(0x11, (size_t)(l_seq + 1) / 2); w.fill(30, (size_t)l_seq); }
}

int upper_cdf(const double* cdf, int n, double u) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    return lo;
}

}   // namespace

extern "C" {

/* Writes the BAM records of pairs [first, first+n) into out (if it fits) and
 * returns the number of bytes the records need.  n_records_out (optional)
 * receives the record count.                                                */
size_t synth_bam_records(const synth_model* M, uint64_t first, uint64_t n, uint8_t* out, size_t cap,
                         uint64_t* n_records_out) {
    Writer w{out, cap, 0};
    uint64_t nrec = 0;
    for (uint64_t i = first; i < first + n; i++) {
        Rng r(M->seed, i);
        Mol mol;
        int32_t ref;
        int strand;
        double u = r.uni();
        if (u < M->noise_frac || M->n_tx == 0) {
            ref = upper_cdf(M->chr_cdf, (int)M->n_chr, r.uni());
            uint32_t L = M->chr_len[ref];
            uint32_t span = 2000 < L ? 2000 : L;
            uint32_t s = (uint32_t)r.below(L - span + 1);
            mol.n = 1; mol.s[0] = s; mol.e[0] = s + span; mol.len = span;
            strand = (int)(r.next() & 1);
        } else {
            int t = upper_cdf(M->tx_cdf, (int)M->n_tx, r.uni());
            ref = M->tx_chr[t];
            strand = M->tx_strand[t];
            int64_t a = M->tx_ex_off[t], b = M->tx_ex_off[t + 1];
            int ne = (int)(b - a);
            if (ne > MAXB) ne = MAXB;
            int retain = -1;
            if (ne > 1 && r.uni() < M->tx_ir_prob[t]) retain = (int)r.below((uint64_t)(ne - 1));
            mol.n = 0; mol.len = 0;
            for (int k = 0; k < ne; k++) {
                if (retain >= 0 && k == retain + 1) { mol.e[mol.n - 1] = M->ex_end[a + k]; }
                else { mol.s[mol.n] = M->ex_start[a + k]; mol.e[mol.n] = M->ex_end[a + k]; mol.n++; }
            }
            for (int k = 0; k < mol.n; k++) mol.len += mol.e[k] - mol.s[k];
        }
        double fl = M->frag_mean + M->frag_sd * r.normal();
        uint64_t L = fl < 1 ? 1 : (uint64_t)(fl + 0.5);
        if (L < M->read_len) L = M->read_len;
        if (L > 1000) L = 1000;
        if (L > mol.len) L = mol.len;
        uint64_t rl = M->read_len < L ? M->read_len : L;
        uint64_t start = r.below(mol.len - L + 1);
        Aln A, B;
        project(mol, start, start + rl, A);
        project(mol, start + L - rl, start + L, B);
        edit(A, r, M->indel_frac, M->clip_frac, true);
        edit(B, r, M->indel_frac, M->clip_frac, false);
        /* which mate is read 1: dUTP => read 1 is antisense to the transcript */
        bool left_is_read1;
        if (M->stranded) left_is_read1 = (strand == 0);
        else left_is_read1 = (r.next() & 1) != 0;
        if (r.uni() < M->single_frac) {
            uint16_t f = (r.next() & 1) ? 0x10 : 0;
            write_record(w, ref, A, f, -1, -1, i, M->with_seq);
            nrec += 1;
            continue;
        }
        uint16_t fl_left = 0x1 | 0x2 | 0x20 | (left_is_read1 ? 0x40 : 0x80);
        uint16_t fl_right = 0x1 | 0x2 | 0x10 | (left_is_read1 ? 0x80 : 0x40);
        /* stream order: read 1 first */
        if (left_is_read1) {
            write_record(w, ref, A, fl_left, ref, (int32_t)B.pos, i, M->with_seq);
            write_record(w, ref, B, fl_right, ref, (int32_t)A.pos, i, M->with_seq);
        } else {
            write_record(w, ref, B, fl_right, ref, (int32_t)A.pos, i, M->with_seq);
            write_record(w, ref, A, fl_left, ref, (int32_t)B.pos, i, M->with_seq);
        }
        nrec += 2;
    }
    if (n_records_out) *n_records_out = nrec;
    return w.n;
}

}   // extern "C"
