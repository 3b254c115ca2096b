// Sanitizer driver for the HOST code (SURVEY §5 "Race detection / sanitizers"): the threaded BGZF reader
// (irfinder_amd/host/bam_reader.cpp behind irf_reader_*), the reference-table parsers and the CPU oracle, built with
// -fsanitize=address,undefined or -fsanitize=thread by tests/sanitize/Makefile and run by tests/test_sanitizers.py on
// the CPU (never on the GPU box).  usage: host_sanitize <raw.bam> <bgzf.bam> <ref-cover> <ref-sj> <ref-read-continues> <ref-ROI> <outdir>
// Checks while it runs: the reader gives the same batch and counters for raw / BGZF input, 1 / 4 inflate threads and
// ragged feeds, with adjacent-mate and name-joined pairing; a corrupt member is refused; the oracle digests the same file.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "../../include/irf.h"
#include "../../irfinder_amd/host/ref_tables.h"

extern "C" {
void orc_params_default(irf_params* p);
void* orc_create(const irf_params* params, const char* const bufs[4], const size_t lens[4]);
void orc_destroy(void* h);
int orc_process_bam(void* h, const uint8_t* bam, size_t len);
int orc_finalize(void* h);
int orc_write_outputs(void* h, const char* outdir);
int orc_reader_stats(void* h, irf_reader_stats* out);
const char* orc_last_error(void* h);
}

// irf_params_default lives in engine.hip (HIP); this CPU-only build takes the oracle's identical defaults
extern "C" void irf_params_default(irf_params* p) { orc_params_default(p); }

static std::vector<uint8_t> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #c); exit(1); } } while (0)

struct Run { std::vector<uint32_t> u32[6]; std::vector<uint8_t> u8[2]; irf_reader_stats st; };

static Run read_all(const irf_params& P, const std::vector<uint8_t>& bytes, bool bgzf, int threads, size_t piece) {
    irf_reader* r = nullptr;
    CHECK(irf_reader_create(&P, &r) == 0);
    CHECK(irf_reader_set_threads(r, threads) == 0);
    for (size_t at = 0; at < bytes.size();) {
        const size_t n = std::min(piece, bytes.size() - at);
        const int rc = irf_reader_feed(r, bytes.data() + at, n, bgzf ? 1 : 0);
        if (rc != 0) { fprintf(stderr, "feed: %s\n", irf_reader_last_error(r)); exit(1); }
        at += n;
        piece = piece * 3 / 2 + 7;                 // ragged feeds
    }
    CHECK(irf_reader_finish(r) == 0);
    irf_batch b;
    CHECK(irf_reader_batch(r, &b) == 0);
    Run out;
    out.u32[0].assign(b.frag_chr, b.frag_chr + b.n_frag);
    out.u32[1].assign(b.frag_start, b.frag_start + b.n_frag);
    out.u32[2].assign(b.frag_end, b.frag_end + b.n_frag);
    out.u32[3].assign(b.frag_blk_off, b.frag_blk_off + b.n_frag + 1);
    out.u32[4].assign(b.blk_start, b.blk_start + b.n_blk);
    out.u32[5].assign(b.blk_len, b.blk_len + b.n_blk);
    out.u8[0].assign(b.frag_flags, b.frag_flags + b.n_frag);
    out.u8[1].assign(b.blk_read, b.blk_read + b.n_blk);
    CHECK(irf_reader_get_stats(r, &out.st) == 0);
    irf_reader_destroy(r);
    return out;
}
static bool same(const Run& a, const Run& b) {
    for (int i = 0; i < 6; i++) if (a.u32[i] != b.u32[i]) return false;
    for (int i = 0; i < 2; i++) if (a.u8[i] != b.u8[i]) return false;
    return memcmp(&a.st, &b.st, sizeof a.st) == 0;
}

int main(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: see the header of this file\n"); return 2; }
    const std::vector<uint8_t> raw = slurp(argv[1]), z = slurp(argv[2]);
    irf_params P;
    orc_params_default(&P);
    for (int join = 0; join < 2; join++) {
        P.mate_join = (uint32_t)join;
        const Run a = read_all(P, raw, false, 1, 1000);
        const Run b = read_all(P, z, true, 1, 4096);
        const Run c = read_all(P, z, true, 4, 70001);
        const Run d = read_all(P, z, true, 4, z.size());
        CHECK(same(a, b));
        CHECK(same(a, c));
        CHECK(same(a, d));
        CHECK(a.st.n_records > 0 && a.u32[0].size() > 0);
    }
    P.mate_join = 0;
    {   // a corrupt member in the last quarter: refused, with 1 and with 4 threads
        std::vector<uint8_t> bad = z;
        bad[bad.size() * 7 / 8] ^= 0x5a;
        for (int th = 1; th <= 4; th += 3) {
            irf_reader* r = nullptr;
            CHECK(irf_reader_create(&P, &r) == 0);
            CHECK(irf_reader_set_threads(r, th) == 0);
            int rc = irf_reader_feed(r, bad.data(), bad.size(), 1);
            if (rc == 0) rc = irf_reader_finish(r);
            CHECK(rc != 0);
            irf_reader_destroy(r);
        }
    }
    {   // cigar2block on every op, and on a cigar that overflows the caller's arrays
        const uint32_t cig[] = {5u << 4 | 4, 45u << 4 | 0, 2u << 4 | 2, 5u << 4 | 7, 100u << 4 | 3, 3u << 4 | 1, 50u << 4 | 8, 1u << 4 | 5, 2u << 4 | 6};
        uint32_t st[4], ln[4], nb = 0, gl = 0;
        CHECK(irf_cigar2block(cig, 9, st, ln, 4, &nb, &gl) == 0 && nb == 2 && st[1] == 152 && ln[0] == 52 && gl == 202);
        CHECK(irf_cigar2block(cig, 9, st, ln, 1, &nb, &gl) == IRF_E_RANGE);
    }
    {   // the reference-table parsers and the oracle on the same sample
        std::vector<uint8_t> f[4] = {slurp(argv[3]), slurp(argv[4]), slurp(argv[5]), slurp(argv[6])};
        irf::RefTables t;
        std::string err;
        CHECK(irf::parse_cover_bed((const char*)f[0].data(), f[0].size(), t, err));
        CHECK(irf::parse_sj_ref((const char*)f[1].data(), f[1].size(), t, err));
        CHECK(irf::parse_read_continues_ref((const char*)f[2].data(), f[2].size(), t, err));
        CHECK(irf::parse_roi_bed((const char*)f[3].data(), f[3].size(), t, err));
        irf::finish_tables(t);
        CHECK(t.n_intron() > 0);
        std::string junk = "chr1\t10\tnot-a-number\n";
        irf::RefTables t2;
        CHECK(!irf::parse_sj_ref(junk.data(), junk.size(), t2, err));
        const char* bufs[4] = {(const char*)f[0].data(), (const char*)f[1].data(), (const char*)f[2].data(), (const char*)f[3].data()};
        const size_t lens[4] = {f[0].size(), f[1].size(), f[2].size(), f[3].size()};
        void* o = orc_create(&P, bufs, lens);
        CHECK(o != nullptr);
        CHECK(orc_process_bam(o, raw.data(), raw.size()) == 0);
        CHECK(orc_finalize(o) == 0);
        CHECK(orc_write_outputs(o, argv[7]) == 0);
        irf_reader_stats os;
        orc_reader_stats(o, &os);
        const Run a = read_all(P, raw, false, 1, 1 << 20);
        CHECK(memcmp(&os, &a.st, sizeof os) == 0);           // the product's reader and the oracle's agree on the counters
        orc_destroy(o);
    }
    puts("host_sanitize ok");
    return 0;
}
