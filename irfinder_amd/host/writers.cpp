// writers.cpp — the six IRFinder text outputs, from the in-memory results.
//
// Replaces the writers at the end of ⟨upstream⟩ irfinder.cpp main() and the
// WriteOutput methods of JunctionCount, SpansPoint, FragmentsInROI,
// FragmentsInChr (ReadBlockProcessor.cpp) and CoverageBlocksIRFinder
// (ReadBlockProcessor_CoverageBlocks.cpp).  Formats: SURVEY App. B.4 (recalled;
// column order of the four small files is OQ-11).  Upstream iterates
// std::map<string,…>, so per-chromosome sections come out in lexicographic
// chromosome-name order; doubles use the default ostream format = "%.{p}g".
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/irf.h"
#include "ref_tables.h"

namespace irf {
const RefTables& ctx_ref(const irf_ctx* c);
const irf_params& ctx_params(const irf_ctx* c);
const std::vector<uint32_t>& ctx_refchr_slot(const irf_ctx* c);
const std::vector<std::string>& ctx_slot_names(const irf_ctx* c);
int ctx_fail(irf_ctx* c, int code, const std::string& msg);
}   // namespace irf

namespace {

struct Out {
    FILE* f = nullptr;
    std::string buf;
    bool open(const std::string& path) { f = fopen(path.c_str(), "wb"); buf.reserve(1 << 20); return f != nullptr; }
    void flush() { if (f && !buf.empty()) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); } }
    bool close() { flush(); bool ok = f && fclose(f) == 0; f = nullptr; return ok; }
    void s(const std::string& v) { buf += v; }
    void c(char ch) { buf += ch; }
    void u(unsigned long long v) { char t[24]; int n = snprintf(t, sizeof t, "%llu", v); buf.append(t, (size_t)n); }
    void d(double v, int prec) { char t[40]; int n = snprintf(t, sizeof t, "%.*g", prec, v); buf.append(t, (size_t)n); }
    void maybe_flush() { if (buf.size() > (1u << 20) - 4096) flush(); }
};

const char* kWarn[] = {"-", "LowCover", "LowSplicing", "MinorIsoform", "NonUniformIntronCover"};

bool write_ir(const std::string& path, const irf::RefTables& R, const irf_ir_table& T, uint32_t n, int prec) {
    Out o;
    if (!o.open(path)) return false;
    o.s("Chr\tStart\tEnd\tName\tNull\tStrand\tExcludedBases\tCoverage\tIntronDepth\t"
        "IntronDepth25Percentile\tIntronDepth50Percentile\tIntronDepth75Percentile\t"
        "ExonToIntronReadsLeft\tExonToIntronReadsRight\tIntronDepthFirst50bp\tIntronDepthLast50bp\t"
        "SpliceLeft\tSpliceRight\tSpliceExact\tIRratio\tWarnings\n");
    for (uint32_t i = 0; i < n; i++) {
        o.s(R.chr_names[R.in_chr[i]]); o.c('\t'); o.u(R.in_start[i]); o.c('\t'); o.u(R.in_end[i]); o.c('\t');
        o.s(R.in_name[i]); o.c('\t'); o.s(R.in_score[i]); o.c('\t'); o.c(R.in_strand[i]); o.c('\t');
        o.u((unsigned long long)(R.in_end[i] - R.in_start[i]) - R.in_measured[i]); o.c('\t');
        o.d(T.coverage[i], prec); o.c('\t'); o.d(T.intron_depth[i], prec); o.c('\t');
        o.d(T.p25[i], prec); o.c('\t'); o.d(T.p50[i], prec); o.c('\t'); o.d(T.p75[i], prec); o.c('\t');
        o.u(T.sp_left[i]); o.c('\t'); o.u(T.sp_right[i]); o.c('\t');
        o.d(T.first50[i], prec); o.c('\t'); o.d(T.last50[i], prec); o.c('\t');
        o.u(T.splice_left[i]); o.c('\t'); o.u(T.splice_right[i]); o.c('\t'); o.u(T.splice_exact[i]); o.c('\t');
        o.d(T.irratio[i], prec); o.c('\t');
        o.s(kWarn[T.warning[i] <= 4 ? T.warning[i] : 0]); o.c('\n');
        o.maybe_flush();
    }
    return o.close();
}

}   // namespace

extern "C" int irf_write_outputs(irf_ctx* c, const char* outdir) {
    if (!c || !outdir) return IRF_E_ARG;
    irf_results r;
    int rc = irf_get_results(c, &r);
    if (rc != 0) return rc;
    const irf::RefTables& R = irf::ctx_ref(c);
    const irf_params& P = irf::ctx_params(c);
    const std::vector<std::string>& slots = irf::ctx_slot_names(c);
    const std::vector<uint32_t>& refslot = irf::ctx_refchr_slot(c);
    const std::string dir(outdir);
    const int prec = (int)P.out_precision;

    // chromosome slots in lexicographic name order (upstream's std::map<string,...> iteration order)
    std::vector<uint32_t> lex(slots.size());
    std::iota(lex.begin(), lex.end(), 0u);
    std::sort(lex.begin(), lex.end(), [&](uint32_t a, uint32_t b) { return slots[a] < slots[b]; });
    // duplicate BAM names would collapse into one map entry upstream; keep the first slot only
    std::vector<uint32_t> rank_of(slots.size());
    for (size_t k = 0; k < lex.size(); k++) rank_of[lex[k]] = (uint32_t)k;

    {   // IRFinder-ROI.txt : name total neg pos ; per chromosome (lexicographic), by end within it
        Out o;
        if (!o.open(dir + "/IRFinder-ROI.txt")) return irf::ctx_fail(c, IRF_E_IO, "cannot write IRFinder-ROI.txt");
        std::vector<uint32_t> idx(r.n_roi);
        std::iota(idx.begin(), idx.end(), 0u);
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
            uint32_t ra = rank_of[refslot[R.roi_chr[a]]], rb = rank_of[refslot[R.roi_chr[b]]];
            if (ra != rb) return ra < rb;
            return R.roi_end[a] < R.roi_end[b];
        });
        for (uint32_t i : idx) {
            unsigned long long neg = r.roi_cnt[i], pos = r.roi_cnt[(size_t)r.n_roi + i];
            o.s(R.roi_name[i]); o.c('\t'); o.u(neg + pos); o.c('\t'); o.u(neg); o.c('\t'); o.u(pos); o.c('\n');
            o.maybe_flush();
        }
        if (!o.close()) return irf::ctx_fail(c, IRF_E_IO, "write error on IRFinder-ROI.txt");
    }
    {   // IRFinder-ChrCoverage.txt : chr total neg pos ; BAM header order
        Out o;
        if (!o.open(dir + "/IRFinder-ChrCoverage.txt")) return irf::ctx_fail(c, IRF_E_IO, "cannot write IRFinder-ChrCoverage.txt");
        for (uint32_t i = 0; i < r.n_chr; i++) {
            unsigned long long neg = r.chr_cnt[2 * (size_t)i], pos = r.chr_cnt[2 * (size_t)i + 1];
            o.s(slots[i]); o.c('\t'); o.u(neg + pos); o.c('\t'); o.u(neg); o.c('\t'); o.u(pos); o.c('\n');
        }
        if (!o.close()) return irf::ctx_fail(c, IRF_E_IO, "write error on IRFinder-ChrCoverage.txt");
    }
    {   // IRFinder-JuncCount.txt : chr start end strand total pos neg
        Out o;
        if (!o.open(dir + "/IRFinder-JuncCount.txt")) return irf::ctx_fail(c, IRF_E_IO, "cannot write IRFinder-JuncCount.txt");
        // rows are sorted by (slot,left,right): find each slot's range, emit slots in name order
        std::vector<uint64_t> first(slots.size() + 1, 0);
        for (uint64_t j = 0; j < r.n_junc; j++) first[r.junc_chr[j] + 1]++;
        for (size_t s = 0; s < slots.size(); s++) first[s + 1] += first[s];
        for (uint32_t s : lex) {
            for (uint64_t j = first[s]; j < first[s + 1]; j++) {
                unsigned long long neg = r.junc_cnt[j], pos = r.junc_cnt[r.n_junc + j];
                uint8_t rd = r.junc_refdir[j];
                o.s(slots[s]); o.c('\t'); o.u(r.junc_left[j]); o.c('\t'); o.u(r.junc_right[j]); o.c('\t');
                o.c(rd == 1 ? '-' : (rd == 2 ? '+' : '.')); o.c('\t');
                o.u(neg + pos); o.c('\t'); o.u(pos); o.c('\t'); o.u(neg); o.c('\n');
                o.maybe_flush();
            }
        }
        if (!o.close()) return irf::ctx_fail(c, IRF_E_IO, "write error on IRFinder-JuncCount.txt");
    }
    {   // IRFinder-SpansPoint.txt : chr pos total pos neg
        Out o;
        if (!o.open(dir + "/IRFinder-SpansPoint.txt")) return irf::ctx_fail(c, IRF_E_IO, "cannot write IRFinder-SpansPoint.txt");
        std::vector<uint32_t> first(slots.size() + 1, 0);
        for (uint32_t j = 0; j < r.n_sp; j++) first[r.sp_chr[j] + 1]++;
        for (size_t s = 0; s < slots.size(); s++) first[s + 1] += first[s];
        for (uint32_t s : lex) {
            for (uint32_t j = first[s]; j < first[s + 1]; j++) {
                unsigned long long neg = r.sp_cnt[j], pos = r.sp_cnt[(size_t)r.n_sp + j];
                o.s(slots[s]); o.c('\t'); o.u(r.sp_pos[j]); o.c('\t'); o.u(neg + pos); o.c('\t'); o.u(pos); o.c('\t'); o.u(neg); o.c('\n');
                o.maybe_flush();
            }
        }
        if (!o.close()) return irf::ctx_fail(c, IRF_E_IO, "write error on IRFinder-SpansPoint.txt");
    }
    if (!write_ir(dir + "/IRFinder-IR-nondir.txt", R, r.nondir, r.n_intron, prec))
        return irf::ctx_fail(c, IRF_E_IO, "cannot write IRFinder-IR-nondir.txt");
    if (r.has_dir && !write_ir(dir + "/IRFinder-IR-dir.txt", R, r.dir, r.n_intron, prec))
        return irf::ctx_fail(c, IRF_E_IO, "cannot write IRFinder-IR-dir.txt");
    return 0;
}
