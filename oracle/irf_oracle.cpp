/*
 * irf_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Single-threaded CPU restatement of IRFinder's read-block counting path
 * (BAM2blocks reader + FragmentsInChr / FragmentsInROI / JunctionCount /
 * SpansPoint / CoverageBlocksIRFinder / FragmentsMap + the text writers).
 *
 * PARITY UNPINNED.  The upstream source (dgao-lab/IRFinder src/irfinder/ sources)
 * is not in /root/reference (that mount holds README.md:1-7 only) and cannot
 * be fetched, and upstream ships no golden vectors for this path.  This file
 * therefore restates SURVEY.md Appendix B (a recalled spec); every function
 * names the upstream class/method it stands for, without line numbers.  It is
 * pinned only by the hand-derived known-answer tests of SURVEY.md App. C
 * (tests/test_oracle_kat.py) and by a brute-force numpy cross-check.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (irfinder_amd/) never links or calls it.
 *
 * Data-structure choices follow what upstream is recalled to use (std::map
 * for junctions, sorted vectors + std::upper_bound/lower_bound for points,
 * ROIs and coverage regions, std::function callbacks from the reader) so that
 * timing this file is a fair stand-in for the CPU engine ("port" baseline).
 */
#include "../include/irf.h"

#include <algorithm>
#include <array>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace orc {

typedef unsigned int uint;

/* ====================================================================== */
/* FragmentBlocks — ⟨upstream⟩ FragmentBlocks.h                            */
struct FragmentBlocks {
    int chr_id = 0;
    bool direction = false;          /* true = fragment from + strand */
    int readCount = 1;
    uint readStart[2] = {0, 0};
    uint readEnd[2] = {0, 0};
    std::vector<int> rStarts[2];     /* relative to readStart[r] */
    std::vector<int> rLens[2];
    std::string readName;
};

/* ReadBlockProcessor — ⟨upstream⟩ ReadBlockProcessor.h                     */
class ReadBlockProcessor {
public:
    virtual ~ReadBlockProcessor() {}
    virtual void ProcessBlocks(const FragmentBlocks&) = 0;
    virtual void ChrMapUpdate(const std::vector<std::string>&) = 0;
};

static std::vector<std::string> split(const std::string& s, char d) {
    std::vector<std::string> out;
    size_t a = 0;
    while (true) {
        size_t b = s.find(d, a);
        if (b == std::string::npos) { out.push_back(s.substr(a)); break; }
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
    return out;
}
static void chomp(std::string& s) {
    while (!s.empty() && (s.back() == '\r' || s.back() == '\n')) s.pop_back();
}

/* ====================================================================== */
/* FragmentsInChr — ⟨upstream⟩ ReadBlockProcessor.cpp                       */
class FragmentsInChr : public ReadBlockProcessor {
public:
    std::vector<std::string> names;
    std::vector<std::array<unsigned long, 2>> cnt;   /* [chr][dir] */
    void ChrMapUpdate(const std::vector<std::string>& chrmap) override {
        names = chrmap;
        cnt.assign(chrmap.size(), {0ul, 0ul});
    }
    void ProcessBlocks(const FragmentBlocks& b) override { cnt.at(b.chr_id)[b.direction ? 1 : 0]++; }
};

/* ====================================================================== */
/* FragmentsInROI — ⟨upstream⟩ ReadBlockProcessor.cpp                       */
class FragmentsInROI : public ReadBlockProcessor {
public:
    struct Roi { uint start, end; uint file_idx; };
    std::vector<std::string> roi_names, roi_chr;       /* file order */
    std::vector<std::array<unsigned long, 2>> cnt;     /* file order, [dir] */
    std::map<std::string, std::vector<Roi>> chrName_roi;   /* sorted by end */
    std::vector<std::vector<Roi>*> chrID_roi;

    void loadRef(std::istream& in) {
        std::string line;
        while (std::getline(in, line)) {
            chomp(line);
            if (line.empty() || line[0] == '#') continue;
            auto f = split(line, '\t');
            if (f.size() < 3) continue;
            Roi r;
            r.start = (uint)std::stoul(f[1]);
            r.end = (uint)std::stoul(f[2]);
            r.file_idx = (uint)roi_names.size();
            roi_names.push_back(f.size() > 3 ? f[3] : std::string("."));
            roi_chr.push_back(f[0]);
            cnt.push_back({0ul, 0ul});
            chrName_roi[f[0]].push_back(r);
        }
        for (auto& kv : chrName_roi)
            std::stable_sort(kv.second.begin(), kv.second.end(),
                             [](const Roi& a, const Roi& b) { return a.end < b.end; });
    }
    void ChrMapUpdate(const std::vector<std::string>& chrmap) override {
        chrID_roi.clear();
        for (auto& n : chrmap) {
            auto it = chrName_roi.find(n);
            chrID_roi.push_back(it == chrName_roi.end() ? nullptr : &it->second);
        }
    }
    void ProcessBlocks(const FragmentBlocks& b) override {
        std::vector<Roi>* v = chrID_roi.at(b.chr_id);
        if (!v) return;
        uint fs = b.readStart[0];
        uint fe = b.readEnd[0];
        if (b.readCount == 2 && b.readEnd[1] > fe) fe = b.readEnd[1];
        /* first ROI whose end is not less than the fragment end */
        auto it = std::lower_bound(v->begin(), v->end(), fe,
                                   [](const Roi& r, uint key) { return r.end < key; });
        if (it == v->end()) return;
        if (it->start <= fs && fe <= it->end) cnt[it->file_idx][b.direction ? 1 : 0]++;
    }
};

/* ====================================================================== */
/* JunctionCount — ⟨upstream⟩ ReadBlockProcessor.cpp                        */
class JunctionCount : public ReadBlockProcessor {
public:
    typedef std::map<std::pair<uint, uint>, std::array<uint, 3>> JMap;   /* {neg,pos,refdir} */
    typedef std::map<uint, std::array<uint, 2>> EMap;
    std::map<std::string, JMap> chrName_junc;
    std::map<std::string, EMap> chrName_juncLeft, chrName_juncRight;
    std::vector<JMap*> chrID_junc;
    std::vector<EMap*> chrID_juncLeft, chrID_juncRight;
    uint min_flank = 5;
    uint dir_ratio = 10;

    void loadRef(std::istream& in) {
        std::string line;
        while (std::getline(in, line)) {
            chomp(line);
            if (line.empty() || line[0] == '#') continue;
            auto f = split(line, '\t');
            if (f.size() < 4) continue;
            uint s = (uint)std::stoul(f[1]), e = (uint)std::stoul(f[2]);
            auto& a = chrName_junc[f[0]][std::make_pair(s, e)];
            if (f[3] == "+") a[2] |= 2u;
            else if (f[3] == "-") a[2] |= 1u;
        }
    }
    void ChrMapUpdate(const std::vector<std::string>& chrmap) override {
        chrID_junc.clear(); chrID_juncLeft.clear(); chrID_juncRight.clear();
        for (auto& n : chrmap) {
            chrID_junc.push_back(&chrName_junc[n]);
            chrID_juncLeft.push_back(&chrName_juncLeft[n]);
            chrID_juncRight.push_back(&chrName_juncRight[n]);
        }
    }
    void ProcessBlocks(const FragmentBlocks& b) override {
        int d = b.direction ? 1 : 0;
        for (int r = 0; r < b.readCount; r++) {
            for (size_t j = 1; j < b.rLens[r].size(); j++) {
                if ((uint)b.rLens[r][j - 1] >= min_flank && (uint)b.rLens[r][j] >= min_flank) {
                    uint left = b.readStart[r] + b.rStarts[r][j - 1] + b.rLens[r][j - 1];
                    uint right = b.readStart[r] + b.rStarts[r][j];
                    (*chrID_junc.at(b.chr_id))[std::make_pair(left, right)][d]++;
                    (*chrID_juncLeft.at(b.chr_id))[left][d]++;
                    (*chrID_juncRight.at(b.chr_id))[right][d]++;
                }
            }
        }
    }
    /* Directional(): vote over junctions of known strand (OQ-5). */
    int Directional() const {
        unsigned long same = 0, opp = 0;
        for (auto& c : chrName_junc)
            for (auto& j : c.second) {
                if (j.second[2] == 1u) { same += j.second[0]; opp += j.second[1]; }
                else if (j.second[2] == 2u) { same += j.second[1]; opp += j.second[0]; }
            }
        if (same > opp * dir_ratio) return 1;
        if (opp > same * dir_ratio) return -1;
        return 0;
    }
    /* plane: 0 neg, 1 pos, 2 both */
    static uint pick2(const std::array<uint, 2>& a, int plane) { return plane == 2 ? a[0] + a[1] : a[plane]; }
    uint lookup(const std::string& chr, uint l, uint r, int plane) const {
        auto c = chrName_junc.find(chr);
        if (c == chrName_junc.end()) return 0;
        auto j = c->second.find(std::make_pair(l, r));
        if (j == c->second.end()) return 0;
        return plane == 2 ? j->second[0] + j->second[1] : j->second[plane];
    }
    uint lookupLeft(const std::string& chr, uint l, int plane) const {
        auto c = chrName_juncLeft.find(chr);
        if (c == chrName_juncLeft.end()) return 0;
        auto j = c->second.find(l);
        return j == c->second.end() ? 0 : pick2(j->second, plane);
    }
    uint lookupRight(const std::string& chr, uint r, int plane) const {
        auto c = chrName_juncRight.find(chr);
        if (c == chrName_juncRight.end()) return 0;
        auto j = c->second.find(r);
        return j == c->second.end() ? 0 : pick2(j->second, plane);
    }
    static char strandSym(uint refdir) { return refdir == 1 ? '-' : (refdir == 2 ? '+' : '.'); }
    /* IRFinder-JuncCount.txt: chr start end strand total pos neg (OQ-11) */
    void WriteOutput(std::ostream& os) const {
        for (auto& c : chrName_junc)
            for (auto& j : c.second)
                os << c.first << '\t' << j.first.first << '\t' << j.first.second << '\t'
                   << strandSym(j.second[2]) << '\t' << (j.second[0] + j.second[1]) << '\t'
                   << j.second[1] << '\t' << j.second[0] << '\n';
    }
};

/* ====================================================================== */
/* SpansPoint — ⟨upstream⟩ ReadBlockProcessor.cpp                           */
class SpansPoint : public ReadBlockProcessor {
public:
    struct Tab { std::vector<uint> pos; std::vector<uint> cnt[2]; };
    std::map<std::string, Tab> chrName_tab;
    std::vector<Tab*> chrID_tab;
    uint overhangLeft = 5, overhangRight = 4;
    bool inclusive = false;

    void setSpanLength(uint l, uint r) { overhangLeft = l; overhangRight = r; }
    void loadRef(std::istream& in) {
        std::string line;
        while (std::getline(in, line)) {
            chomp(line);
            if (line.empty() || line[0] == '#') continue;
            auto f = split(line, '\t');
            if (f.size() < 2) continue;
            chrName_tab[f[0]].pos.push_back((uint)std::stoul(f[1]));
        }
        for (auto& kv : chrName_tab) {
            auto& p = kv.second.pos;
            std::sort(p.begin(), p.end());
            p.erase(std::unique(p.begin(), p.end()), p.end());
            kv.second.cnt[0].assign(p.size(), 0);
            kv.second.cnt[1].assign(p.size(), 0);
        }
    }
    void ChrMapUpdate(const std::vector<std::string>& chrmap) override {
        chrID_tab.clear();
        for (auto& n : chrmap) {
            auto it = chrName_tab.find(n);
            chrID_tab.push_back(it == chrName_tab.end() ? nullptr : &it->second);
        }
    }
    void ProcessBlocks(const FragmentBlocks& b) override {
        Tab* t = chrID_tab.at(b.chr_id);
        if (!t) return;
        int d = b.direction ? 1 : 0;
        for (int r = 0; r < b.readCount; r++) {
            for (size_t j = 0; j < b.rLens[r].size(); j++) {
                uint len = (uint)b.rLens[r][j];
                if (len <= overhangLeft + overhangRight) continue;
                uint bs = b.readStart[r] + b.rStarts[r][j];
                uint s = bs + overhangLeft;            /* first candidate (exclusive unless `inclusive`) */
                uint e = bs + len - overhangRight;     /* last candidate  (exclusive unless `inclusive`) */
                std::vector<uint>::iterator it =
                    inclusive ? std::lower_bound(t->pos.begin(), t->pos.end(), s)
                              : std::upper_bound(t->pos.begin(), t->pos.end(), s);
                for (; it != t->pos.end() && (inclusive ? *it <= e : *it < e); ++it)
                    t->cnt[d][it - t->pos.begin()]++;
            }
        }
    }
    uint lookup(const std::string& chr, uint pos, int plane) const {
        auto c = chrName_tab.find(chr);
        if (c == chrName_tab.end()) return 0;
        auto it = std::lower_bound(c->second.pos.begin(), c->second.pos.end(), pos);
        if (it == c->second.pos.end() || *it != pos) return 0;
        size_t i = it - c->second.pos.begin();
        return plane == 2 ? c->second.cnt[0][i] + c->second.cnt[1][i] : c->second.cnt[plane][i];
    }
    /* IRFinder-SpansPoint.txt: chr pos total pos neg */
    void WriteOutput(std::ostream& os) const {
        for (auto& c : chrName_tab)
            for (size_t i = 0; i < c.second.pos.size(); i++)
                os << c.first << '\t' << c.second.pos[i] << '\t'
                   << (c.second.cnt[0][i] + c.second.cnt[1][i]) << '\t' << c.second.cnt[1][i] << '\t'
                   << c.second.cnt[0][i] << '\n';
    }
};

/* ====================================================================== */
/* CoverageBlock — ⟨upstream⟩ CoverageBlock.cpp.  Upstream keeps sparse
 * start/stop event lists per region; the semantics (App. B.3) are per-base
 * depth by strand.  Here: one dense start/stop difference array per region
 * and strand, turned into depth on first query.                            */
struct CoverageBlock {
    uint start = 0, end = 0;
    std::vector<int> ev[2];     /* len+1 difference events, then depth */
    bool summed = false;
    void RecordCover(uint s, uint e, int dir) {       /* [s,e) clipped to the region */
        if (s < start) s = start;
        if (e > end) e = end;
        if (s >= e) return;
        ev[dir][s - start]++;
        ev[dir][e - start]--;
    }
    void finish() {
        if (summed) return;
        for (int d = 0; d < 2; d++) {
            int run = 0;
            for (size_t i = 0; i < ev[d].size(); i++) { run += ev[d][i]; ev[d][i] = run; }
        }
        summed = true;
    }
    /* plane 0/1/2 */
    void updateCoverageHist(std::map<uint, uint>& hist, uint s, uint e, int plane) {
        finish();
        if (s < start) s = start;
        if (e > end) e = end;
        for (uint x = s; x < e; x++) {
            uint v = plane == 2 ? (uint)ev[0][x - start] + (uint)ev[1][x - start] : (uint)ev[plane][x - start];
            hist[v]++;
        }
    }
};

struct BEDrecord {
    std::string chr, name, strand, score;
    uint start = 0, end = 0;
    std::vector<std::pair<uint, uint>> blocks;   /* absolute [s,e) */
};

/* histogram statistics — ⟨upstream⟩ CoverageBlocks::{coverage,percentile,trimmedMean}FromHist */
static double coverageFromHist(const std::map<uint, uint>& hist) {
    unsigned long n = 0, z = 0;
    for (auto& kv : hist) { n += kv.second; if (kv.first == 0) z += kv.second; }
    if (n == 0) return 0.0;
    return (double)(n - z) / (double)n;
}
static double meanFromHist(const std::map<uint, uint>& hist) {
    unsigned long n = 0; unsigned long long s = 0;
    for (auto& kv : hist) { n += kv.second; s += (unsigned long long)kv.first * kv.second; }
    return n ? (double)s / (double)n : 0.0;
}
/* value of the 1-based rank k in the sorted multiset */
static uint valueAtRank(const std::map<uint, uint>& hist, unsigned long k) {
    unsigned long c = 0;
    uint last = 0;
    for (auto& kv : hist) {
        c += kv.second; last = kv.first;
        if (c >= k) return kv.first;
    }
    return last;
}
static double percentileFromHist(const std::map<uint, uint>& hist, uint pct, bool interpolate) {
    unsigned long n = 0;
    for (auto& kv : hist) n += kv.second;
    if (n == 0) return 0.0;
    unsigned long long num = (unsigned long long)(n + 1) * pct;     /* OQ-7: rank = (N+1)p/100 */
    unsigned long k = (unsigned long)(num / 100);
    uint rem = (uint)(num % 100);
    if (k < 1) return (double)valueAtRank(hist, 1);
    if (k >= n) return (double)valueAtRank(hist, n);
    double vk = (double)valueAtRank(hist, k);
    if (!interpolate || rem == 0) return vk;
    double vk1 = (double)valueAtRank(hist, k + 1);
    return vk + ((double)rem / 100.0) * (vk1 - vk);
}
static double trimmedMeanFromHist(const std::map<uint, uint>& hist, uint centrePct) {
    unsigned long n = 0;
    for (auto& kv : hist) n += kv.second;
    if (n == 0) return 0.0;
    unsigned long skip = (unsigned long)((unsigned long long)n * (100 - centrePct) / 200);   /* OQ-8 */
    unsigned long lo = skip, hi = n - skip;      /* keep 0-based ranks [lo,hi) */
    if (hi <= lo) return 0.0;                    /* centre 0 %: nothing kept */
    unsigned long long sum = 0;
    unsigned long c = 0;
    for (auto& kv : hist) {
        unsigned long a = c, b = c + kv.second;  /* this depth holds ranks [a,b) */
        unsigned long x = std::max(a, lo), y = std::min(b, hi);
        if (y > x) sum += (unsigned long long)kv.first * (y - x);
        c = b;
    }
    return (double)sum / (double)(hi - lo);
}

/* FragmentsMap — named by north_star (SURVEY OQ-14): whole-chromosome
 * per-strand fragment-block depth as difference arrays.                    */
class FragmentsMap : public ReadBlockProcessor {
public:
    std::vector<uint> lens;
    std::vector<std::vector<int>> ev[2];
    bool summed = false;
    bool enabled = false;
    void setLens(const std::vector<uint>& l) { lens = l; }
    void ChrMapUpdate(const std::vector<std::string>& chrmap) override {
        if (!enabled) return;
        for (int d = 0; d < 2; d++) {
            ev[d].assign(chrmap.size(), std::vector<int>());
            for (size_t c = 0; c < chrmap.size(); c++) ev[d][c].assign((size_t)lens.at(c) + 1, 0);
        }
        summed = false;
    }
    void ProcessBlocks(const FragmentBlocks& b) override {
        if (!enabled) return;
        int d = b.direction ? 1 : 0;
        auto& v = ev[d].at(b.chr_id);
        for (int r = 0; r < b.readCount; r++)
            for (size_t j = 0; j < b.rLens[r].size(); j++) {
                size_t s = (size_t)b.readStart[r] + b.rStarts[r][j];
                size_t e = s + b.rLens[r][j];
                if (s >= v.size()) continue;
                if (e >= v.size()) e = v.size() - 1;
                v[s]++; v[e]--;
            }
    }
    void finish() {
        if (summed || !enabled) return;
        for (int d = 0; d < 2; d++)
            for (auto& v : ev[d]) { int run = 0; for (auto& x : v) { run += x; x = run; } }
        summed = true;
    }
    uint depth(int chr, uint x, int plane) {
        finish();
        if (chr < 0 || (size_t)chr >= ev[0].size() || x >= ev[0][chr].size()) return 0;
        return plane == 2 ? (uint)ev[0][chr][x] + (uint)ev[1][chr][x] : (uint)ev[plane][chr][x];
    }
};

/* ====================================================================== */
/* CoverageBlocks / CoverageBlocksIRFinder — ⟨upstream⟩
 * ReadBlockProcessor_CoverageBlocks.cpp                                     */
struct IRRow {
    double coverage = 0, depth = 0, p25 = 0, p50 = 0, p75 = 0, first50 = 0, last50 = 0, irratio = 0;
    uint sp_left = 0, sp_right = 0, splice_left = 0, splice_right = 0, splice_exact = 0;
    uint8_t warning = 0;
};

class CoverageBlocks : public ReadBlockProcessor {
public:
    std::vector<BEDrecord> BEDrecords;                                  /* file order */
    std::map<std::string, std::vector<CoverageBlock>> chrName_CoverageBlocks;   /* sorted, disjoint */
    std::vector<std::vector<CoverageBlock>*> chrID_CoverageBlocks;

    void loadRef(std::istream& in) {
        std::string line;
        std::map<std::string, std::vector<std::pair<uint, uint>>> spans;
        while (std::getline(in, line)) {
            chomp(line);
            if (line.empty() || line[0] == '#') continue;
            auto f = split(line, '\t');
            if (f.size() < 12) continue;
            BEDrecord r;
            r.chr = f[0]; r.start = (uint)std::stoul(f[1]); r.end = (uint)std::stoul(f[2]);
            r.name = f[3]; r.score = f[4]; r.strand = f[5];
            uint nb = (uint)std::stoul(f[9]);
            auto sz = split(f[10], ','), st = split(f[11], ',');
            for (uint i = 0; i < nb; i++) {
                uint s = r.start + (uint)std::stoul(st.at(i));
                uint e = s + (uint)std::stoul(sz.at(i));
                r.blocks.push_back(std::make_pair(s, e));
                if (e > s) spans[r.chr].push_back(std::make_pair(s, e));
            }
            BEDrecords.push_back(r);
        }
        /* regions = union of all measured blocks of a chromosome */
        for (auto& kv : spans) {
            auto& v = kv.second;
            std::sort(v.begin(), v.end());
            std::vector<CoverageBlock>& out = chrName_CoverageBlocks[kv.first];
            for (auto& se : v) {
                if (!out.empty() && se.first <= out.back().end) {
                    if (se.second > out.back().end) out.back().end = se.second;
                } else {
                    CoverageBlock cb; cb.start = se.first; cb.end = se.second;
                    out.push_back(cb);
                }
            }
            for (auto& cb : out) {
                cb.ev[0].assign((size_t)(cb.end - cb.start) + 1, 0);
                cb.ev[1].assign((size_t)(cb.end - cb.start) + 1, 0);
            }
        }
    }
    void ChrMapUpdate(const std::vector<std::string>& chrmap) override {
        chrID_CoverageBlocks.clear();
        for (auto& n : chrmap) {
            auto it = chrName_CoverageBlocks.find(n);
            chrID_CoverageBlocks.push_back(it == chrName_CoverageBlocks.end() ? nullptr : &it->second);
        }
    }
    void ProcessBlocks(const FragmentBlocks& b) override {
        std::vector<CoverageBlock>* v = chrID_CoverageBlocks.at(b.chr_id);
        if (!v) return;
        int d = b.direction ? 1 : 0;
        for (int r = 0; r < b.readCount; r++) {
            for (size_t j = 0; j < b.rLens[r].size(); j++) {
                uint s = b.readStart[r] + b.rStarts[r][j];
                uint e = s + b.rLens[r][j];
                /* first region whose end is beyond s */
                auto it = std::upper_bound(v->begin(), v->end(), s,
                                           [](uint key, const CoverageBlock& cb) { return key < cb.end; });
                for (; it != v->end() && it->start < e; ++it) it->RecordCover(s, e, d);
            }
        }
    }
    void fillHist(std::map<uint, uint>& hist, const std::string& chr, uint s, uint e, int plane) {
        auto c = chrName_CoverageBlocks.find(chr);
        if (c == chrName_CoverageBlocks.end() || s >= e) return;
        auto& v = c->second;
        auto it = std::upper_bound(v.begin(), v.end(), s,
                                   [](uint key, const CoverageBlock& cb) { return key < cb.end; });
        for (; it != v.end() && it->start < e; ++it) it->updateCoverageHist(hist, s, e, plane);
    }
};

class CoverageBlocksIRFinder : public CoverageBlocks {
public:
    irf_params P;
    FragmentsMap* fmap = nullptr;                 /* only for flank_use_measured == 0 */
    const std::vector<std::string>* bamNames = nullptr;

    double flankMedian(const BEDrecord& r, uint lo, uint hi, int plane) {
        if (lo < r.start) lo = r.start;
        if (hi > r.end) hi = r.end;
        std::map<uint, uint> h;
        if (lo < hi) {
            if (P.flank_use_measured) {
                for (auto& b : r.blocks) {
                    uint s = std::max(b.first, lo), e = std::min(b.second, hi);
                    if (s < e) fillHist(h, r.chr, s, e, plane);
                }
            } else {
                int cid = -1;
                if (bamNames)
                    for (size_t i = 0; i < bamNames->size(); i++) if ((*bamNames)[i] == r.chr) cid = (int)i;
                for (uint x = lo; x < hi; x++) h[fmap && cid >= 0 ? fmap->depth(cid, x, plane) : 0]++;
            }
        }
        return percentileFromHist(h, 50, P.pct_interpolate != 0);
    }

    /* directionality 0 => nondir table; ±1 => dir table */
    std::vector<IRRow> computeTable(const JunctionCount& JC, const SpansPoint& SP, int directionality) {
        std::vector<IRRow> rows;
        rows.reserve(BEDrecords.size());
        for (auto& r : BEDrecords) {
            int plane = 2;
            if (directionality != 0) plane = ((r.strand != "-") == (directionality > 0)) ? 1 : 0;
            IRRow o;
            std::map<uint, uint> hist;
            for (auto& b : r.blocks) fillHist(hist, r.chr, b.first, b.second, plane);
            /* measured blocks on a chromosome with no region table: depth 0 everywhere */
            if (chrName_CoverageBlocks.find(r.chr) == chrName_CoverageBlocks.end())
                for (auto& b : r.blocks) if (b.second > b.first) hist[0] += b.second - b.first;
            o.coverage = coverageFromHist(hist);
            o.depth = trimmedMeanFromHist(hist, P.trim_centre_pct);
            o.p25 = percentileFromHist(hist, 25, P.pct_interpolate != 0);
            o.p50 = percentileFromHist(hist, 50, P.pct_interpolate != 0);
            o.p75 = percentileFromHist(hist, 75, P.pct_interpolate != 0);
            o.sp_left = SP.lookup(r.chr, r.start, plane);
            o.sp_right = SP.lookup(r.chr, r.end, plane);
            unsigned long fo = P.flank_offset, fl = P.flank_len;
            unsigned long a0 = (unsigned long)r.start + fo, a1 = a0 + fl;
            long b1 = (long)r.end - (long)fo, b0 = b1 - (long)fl;
            if (a0 > r.end) a0 = r.end;
            if (a1 > r.end) a1 = r.end;
            if (b1 < (long)r.start) b1 = r.start;
            if (b0 < (long)r.start) b0 = r.start;
            o.first50 = flankMedian(r, (uint)a0, (uint)a1, plane);
            o.last50 = flankMedian(r, (uint)b0, (uint)b1, plane);
            o.splice_left = JC.lookupLeft(r.chr, r.start, plane);
            o.splice_right = JC.lookupRight(r.chr, r.end, plane);
            o.splice_exact = JC.lookup(r.chr, r.start, r.end, plane);
            uint smax = std::max(o.splice_left, o.splice_right);
            double den = o.depth + (double)smax;
            o.irratio = den > 0.0 ? o.depth / den : 0.0;
            /* Warnings (OQ-12) */
            double maxflank = std::max(std::max((double)o.sp_left, (double)o.sp_right),
                                       std::max(o.first50, o.last50));
            if ((double)o.splice_exact + o.depth < P.warn_lowcover_min) o.warning = IRF_WARN_LOWCOVER;
            else if ((double)o.splice_exact < P.warn_lowsplice_min) o.warning = IRF_WARN_LOWSPLICING;
            else if ((double)o.splice_exact * P.warn_minor_ratio < (double)smax) o.warning = IRF_WARN_MINORISOFORM;
            else if (P.warn_nonuniform_and ? (maxflank > o.depth + P.warn_nonuniform_abs && maxflank > o.depth * P.warn_nonuniform_rel)
                                           : (maxflank > o.depth * P.warn_nonuniform_rel + P.warn_nonuniform_abs)) o.warning = IRF_WARN_NONUNIFORM;
            else o.warning = IRF_WARN_NONE;
            rows.push_back(o);
        }
        return rows;
    }

    static const char* warnText(uint8_t w) {
        switch (w) {
            case IRF_WARN_LOWCOVER: return "LowCover";
            case IRF_WARN_LOWSPLICING: return "LowSplicing";
            case IRF_WARN_MINORISOFORM: return "MinorIsoform";
            case IRF_WARN_NONUNIFORM: return "NonUniformIntronCover";
            default: return "-";
        }
    }
    void WriteOutput(std::ostream& os, const std::vector<IRRow>& rows) const {
        os << "Chr\tStart\tEnd\tName\tNull\tStrand\tExcludedBases\tCoverage\tIntronDepth\t"
              "IntronDepth25Percentile\tIntronDepth50Percentile\tIntronDepth75Percentile\t"
              "ExonToIntronReadsLeft\tExonToIntronReadsRight\tIntronDepthFirst50bp\tIntronDepthLast50bp\t"
              "SpliceLeft\tSpliceRight\tSpliceExact\tIRratio\tWarnings\n";
        os << std::setprecision((int)P.out_precision);
        for (size_t i = 0; i < BEDrecords.size(); i++) {
            const BEDrecord& r = BEDrecords[i];
            const IRRow& o = rows[i];
            unsigned long meas = 0;
            for (auto& b : r.blocks) meas += b.second - b.first;
            os << r.chr << '\t' << r.start << '\t' << r.end << '\t' << r.name << '\t' << r.score << '\t'
               << r.strand << '\t' << ((unsigned long)(r.end - r.start) - meas) << '\t' << o.coverage << '\t'
               << o.depth << '\t' << o.p25 << '\t' << o.p50 << '\t' << o.p75 << '\t' << o.sp_left << '\t'
               << o.sp_right << '\t' << o.first50 << '\t' << o.last50 << '\t' << o.splice_left << '\t'
               << o.splice_right << '\t' << o.splice_exact << '\t' << o.irratio << '\t' << warnText(o.warning)
               << '\n';
        }
    }
};

/* ====================================================================== */
/* BAM2blocks — ⟨upstream⟩ BAM2blocks.cpp                                   */
struct BamRec {
    int32_t refID = -1, pos = 0;
    uint16_t flag = 0;
    int32_t l_seq = 0;
    std::vector<uint32_t> cigar;
    std::string name;
};

static void cigar2block(const uint32_t* cigar, uint n, std::vector<int>& starts, std::vector<int>& lens,
                        uint& genome_len) {
    starts.clear(); lens.clear();
    uint cur_start = 0, cur_len = 0;
    for (uint i = 0; i < n; i++) {
        uint op = cigar[i] & 0xf, l = cigar[i] >> 4;
        switch (op) {
            case 0: case 2: case 7: case 8:      /* M D = X : extend block */
                cur_len += l; break;
            case 3:                              /* N : close block, open the next */
                if (cur_len > 0) { starts.push_back((int)cur_start); lens.push_back((int)cur_len); }
                cur_start += cur_len + l; cur_len = 0; break;
            default: break;                      /* I S H P : ignored */
        }
    }
    if (cur_len > 0) { starts.push_back((int)cur_start); lens.push_back((int)cur_len); }
    genome_len = cur_start + cur_len;
}

class BAM2blocks {
public:
    irf_params P;
    std::vector<std::function<void(const std::vector<std::string>&)>> callbacksChrMappingChange;
    std::vector<std::function<void(const FragmentBlocks&)>> callbacksProcessBlocks;
    std::vector<std::string> chr_names;
    std::vector<uint> chr_lens;
    irf_reader_stats st;
    const uint8_t* buf = nullptr; size_t len = 0, off = 0;
    FragmentBlocks oBlocks;
    std::string error;

    BAM2blocks() { memset(&st, 0, sizeof st); }
    void registerCallbackChrMappingChange(std::function<void(const std::vector<std::string>&)> f) { callbacksChrMappingChange.push_back(f); }
    void registerCallbackProcessBlocks(std::function<void(const FragmentBlocks&)> f) { callbacksProcessBlocks.push_back(f); }
    void openBuffer(const uint8_t* b, size_t n) { buf = b; len = n; off = 0; }

    bool rd(void* dst, size_t n) {
        if (off + n > len) return false;
        memcpy(dst, buf + off, n); off += n; return true;
    }
    int readBamHeader() {
        char magic[4]; int32_t l_text, n_ref;
        if (!rd(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) { error = "bad BAM magic"; return -1; }
        if (!rd(&l_text, 4) || l_text < 0 || off + (size_t)l_text > len) { error = "bad l_text"; return -1; }
        off += (size_t)l_text;
        if (!rd(&n_ref, 4) || n_ref < 0) { error = "bad n_ref"; return -1; }
        for (int i = 0; i < n_ref; i++) {
            int32_t l_name, l_ref;
            if (!rd(&l_name, 4) || l_name <= 0 || off + (size_t)l_name > len) { error = "bad l_name"; return -1; }
            std::string nm((const char*)buf + off, (size_t)l_name - 1); off += (size_t)l_name;
            if (!rd(&l_ref, 4)) { error = "bad l_ref"; return -1; }
            chr_names.push_back(nm); chr_lens.push_back((uint)l_ref);
        }
        for (auto& cb : callbacksChrMappingChange) cb(chr_names);
        return 0;
    }
    /* 0 = EOF, 1 = record, -1 = malformed */
    int readRecord(BamRec& r) {
        if (off == len) return 0;
        int32_t block_size;
        if (!rd(&block_size, 4) || block_size < 32 || off + (size_t)block_size > len) { error = "truncated record"; return -1; }
        const uint8_t* p = buf + off;
        off += (size_t)block_size;
        memcpy(&r.refID, p, 4); memcpy(&r.pos, p + 4, 4);
        uint8_t l_read_name = p[8];
        uint16_t n_cigar; memcpy(&n_cigar, p + 12, 2);
        memcpy(&r.flag, p + 14, 2); memcpy(&r.l_seq, p + 16, 4);
        if (32 + (size_t)l_read_name + 4 * (size_t)n_cigar > (size_t)block_size) { error = "record fields overrun"; return -1; }
        r.name.assign((const char*)p + 32, l_read_name ? l_read_name - 1 : 0);
        r.cigar.resize(n_cigar);
        if (n_cigar) memcpy(r.cigar.data(), p + 32 + l_read_name, 4 * (size_t)n_cigar);
        st.n_records++; st.n_nucleotides += (uint64_t)(r.l_seq > 0 ? r.l_seq : 0);
        return 1;
    }
    void emit() { for (auto& cb : callbacksProcessBlocks) cb(oBlocks); }

    void processSingle(const BamRec& r) {
        uint glen;
        cigar2block(r.cigar.data(), (uint)r.cigar.size(), oBlocks.rStarts[0], oBlocks.rLens[0], glen);
        if (oBlocks.rLens[0].empty()) { st.n_error++; return; }
        oBlocks.chr_id = r.refID;
        oBlocks.direction = !(r.flag & 0x10);
        oBlocks.readCount = 1;
        oBlocks.readStart[0] = (uint)r.pos; oBlocks.readEnd[0] = (uint)r.pos + glen;
        oBlocks.readStart[1] = oBlocks.readEnd[1] = 0;
        oBlocks.rStarts[1].clear(); oBlocks.rLens[1].clear();
        oBlocks.readName = r.name;
        st.n_singles++;
        emit();
    }
    void processPair(const BamRec& a, const BamRec& b) {    /* a.pos <= b.pos */
        uint len1, len2;
        std::vector<int> s1, l1, s2, l2;
        cigar2block(a.cigar.data(), (uint)a.cigar.size(), s1, l1, len1);
        cigar2block(b.cigar.data(), (uint)b.cigar.size(), s2, l2, len2);
        if (l1.empty() || l2.empty()) { st.n_error += 2; return; }
        uint pos1 = (uint)a.pos, pos2 = (uint)b.pos, e1 = pos1 + len1, e2 = pos2 + len2;
        oBlocks.chr_id = a.refID;
        oBlocks.direction = (a.flag & 0x40) ? !(a.flag & 0x10) : !(a.flag & 0x20);   /* strand of read 1 */
        oBlocks.readName = a.name;
        bool isLong = P.pair_long_inclusive ? (e1 <= pos2) : (e1 < pos2);
        bool isShort = !isLong && (P.pair_short_inclusive ? (e1 >= e2) : (e1 > e2));
        bool merged = false;
        if (isLong) {
            st.n_pairs_long++;
        } else if (isShort) {
            st.n_pairs_short++;
            oBlocks.readCount = 1;
            oBlocks.rStarts[0] = s1; oBlocks.rLens[0] = l1;
            oBlocks.rStarts[1].clear(); oBlocks.rLens[1].clear();
            oBlocks.readStart[0] = pos1; oBlocks.readEnd[0] = e1;
            oBlocks.readStart[1] = oBlocks.readEnd[1] = 0;
            merged = true;
        } else {
            /* Intersect: splice read 2 onto the block of read 1 that holds pos2 */
            int hit = -1;
            for (size_t i = 0; i < l1.size(); i++) {
                uint bs = pos1 + (uint)s1[i], be = bs + (uint)l1[i];
                if (bs <= pos2 && pos2 <= be) { hit = (int)i; break; }
            }
            if (hit >= 0) {
                std::vector<int> ns(s1.begin(), s1.begin() + hit + 1), nl(l1.begin(), l1.begin() + hit + 1);
                nl[hit] = (int)(pos2 - pos1) - s1[hit] + l2[0];
                for (size_t j = 1; j < l2.size(); j++) { ns.push_back(s2[j] + (int)(pos2 - pos1)); nl.push_back(l2[j]); }
                oBlocks.readCount = 1;
                oBlocks.rStarts[0] = ns; oBlocks.rLens[0] = nl;
                oBlocks.rStarts[1].clear(); oBlocks.rLens[1].clear();
                oBlocks.readStart[0] = pos1; oBlocks.readEnd[0] = std::max(e1, e2);
                oBlocks.readStart[1] = oBlocks.readEnd[1] = 0;
                st.n_pairs_intersect++;
                merged = true;
            } else {
                /* OQ-3: pos2 in an N gap of read 1: two reads, originals; the summary class is a switch */
                if (P.gap_pair_is_intersect) st.n_pairs_intersect++; else st.n_pairs_long++;
            }
        }
        if (!merged) {
            oBlocks.readCount = 2;
            oBlocks.rStarts[0] = s1; oBlocks.rLens[0] = l1;
            oBlocks.rStarts[1] = s2; oBlocks.rLens[1] = l2;
            oBlocks.readStart[0] = pos1; oBlocks.readEnd[0] = e1;
            oBlocks.readStart[1] = pos2; oBlocks.readEnd[1] = e2;
        }
        st.n_pairs++;
        emit();
    }
    int processAll() {
        BamRec a, b;
        int n_ref = (int)chr_names.size();
        while (true) {
            int rc = readRecord(a);
            if (rc < 0) return -1;
            if (rc == 0) break;
            if (!(a.flag & 0x1)) {
                if (a.flag & P.skip_flag_mask) { st.n_skipped++; continue; }
                if (a.refID < 0 || a.refID >= n_ref || a.pos < 0) { st.n_error++; continue; }
                processSingle(a);
            } else {
                rc = readRecord(b);
                if (rc < 0) return -1;
                if (rc == 0) { st.n_error++; break; }      /* dangling mate */
                if ((a.flag | b.flag) & P.skip_flag_mask) { st.n_skipped += 2; continue; }
                if (a.refID != b.refID || a.refID < 0 || a.refID >= n_ref || a.pos < 0 || b.pos < 0) { st.n_error += 2; continue; }
                if (a.pos <= b.pos) processPair(a, b); else processPair(b, a);
            }
        }
        return 0;
    }
};

/* ====================================================================== */
/* The assembled engine = ⟨upstream⟩ irfinder.cpp main()                    */
struct Engine {
    irf_params P;
    JunctionCount oJC;
    SpansPoint oSP;
    FragmentsInROI oROI;
    FragmentsInChr oChr;
    CoverageBlocksIRFinder oCB;
    FragmentsMap oFM;
    BAM2blocks BB;
    std::vector<std::string> bamNames;
    std::vector<uint> bamLens;
    bool finalized = false;
    int directionality = 0;
    std::vector<IRRow> rowsNondir, rowsDir;
    uint64_t n_frag = 0, n_blk = 0, n_jev = 0;
    std::string error;

    /* flattened result storage (irf_results view) */
    std::vector<std::string> slotNames;
    std::vector<uint64_t> r_chr_cnt, r_roi_cnt;
    std::vector<uint32_t> r_sp_chr, r_sp_pos, r_sp_cnt;
    std::vector<uint32_t> r_j_chr, r_j_left, r_j_right, r_j_cnt;
    std::vector<uint8_t> r_j_refdir;
    struct Cols {
        std::vector<double> f[8];
        std::vector<uint32_t> u[5];
        std::vector<uint8_t> w;
    } cNondir, cDir;

    void chrMap(const std::vector<std::string>& names) {
        bamNames = names;
        oJC.ChrMapUpdate(names); oSP.ChrMapUpdate(names); oROI.ChrMapUpdate(names);
        oChr.ChrMapUpdate(names); oCB.ChrMapUpdate(names);
        oFM.setLens(bamLens); oFM.ChrMapUpdate(names);
        oCB.bamNames = &bamNames;
    }
    void process(const FragmentBlocks& b) {
        oJC.ProcessBlocks(b); oSP.ProcessBlocks(b); oROI.ProcessBlocks(b);
        oChr.ProcessBlocks(b); oCB.ProcessBlocks(b); oFM.ProcessBlocks(b);
        n_frag++;
        for (int r = 0; r < b.readCount; r++) n_blk += b.rLens[r].size();
    }
    static void fillCols(Cols& c, const std::vector<IRRow>& rows) {
        for (int i = 0; i < 8; i++) c.f[i].clear();
        for (int i = 0; i < 5; i++) c.u[i].clear();
        c.w.clear();
        for (auto& o : rows) {
            c.f[0].push_back(o.coverage); c.f[1].push_back(o.depth); c.f[2].push_back(o.p25);
            c.f[3].push_back(o.p50); c.f[4].push_back(o.p75); c.f[5].push_back(o.first50);
            c.f[6].push_back(o.last50); c.f[7].push_back(o.irratio);
            c.u[0].push_back(o.sp_left); c.u[1].push_back(o.sp_right); c.u[2].push_back(o.splice_left);
            c.u[3].push_back(o.splice_right); c.u[4].push_back(o.splice_exact);
            c.w.push_back(o.warning);
        }
    }
    void finalize() {
        if (finalized) return;
        directionality = oJC.Directional();
        rowsNondir = oCB.computeTable(oJC, oSP, 0);
        if (directionality != 0) rowsDir = oCB.computeTable(oJC, oSP, directionality);
        /* chromosome slots: BAM references, then reference-only names (sorted) */
        slotNames = bamNames;
        std::map<std::string, uint32_t> slot;
        for (size_t i = 0; i < bamNames.size(); i++) if (!slot.count(bamNames[i])) slot[bamNames[i]] = (uint32_t)i;
        std::vector<std::string> extra;
        auto note = [&](const std::string& n) { if (!slot.count(n)) { slot[n] = 0xffffffffu; extra.push_back(n); } };
        for (auto& kv : oJC.chrName_junc) note(kv.first);
        for (auto& kv : oSP.chrName_tab) note(kv.first);
        for (auto& n : oROI.roi_chr) note(n);
        for (auto& r : oCB.BEDrecords) note(r.chr);
        std::sort(extra.begin(), extra.end());
        for (auto& n : extra) { slot[n] = (uint32_t)slotNames.size(); slotNames.push_back(n); }

        r_chr_cnt.clear();
        for (auto& c : oChr.cnt) { r_chr_cnt.push_back(c[0]); r_chr_cnt.push_back(c[1]); }
        size_t nroi = oROI.cnt.size();
        r_roi_cnt.assign(2 * nroi, 0);
        for (size_t i = 0; i < nroi; i++) { r_roi_cnt[i] = oROI.cnt[i][0]; r_roi_cnt[nroi + i] = oROI.cnt[i][1]; }
        /* SpansPoint sorted by (slot,pos) */
        {
            std::vector<std::pair<uint32_t, const SpansPoint::Tab*>> order;
            for (auto& kv : oSP.chrName_tab) order.push_back(std::make_pair(slot[kv.first], &kv.second));
            std::sort(order.begin(), order.end());
            r_sp_chr.clear(); r_sp_pos.clear();
            std::vector<uint32_t> c0, c1;
            for (auto& o : order)
                for (size_t i = 0; i < o.second->pos.size(); i++) {
                    r_sp_chr.push_back(o.first); r_sp_pos.push_back(o.second->pos[i]);
                    c0.push_back(o.second->cnt[0][i]); c1.push_back(o.second->cnt[1][i]);
                }
            r_sp_cnt = c0; r_sp_cnt.insert(r_sp_cnt.end(), c1.begin(), c1.end());
        }
        /* junctions sorted by (slot,left,right) */
        {
            std::vector<std::pair<uint32_t, const JunctionCount::JMap*>> order;
            for (auto& kv : oJC.chrName_junc) order.push_back(std::make_pair(slot[kv.first], &kv.second));
            std::sort(order.begin(), order.end());
            r_j_chr.clear(); r_j_left.clear(); r_j_right.clear(); r_j_refdir.clear();
            std::vector<uint32_t> c0, c1;
            n_jev = 0;
            for (auto& o : order)
                for (auto& j : *o.second) {
                    r_j_chr.push_back(o.first); r_j_left.push_back(j.first.first); r_j_right.push_back(j.first.second);
                    c0.push_back(j.second[0]); c1.push_back(j.second[1]); r_j_refdir.push_back((uint8_t)j.second[2]);
                    n_jev += j.second[0] + j.second[1];
                }
            r_j_cnt = c0; r_j_cnt.insert(r_j_cnt.end(), c1.begin(), c1.end());
        }
        fillCols(cNondir, rowsNondir);
        fillCols(cDir, rowsDir);
        finalized = true;
    }
    static void viewCols(irf_ir_table& t, const Cols& c) {
        t.coverage = c.f[0].data(); t.intron_depth = c.f[1].data(); t.p25 = c.f[2].data(); t.p50 = c.f[3].data();
        t.p75 = c.f[4].data(); t.first50 = c.f[5].data(); t.last50 = c.f[6].data(); t.irratio = c.f[7].data();
        t.sp_left = c.u[0].data(); t.sp_right = c.u[1].data(); t.splice_left = c.u[2].data();
        t.splice_right = c.u[3].data(); t.splice_exact = c.u[4].data(); t.warning = c.w.data();
    }
    int writeOutputs(const std::string& dir) {
        finalize();
        {
            std::ofstream os(dir + "/IRFinder-ROI.txt");
            if (!os) return IRF_E_IO;
            /* name total neg pos, grouped by chromosome name, by end within it */
            for (auto& kv : oROI.chrName_roi)
                for (auto& r : kv.second)
                    os << oROI.roi_names[r.file_idx] << '\t' << (oROI.cnt[r.file_idx][0] + oROI.cnt[r.file_idx][1])
                       << '\t' << oROI.cnt[r.file_idx][0] << '\t' << oROI.cnt[r.file_idx][1] << '\n';
        }
        {
            std::ofstream os(dir + "/IRFinder-ChrCoverage.txt");
            if (!os) return IRF_E_IO;
            for (size_t i = 0; i < oChr.names.size(); i++)
                os << oChr.names[i] << '\t' << (oChr.cnt[i][0] + oChr.cnt[i][1]) << '\t' << oChr.cnt[i][0] << '\t'
                   << oChr.cnt[i][1] << '\n';
        }
        { std::ofstream os(dir + "/IRFinder-JuncCount.txt"); if (!os) return IRF_E_IO; oJC.WriteOutput(os); }
        { std::ofstream os(dir + "/IRFinder-SpansPoint.txt"); if (!os) return IRF_E_IO; oSP.WriteOutput(os); }
        { std::ofstream os(dir + "/IRFinder-IR-nondir.txt"); if (!os) return IRF_E_IO; oCB.WriteOutput(os, rowsNondir); }
        if (directionality != 0) {
            std::ofstream os(dir + "/IRFinder-IR-dir.txt");
            if (!os) return IRF_E_IO;
            oCB.WriteOutput(os, rowsDir);
        }
        return 0;
    }
};

}   // namespace orc

/* ====================================================================== */
/* flat C interface for ctypes (tests / smoke / bench cpu_baseline only)   */
extern "C" {

void orc_params_default(irf_params* p) {
    memset(p, 0, sizeof *p);
    p->struct_size = (uint32_t)sizeof *p;
    p->skip_flag_mask = 0x4;
    p->pair_long_inclusive = 0;
    p->pair_short_inclusive = 1;
    p->junc_min_flank = 5;
    p->sp_overhang_left = 5;
    p->sp_overhang_right = 4;
    p->sp_inclusive = 0;
    p->trim_centre_pct = 40;
    p->pct_interpolate = 1;
    p->flank_offset = 5;
    p->flank_len = 50;
    p->flank_use_measured = 1;
    p->dir_ratio = 10;
    p->out_precision = 6;
    p->warn_lowcover_min = 10.0;
    p->warn_lowsplice_min = 4.0;
    p->warn_minor_ratio = 1.33;
    p->warn_nonuniform_rel = 1.5;
    p->warn_nonuniform_abs = 4.0;
    p->junc_event_capacity = 1ull << 26;
    p->keep_depth = 0;
}

void* orc_create(const irf_params* params, const char* const bufs[4], const size_t lens[4]) {
    orc::Engine* E = new orc::Engine();
    if (params) E->P = *params; else orc_params_default(&E->P);
    E->oJC.min_flank = E->P.junc_min_flank;
    E->oJC.dir_ratio = E->P.dir_ratio;
    E->oSP.setSpanLength(E->P.sp_overhang_left, E->P.sp_overhang_right);
    E->oSP.inclusive = E->P.sp_inclusive != 0;
    E->oCB.P = E->P;
    E->oFM.enabled = E->P.keep_depth != 0 || E->P.flank_use_measured == 0;
    E->oCB.fmap = &E->oFM;
    E->BB.P = E->P;
    try {
        { std::istringstream is(std::string(bufs[0], lens[0])); E->oCB.loadRef(is); }
        { std::istringstream is(std::string(bufs[1], lens[1])); E->oJC.loadRef(is); }
        { std::istringstream is(std::string(bufs[2], lens[2])); E->oSP.loadRef(is); }
        { std::istringstream is(std::string(bufs[3], lens[3])); E->oROI.loadRef(is); }
    } catch (const std::exception& e) {
        delete E;
        return nullptr;
    }
    using namespace std::placeholders;
    E->BB.registerCallbackChrMappingChange([E](const std::vector<std::string>& n) {
        E->bamLens = E->BB.chr_lens;
        E->chrMap(n);
    });
    E->BB.registerCallbackProcessBlocks(std::bind(&orc::Engine::process, E, _1));
    return E;
}
void orc_destroy(void* h) { delete (orc::Engine*)h; }

int orc_chr_map_update(void* h, uint32_t n, const char* const* names, const uint32_t* lens) {
    orc::Engine* E = (orc::Engine*)h;
    std::vector<std::string> v;
    E->bamLens.clear();
    for (uint32_t i = 0; i < n; i++) { v.push_back(names[i]); E->bamLens.push_back(lens ? lens[i] : 0); }
    try { E->chrMap(v); } catch (const std::exception& e) { E->error = e.what(); return IRF_E_ARG; }
    return 0;
}

/* n_frag ProcessBlocks calls per processor, rebuilt from the SoA batch */
int orc_process_batch(void* h, const irf_batch* b) {
    orc::Engine* E = (orc::Engine*)h;
    orc::FragmentBlocks fb;
    try {
        for (uint32_t i = 0; i < b->n_frag; i++) {
            fb.chr_id = (int)b->frag_chr[i];
            fb.direction = (b->frag_flags[i] & IRF_FRAG_DIR) != 0;
            fb.readCount = (b->frag_flags[i] & IRF_FRAG_TWO) ? 2 : 1;
            for (int r = 0; r < 2; r++) { fb.rStarts[r].clear(); fb.rLens[r].clear(); fb.readStart[r] = fb.readEnd[r] = 0; }
            /* absolute block coordinates are what every processor uses; re-express them
             * relative to readStart[0] = frag_start and readStart[1] = first block of read 1 */
            bool seen1 = false;
            fb.readStart[0] = b->frag_start[i];
            for (uint32_t k = b->frag_blk_off[i]; k < b->frag_blk_off[i + 1]; k++) {
                int r = b->blk_read[k] ? 1 : 0;
                if (r == 1 && !seen1) { fb.readStart[1] = b->blk_start[k]; seen1 = true; }
                fb.rStarts[r].push_back((int)(b->blk_start[k] - fb.readStart[r]));
                fb.rLens[r].push_back((int)b->blk_len[k]);
                fb.readEnd[r] = b->blk_start[k] + b->blk_len[k];
            }
            if (fb.readCount == 2) fb.readEnd[1] = b->frag_end[i]; else fb.readEnd[0] = b->frag_end[i];
            E->process(fb);
        }
    } catch (const std::exception& e) { E->error = e.what(); return IRF_E_ARG; }
    return 0;
}

/* whole reader path on an uncompressed BAM stream held in memory */
int orc_process_bam(void* h, const uint8_t* bam, size_t len) {
    orc::Engine* E = (orc::Engine*)h;
    E->BB.openBuffer(bam, len);
    try {
        if (E->BB.readBamHeader() != 0) { E->error = E->BB.error; return IRF_E_PARSE; }
        if (E->BB.processAll() != 0) { E->error = E->BB.error; return IRF_E_PARSE; }
    } catch (const std::exception& e) { E->error = e.what(); return IRF_E_ARG; }
    return 0;
}
int orc_reader_stats(void* h, irf_reader_stats* out) { *out = ((orc::Engine*)h)->BB.st; return 0; }
int orc_bam_header(void* h, const char** names, uint32_t* lens, uint32_t cap) {
    orc::Engine* E = (orc::Engine*)h;
    uint32_t n = (uint32_t)E->BB.chr_names.size();
    for (uint32_t i = 0; i < n && i < cap; i++) { names[i] = E->BB.chr_names[i].c_str(); lens[i] = E->BB.chr_lens[i]; }
    return (int)n;
}

int orc_finalize(void* h) {
    orc::Engine* E = (orc::Engine*)h;
    try { E->finalize(); } catch (const std::exception& e) { E->error = e.what(); return IRF_E_ARG; }
    return 0;
}
int orc_get_results(void* h, irf_results* out) {
    orc::Engine* E = (orc::Engine*)h;
    E->finalize();
    memset(out, 0, sizeof *out);
    out->directionality = E->directionality;
    out->n_chr = (uint32_t)E->bamNames.size(); out->chr_cnt = E->r_chr_cnt.data();
    out->n_roi = (uint32_t)E->oROI.cnt.size(); out->roi_cnt = E->r_roi_cnt.data();
    out->n_sp = (uint32_t)E->r_sp_pos.size(); out->sp_chr = E->r_sp_chr.data(); out->sp_pos = E->r_sp_pos.data(); out->sp_cnt = E->r_sp_cnt.data();
    out->n_junc = E->r_j_left.size(); out->junc_chr = E->r_j_chr.data(); out->junc_left = E->r_j_left.data();
    out->junc_right = E->r_j_right.data(); out->junc_cnt = E->r_j_cnt.data(); out->junc_refdir = E->r_j_refdir.data();
    out->n_intron = (uint32_t)E->oCB.BEDrecords.size();
    out->has_dir = E->directionality != 0;
    orc::Engine::viewCols(out->nondir, E->cNondir);
    orc::Engine::viewCols(out->dir, E->cDir);
    out->n_fragments = E->n_frag; out->n_blocks = E->n_blk; out->n_junction_events = E->n_jev;
    return 0;
}
const char* orc_chr_name(void* h, uint32_t slot) {
    orc::Engine* E = (orc::Engine*)h;
    return slot < E->slotNames.size() ? E->slotNames[slot].c_str() : nullptr;
}
uint32_t orc_n_chr_slots(void* h) { return (uint32_t)((orc::Engine*)h)->slotNames.size(); }
int orc_write_outputs(void* h, const char* outdir) {
    orc::Engine* E = (orc::Engine*)h;
    try { return E->writeOutputs(outdir); } catch (const std::exception& e) { E->error = e.what(); return IRF_E_IO; }
}
int orc_get_depth(void* h, uint32_t chr, uint32_t start, uint32_t end, uint32_t plane, uint32_t* out) {
    orc::Engine* E = (orc::Engine*)h;
    if (!E->oFM.enabled) return IRF_E_STATE;
    for (uint32_t x = start; x < end; x++) out[x - start] = E->oFM.depth((int)chr, x, (int)plane);
    return 0;
}
const char* orc_last_error(void* h) { return ((orc::Engine*)h)->error.c_str(); }

/* ---- pure functions for the known-answer tests ---- */
int orc_cigar2block(const uint32_t* cigar, uint32_t n, uint32_t* starts, uint32_t* lens, uint32_t cap,
                    uint32_t* n_blocks, uint32_t* genome_len) {
    std::vector<int> s, l; orc::uint g;
    orc::cigar2block(cigar, n, s, l, g);
    *n_blocks = (uint32_t)s.size(); *genome_len = g;
    if (s.size() > cap) return IRF_E_RANGE;
    for (size_t i = 0; i < s.size(); i++) { starts[i] = (uint32_t)s[i]; lens[i] = (uint32_t)l[i]; }
    return 0;
}
/* out: coverage, mean, trimmed mean, p25, p50, p75 */
int orc_stats_from_values(const uint32_t* v, uint64_t n, const irf_params* p, double out[6]) {
    std::map<orc::uint, orc::uint> h;
    for (uint64_t i = 0; i < n; i++) h[v[i]]++;
    out[0] = orc::coverageFromHist(h);
    out[1] = orc::meanFromHist(h);
    out[2] = orc::trimmedMeanFromHist(h, p->trim_centre_pct);
    out[3] = orc::percentileFromHist(h, 25, p->pct_interpolate != 0);
    out[4] = orc::percentileFromHist(h, 50, p->pct_interpolate != 0);
    out[5] = orc::percentileFromHist(h, 75, p->pct_interpolate != 0);
    return 0;
}

/* reader only: BAM bytes -> fragments, flattened to an SoA batch held by the
 * returned handle (to compare with the product's reader).                   */
struct orc_frag_dump {
    std::vector<uint32_t> chr, start, end, off, bstart, blen;
    std::vector<uint8_t> flags, bread;
    std::vector<std::string> names; std::vector<uint32_t> lens;
    irf_reader_stats st;
};
void* orc_bam_to_batch(const irf_params* params, const uint8_t* bam, size_t len, irf_batch* out, int* status) {
    orc_frag_dump* D = new orc_frag_dump();
    orc::BAM2blocks BB;
    if (params) BB.P = *params; else orc_params_default(&BB.P);
    D->off.push_back(0);
    BB.registerCallbackProcessBlocks([D](const orc::FragmentBlocks& f) {
        D->chr.push_back((uint32_t)f.chr_id);
        D->flags.push_back((uint8_t)((f.direction ? IRF_FRAG_DIR : 0) | (f.readCount == 2 ? IRF_FRAG_TWO : 0)));
        D->start.push_back(f.readStart[0]);
        uint32_t fe = f.readEnd[0];
        if (f.readCount == 2 && f.readEnd[1] > fe) fe = f.readEnd[1];
        D->end.push_back(fe);
        for (int r = 0; r < f.readCount; r++)
            for (size_t j = 0; j < f.rLens[r].size(); j++) {
                D->bstart.push_back(f.readStart[r] + (uint32_t)f.rStarts[r][j]);
                D->blen.push_back((uint32_t)f.rLens[r][j]);
                D->bread.push_back((uint8_t)r);
            }
        D->off.push_back((uint32_t)D->bstart.size());
    });
    BB.openBuffer(bam, len);
    int rc = 0;
    if (BB.readBamHeader() != 0 || BB.processAll() != 0) rc = IRF_E_PARSE;
    D->st = BB.st; D->names = BB.chr_names; D->lens = BB.chr_lens;
    memset(out, 0, sizeof *out);
    out->n_frag = (uint32_t)D->chr.size(); out->n_blk = (uint32_t)D->bstart.size();
    out->frag_chr = D->chr.data(); out->frag_flags = D->flags.data(); out->frag_start = D->start.data();
    out->frag_end = D->end.data(); out->frag_blk_off = D->off.data(); out->blk_start = D->bstart.data();
    out->blk_len = D->blen.data(); out->blk_read = D->bread.data(); out->mem_kind = IRF_MEM_HOST;
    if (status) *status = rc;
    return D;
}
int orc_dump_stats(void* d, irf_reader_stats* out) { *out = ((orc_frag_dump*)d)->st; return 0; }
void orc_dump_free(void* d) { delete (orc_frag_dump*)d; }

}   // extern "C"

/* ====================================================================== */
#ifdef ORC_MAIN
/* irfinder_oracle OUTDIR ref-cover.bed ref-sj.ref ref-read-continues.ref ref-ROI.bed < raw.bam
 * Same argv shape as the upstream engine.                     */
static std::string slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    std::stringstream ss; ss << f.rdbuf(); return ss.str();
}
int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s OUTDIR cover.bed sj.ref read-continues.ref ROI.bed < raw.bam\n", argv[0]); return 2; }
    std::string r[4] = {slurp(argv[2]), slurp(argv[3]), slurp(argv[4]), slurp(argv[5])};
    const char* bufs[4] = {r[0].data(), r[1].data(), r[2].data(), r[3].data()};
    size_t lens[4] = {r[0].size(), r[1].size(), r[2].size(), r[3].size()};
    irf_params P; orc_params_default(&P);
    void* E = orc_create(&P, bufs, lens);
    if (!E) { fprintf(stderr, "reference load failed\n"); return 1; }
    std::string bam((std::istreambuf_iterator<char>(std::cin)), std::istreambuf_iterator<char>());
    if (orc_process_bam(E, (const uint8_t*)bam.data(), bam.size()) != 0) { fprintf(stderr, "BAM: %s\n", orc_last_error(E)); return 1; }
    orc_finalize(E);
    int rc = orc_write_outputs(E, argv[1]);
    irf_reader_stats st; orc_reader_stats(E, &st);
    printf("Total reads processed\t%llu\nTotal nucleotides\t%llu\nTotal singles processed\t%llu\nTotal pairs processed\t%llu\n"
           "Short pairs\t%llu\nIntersect pairs\t%llu\nLong pairs\t%llu\nSkipped reads\t%llu\nError reads\t%llu\n",
           (unsigned long long)st.n_records, (unsigned long long)st.n_nucleotides, (unsigned long long)st.n_singles,
           (unsigned long long)st.n_pairs, (unsigned long long)st.n_pairs_short, (unsigned long long)st.n_pairs_intersect,
           (unsigned long long)st.n_pairs_long, (unsigned long long)st.n_skipped, (unsigned long long)st.n_error);
    orc_destroy(E);
    return rc;
}
#endif
