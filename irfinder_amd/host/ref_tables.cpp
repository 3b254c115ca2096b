// ref_tables.cpp — see ref_tables.h
#include "ref_tables.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>

namespace irf {

uint32_t RefTables::chr_id(const std::string& name) {
    auto it = chr_index.find(name);
    if (it != chr_index.end()) return it->second;
    uint32_t id = (uint32_t)chr_names.size();
    chr_names.push_back(name);
    chr_index.emplace(name, id);
    return id;
}

namespace {

// A line split into tab-separated fields, as (pointer,length) views into the buffer.
struct Fields {
    const char* p[16];
    size_t n[16];
    int count = 0;
};

// Iterates lines of buf; skips blank lines and '#' comments; strips '\r'.
template <class F>
bool for_each_line(const char* buf, size_t len, int min_fields, const char* what, std::string& err, F&& fn) {
    size_t pos = 0, line_no = 0;
    while (pos < len) {
        size_t eol = pos;
        while (eol < len && buf[eol] != '\n') eol++;
        size_t end = eol;
        if (end > pos && buf[end - 1] == '\r') end--;
        line_no++;
        if (end > pos && buf[pos] != '#') {
            Fields f;
            size_t a = pos;
            for (size_t i = pos; i <= end; i++) {
                if (i == end || buf[i] == '\t') {
                    if (f.count < 16) { f.p[f.count] = buf + a; f.n[f.count] = i - a; f.count++; }
                    a = i + 1;
                }
            }
            if (f.count < min_fields) {
                char m[160];
                snprintf(m, sizeof m, "%s line %zu: expected >= %d tab-separated fields, got %d", what, line_no, min_fields, f.count);
                err = m;
                return false;
            }
            if (!fn(f, line_no)) {
                if (err.empty()) {
                    char m[160];
                    snprintf(m, sizeof m, "%s line %zu: malformed number", what, line_no);
                    err = m;
                }
                return false;
            }
        }
        pos = eol + 1;
    }
    return true;
}

bool to_u32(const char* p, size_t n, uint32_t& out) {
    if (n == 0 || n > 10) return false;
    uint64_t v = 0;
    for (size_t i = 0; i < n; i++) {
        if (p[i] < '0' || p[i] > '9') return false;
        v = v * 10 + (uint64_t)(p[i] - '0');
    }
    if (v > 0xffffffffull) return false;
    out = (uint32_t)v;
    return true;
}

// comma-separated list of unsigned ints (optional trailing comma)
bool to_list(const char* p, size_t n, std::vector<uint32_t>& out) {
    out.clear();
    size_t a = 0;
    for (size_t i = 0; i <= n; i++) {
        if (i == n || p[i] == ',') {
            if (i > a) {
                uint32_t v;
                if (!to_u32(p + a, i - a, v)) return false;
                out.push_back(v);
            }
            a = i + 1;
        }
    }
    return true;
}

}   // namespace

bool parse_cover_bed(const char* buf, size_t len, RefTables& t, std::string& err) {
    if (t.in_blk_off.empty()) t.in_blk_off.push_back(0);
    std::vector<uint32_t> sizes, starts;
    return for_each_line(buf, len, 12, "ref-cover.bed", err, [&](const Fields& f, size_t line_no) {
        uint32_t s, e, nb;
        if (!to_u32(f.p[1], f.n[1], s) || !to_u32(f.p[2], f.n[2], e) || !to_u32(f.p[9], f.n[9], nb)) return false;
        if (!to_list(f.p[10], f.n[10], sizes) || !to_list(f.p[11], f.n[11], starts)) return false;
        if (e < s || sizes.size() < nb || starts.size() < nb) {
            char m[160];
            snprintf(m, sizeof m, "ref-cover.bed line %zu: inconsistent block list", line_no);
            err = m;
            return false;
        }
        t.in_chr.push_back(t.chr_id(std::string(f.p[0], f.n[0])));
        t.in_start.push_back(s);
        t.in_end.push_back(e);
        t.in_name.emplace_back(f.p[3], f.n[3]);
        t.in_score.emplace_back(f.p[4], f.n[4]);
        t.in_strand.push_back(f.n[5] ? f.p[5][0] : '.');
        uint64_t meas = 0;
        for (uint32_t k = 0; k < nb; k++) {
            uint32_t bs = s + starts[k], be = bs + sizes[k];
            t.blk_start.push_back(bs);
            t.blk_end.push_back(be);
            meas += sizes[k];
        }
        t.in_measured.push_back(meas);
        t.in_blk_off.push_back((uint32_t)t.blk_start.size());
        return true;
    });
}

bool parse_sj_ref(const char* buf, size_t len, RefTables& t, std::string& err) {
    return for_each_line(buf, len, 4, "ref-sj.ref", err, [&](const Fields& f, size_t) {
        uint32_t s, e;
        if (!to_u32(f.p[1], f.n[1], s) || !to_u32(f.p[2], f.n[2], e)) return false;
        t.kj_chr.push_back(t.chr_id(std::string(f.p[0], f.n[0])));
        t.kj_left.push_back(s);
        t.kj_right.push_back(e);
        uint8_t d = 0;
        if (f.n[3] == 1 && f.p[3][0] == '+') d = 2;
        else if (f.n[3] == 1 && f.p[3][0] == '-') d = 1;
        t.kj_dir.push_back(d);
        return true;
    });
}

bool parse_read_continues_ref(const char* buf, size_t len, RefTables& t, std::string& err) {
    return for_each_line(buf, len, 2, "ref-read-continues.ref", err, [&](const Fields& f, size_t) {
        uint32_t p;
        if (!to_u32(f.p[1], f.n[1], p)) return false;
        t.sp_chr.push_back(t.chr_id(std::string(f.p[0], f.n[0])));
        t.sp_pos.push_back(p);
        return true;
    });
}

bool parse_roi_bed(const char* buf, size_t len, RefTables& t, std::string& err) {
    return for_each_line(buf, len, 3, "ref-ROI.bed", err, [&](const Fields& f, size_t) {
        uint32_t s, e;
        if (!to_u32(f.p[1], f.n[1], s) || !to_u32(f.p[2], f.n[2], e)) return false;
        t.roi_chr.push_back(t.chr_id(std::string(f.p[0], f.n[0])));
        t.roi_start.push_back(s);
        t.roi_end.push_back(e);
        t.roi_name.emplace_back(f.count > 3 ? std::string(f.p[3], f.n[3]) : std::string("."));
        return true;
    });
}

void finish_tables(RefTables& t) {
    {   // junctions: sort by (chr,left,right), merge duplicates by OR-ing the strand bits
        size_t n = t.kj_left.size();
        std::vector<uint32_t> idx(n);
        std::iota(idx.begin(), idx.end(), 0u);
        std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
            if (t.kj_chr[a] != t.kj_chr[b]) return t.kj_chr[a] < t.kj_chr[b];
            if (t.kj_left[a] != t.kj_left[b]) return t.kj_left[a] < t.kj_left[b];
            return t.kj_right[a] < t.kj_right[b];
        });
        std::vector<uint32_t> c, l, r;
        std::vector<uint8_t> d;
        for (uint32_t i : idx) {
            if (!c.empty() && c.back() == t.kj_chr[i] && l.back() == t.kj_left[i] && r.back() == t.kj_right[i]) {
                d.back() |= t.kj_dir[i];
            } else {
                c.push_back(t.kj_chr[i]); l.push_back(t.kj_left[i]); r.push_back(t.kj_right[i]); d.push_back(t.kj_dir[i]);
            }
        }
        t.kj_chr.swap(c); t.kj_left.swap(l); t.kj_right.swap(r); t.kj_dir.swap(d);
    }
    {   // points: sorted unique (chr,pos)
        size_t n = t.sp_pos.size();
        std::vector<uint64_t> key(n);
        for (size_t i = 0; i < n; i++) key[i] = ((uint64_t)t.sp_chr[i] << 32) | t.sp_pos[i];
        std::sort(key.begin(), key.end());
        key.erase(std::unique(key.begin(), key.end()), key.end());
        t.sp_chr.resize(key.size());
        t.sp_pos.resize(key.size());
        for (size_t i = 0; i < key.size(); i++) { t.sp_chr[i] = (uint32_t)(key[i] >> 32); t.sp_pos[i] = (uint32_t)key[i]; }
    }
    if (t.in_blk_off.empty()) t.in_blk_off.push_back(0);
}

bool read_file(const char* path, std::string& out, std::string& err) {
    FILE* f = fopen(path, "rb");
    if (!f) { err = std::string("cannot open ") + path; return false; }
    char buf[1 << 16];
    size_t n;
    out.clear();
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
    fclose(f);
    return true;
}

}   // namespace irf
