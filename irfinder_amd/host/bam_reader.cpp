// bam_reader.cpp — see bam_reader.h
#include "bam_reader.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace irf {

namespace {
inline uint32_t le32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline uint16_t le16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
const uint32_t kMaxBlockSize = 1u << 28;
}   // namespace

// ⟨upstream⟩ BAM2blocks::cigar2block.  M,D,=,X extend the open block; N closes
// it and moves the origin; I,S,H,P are ignored.
uint32_t cigar_to_blocks(uint32_t pos, const uint32_t* cigar, uint32_t n_cigar, std::vector<BlockSpan>& out) {
    out.clear();
    uint32_t origin = pos, run = 0;
    for (uint32_t i = 0; i < n_cigar; i++) {
        const uint32_t op = cigar[i] & 0xfu, l = cigar[i] >> 4;
        if (op == 0 || op == 2 || op == 7 || op == 8) {
            run += l;
        } else if (op == 3) {
            if (run) out.push_back(BlockSpan{origin, run});
            origin += run + l;
            run = 0;
        }
    }
    if (run) out.push_back(BlockSpan{origin, run});
    return origin + run;
}

WorkerPool::WorkerPool(int n_threads) {
    for (int i = 1; i < n_threads; i++) threads_.emplace_back(&WorkerPool::worker_main, this, i);
}

WorkerPool::~WorkerPool() {
    {
        std::lock_guard<std::mutex> g(m_);
        stop_ = true;
    }
    wake_.notify_all();
    for (auto& t : threads_) t.join();
}

void WorkerPool::drain(int id) {
    for (size_t i = next_.fetch_add(1); i < n_items_; i = next_.fetch_add(1)) (*fn_)(id, i);
}

void WorkerPool::worker_main(int id) {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> lk(m_);
    while (true) {
        wake_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
        lk.unlock();
        drain(id);
        lk.lock();
        if (--busy_ == 0) done_.notify_one();
    }
}

void WorkerPool::start(size_t first, size_t end, const std::function<void(int, size_t)>& fn) {
    {
        std::lock_guard<std::mutex> g(m_);
        fn_ = &fn;
        n_items_ = end;
        next_.store(first);
        busy_ = (int)threads_.size();
        generation_++;
    }
    wake_.notify_all();
}

void WorkerPool::wait() {
    drain(0);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return busy_ == 0; });
    fn_ = nullptr;
}

static void* new_zstream() {
    z_stream* zs = new z_stream();
    memset(zs, 0, sizeof *zs);
    inflateInit2(zs, -15);
    return zs;
}

BamReader::BamReader(const irf_params& p) : P(p) {
    memset(&stats, 0, sizeof stats);
    zs_.push_back(new_zstream());
}

BamReader::~BamReader() {
    delete pool_;
    for (void* z : zs_) {
        inflateEnd((z_stream*)z);
        delete (z_stream*)z;
    }
}

void BamReader::set_threads(int n) {
    if (n < 1) n = 1;
    if (n > 64) n = 64;
    if (pool_ && pool_->size() == n) return;
    delete pool_;
    pool_ = n > 1 ? new WorkerPool(n) : nullptr;
    while ((int)zs_.size() < n) zs_.push_back(new_zstream());
}

int BamReader::feed(const uint8_t* data, size_t len, bool bgzf) {
    if (!error.empty()) return IRF_E_PARSE;
    if (bgzf) {
        if (zoff_ > 0 && zoff_ == zbuf_.size()) { zbuf_.clear(); zoff_ = 0; }
        zbuf_.insert(zbuf_.end(), data, data + len);
        int rc = inflate_blocks();
        if (rc != 0) return rc;
    } else {
        if (roff_ > 0 && roff_ == raw_.size()) { raw_.clear(); roff_ = 0; }
        raw_.insert(raw_.end(), data, data + len);
    }
    return parse_raw(raw_.size(), true);
}

// BGZF (SAM spec §4.1): gzip members with a 'BC' extra subfield carrying BSIZE.  The complete members in the buffer are
// located first (their inflated sizes are in the trailers, so every member knows where its output goes), then
// inflated and CRC-checked independently — by the worker pool when there is one.
int bgzf_next_member(const uint8_t* buf, size_t size, size_t scan, size_t total_out, ZBlock& b, size_t& member_bytes, const char** what) {
    const size_t avail = size - scan;
    if (avail < 18) return 0;
    const uint8_t* h = buf + scan;
    if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) { *what = "BGZF: bad member header"; return -1; }
    const uint32_t xlen = le16(h + 10);
    if (avail < 12 + (size_t)xlen) return 0;
    uint32_t bsize = 0;
    bool found = false;
    for (uint32_t x = 0; x + 4 <= xlen;) {
        const uint8_t* sf = h + 12 + x;
        const uint32_t slen = le16(sf + 2);
        if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) { bsize = le16(sf + 4); found = true; }
        x += 4 + slen;
    }
    if (!found) { *what = "BGZF: missing BC subfield"; return -1; }
    const size_t total = (size_t)bsize + 1;
    if (total < 12 + (size_t)xlen + 8) { *what = "BGZF: bad BSIZE"; return -1; }
    if (avail < total) return 0;
    b.in_off = scan + 12 + xlen;
    b.clen = total - 12 - xlen - 8;
    b.crc = le32(h + total - 8);
    b.isize = le32(h + total - 4);
    if (b.isize > 65536) { *what = "BGZF: ISIZE > 64 KiB"; return -1; }
    b.out_off = total_out;
    member_bytes = total;
    return 1;
}

std::string bgzf_locate(const uint8_t* buf, size_t size, size_t& scan, size_t& total_out, std::vector<ZBlock>& out) {
    while (true) {
        ZBlock b;
        size_t total = 0;
        const char* what = "";
        const int rc = bgzf_next_member(buf, size, scan, total_out, b, total, &what);
        if (rc < 0) return what;
        if (rc == 0) break;
        total_out += b.isize;
        out.push_back(b);
        scan += total;
    }
    return std::string();
}

int BamReader::inflate_blocks() {
    zblocks_.clear();
    size_t scan = zoff_, total_out = 0;
    const std::string scan_error = bgzf_locate(zbuf_.data(), zbuf_.size(), scan, total_out, zblocks_);
    // (the members before a malformed one are still inflated, as when reading them one by one; the error is sticky)
    if (roff_ > (1u << 20) && roff_ * 2 > raw_.size()) { raw_.erase(raw_.begin(), raw_.begin() + (ptrdiff_t)roff_); roff_ = 0; }
    const size_t at = raw_.size();
    raw_.resize(at + total_out);
    std::atomic<size_t> first_bad(zblocks_.size());
    std::vector<int> why(zblocks_.size(), 0);
    const uint8_t* in = zbuf_.data();
    uint8_t* out = raw_.data() + at;
    // (a std::function object of its own: the pool keeps a pointer to it between start() and wait())
    const std::function<void(int, size_t)> one = [&](int worker, size_t i) {
        const ZBlock& b = zblocks_[i];
        if (!b.isize) return;
        z_stream* zs = (z_stream*)zs_[(size_t)worker];
        inflateReset(zs);
        zs->next_in = const_cast<Bytef*>(in + b.in_off);
        zs->avail_in = (uInt)b.clen;
        zs->next_out = out + b.out_off;
        zs->avail_out = b.isize;
        int zr = inflate(zs, Z_FINISH);
        int bad = 0;
        if (zr != Z_STREAM_END || zs->avail_out != 0) bad = 1;
        else if ((uint32_t)crc32(0L, out + b.out_off, b.isize) != b.crc) bad = 2;
        if (bad) {
            why[i] = bad;
            size_t cur = first_bad.load();
            while (i < cur && !first_bad.compare_exchange_weak(cur, i)) {}
        }
    };
    int parse_rc = 0;
    if (pool_ && zblocks_.size() >= 16) {
        // four waves: while the workers inflate wave w + 1, this thread parses what wave w produced
        const size_t n = zblocks_.size(), step = (n + 3) / 4;
        pool_->run(step, one);
        for (size_t w0 = step; w0 < n && parse_rc == 0; w0 += step) {
            const size_t w1 = w0 + step < n ? w0 + step : n;
            pool_->start(w0, w1, one);
            if (first_bad.load() == n) parse_rc = parse_raw(at + zblocks_[w0].out_off, false);
            pool_->wait();
        }
    } else if (pool_ && zblocks_.size() >= 4) {
        pool_->run(zblocks_.size(), one);
    } else {
        for (size_t i = 0; i < zblocks_.size() && first_bad.load() == zblocks_.size(); i++) one(0, i);
    }
    if (parse_rc != 0) return parse_rc;
    const size_t fb = first_bad.load();
    if (fb < zblocks_.size()) {
        // Everything in front of the corrupt member is parsed, whatever the thread count (the four-wave path may or may not
        // have parsed it already), so batch and counters after the error do not depend on irf_reader_set_threads.  They
        // are still not what a caller should keep: the input is damaged (include/irf.h: irf_reader_feed).
        raw_.resize(at + zblocks_[fb].out_off);
        (void)parse_raw(raw_.size(), false);
        error = why[fb] == 2 ? "BGZF: CRC mismatch" : "BGZF: inflate failed";
        return IRF_E_PARSE;
    }
    zoff_ = scan;
    if (!scan_error.empty()) { error = scan_error; return IRF_E_PARSE; }
    if (zoff_ > (1u << 20) && zoff_ * 2 > zbuf_.size()) { zbuf_.erase(zbuf_.begin(), zbuf_.begin() + (ptrdiff_t)zoff_); zoff_ = 0; }
    return 0;
}

// ⟨upstream⟩ BAM2blocks::readBamHeader
int parse_bam_header(const uint8_t* p, size_t avail, std::vector<std::string>& names_out, std::vector<uint32_t>& lens_out,
                     size_t& consumed, std::string& error) {
    size_t o = 0;
    if (avail < 12) return 1;
    if (memcmp(p, "BAM\1", 4) != 0) { error = "BAM: bad magic"; return IRF_E_PARSE; }
    int32_t l_text = (int32_t)le32(p + 4);
    if (l_text < 0) { error = "BAM: negative l_text"; return IRF_E_PARSE; }
    o = 8 + (size_t)l_text;
    if (avail < o + 4) return 1;
    int32_t n_ref = (int32_t)le32(p + o);
    if (n_ref < 0) { error = "BAM: negative n_ref"; return IRF_E_PARSE; }
    o += 4;
    std::vector<std::string> names;
    std::vector<uint32_t> lens;
    for (int32_t i = 0; i < n_ref; i++) {
        if (avail < o + 4) return 1;
        int32_t l_name = (int32_t)le32(p + o);
        if (l_name <= 0 || l_name > (1 << 20)) { error = "BAM: bad l_name"; return IRF_E_PARSE; }
        o += 4;
        if (avail < o + (size_t)l_name + 4) return 1;
        names.emplace_back((const char*)p + o, (size_t)l_name - 1);
        o += (size_t)l_name;
        lens.push_back(le32(p + o));
        o += 4;
    }
    names_out.swap(names);
    lens_out.swap(lens);
    consumed = o;
    return 0;
}

int BamReader::parse_header(bool& need_more, size_t limit) {
    need_more = true;
    size_t consumed = 0;
    const int rc = parse_bam_header(raw_.data() + roff_, limit - roff_, chr_names, chr_lens, consumed, error);
    if (rc < 0) return rc;
    if (rc == 1) return 0;
    chr_name_ptrs.clear();
    for (auto& s : chr_names) chr_name_ptrs.push_back(s.c_str());
    roff_ += consumed;
    header_done_ = true;
    need_more = false;
    return 0;
}

// ⟨upstream⟩ BAM2blocks::processAll — record loop with adjacent-mate pairing
int BamReader::parse_raw(size_t limit, bool may_compact) {
    // only raw_[.. limit) is final (the worker threads may still be inflating behind it; then raw_ must not move)
    if (!header_done_) {
        bool need_more;
        int rc = parse_header(need_more, limit);
        if (rc != 0) return rc;
        if (need_more) return 0;
    }
    while (true) {
        size_t avail = limit - roff_;
        if (avail < 4) break;
        const uint8_t* p = raw_.data() + roff_;
        uint32_t block_size = le32(p);
        if (block_size < 32 || block_size > kMaxBlockSize) { error = "BAM: bad block_size"; return IRF_E_PARSE; }
        if (avail < 4 + (size_t)block_size) break;
        bool bad = false;
        on_record(p + 4, block_size, bad);
        if (bad) return IRF_E_PARSE;
        roff_ += 4 + (size_t)block_size;
    }
    if (may_compact && roff_ > (1u << 22) && roff_ * 2 > raw_.size()) { raw_.erase(raw_.begin(), raw_.begin() + (ptrdiff_t)roff_); roff_ = 0; }
    return 0;
}

void BamReader::on_record(const uint8_t* p, uint32_t block_size, bool& bad) {
    Rec& r = cur_;
    r.ref_id = (int32_t)le32(p);
    r.pos = (int32_t)le32(p + 4);
    uint32_t l_read_name = p[8];
    uint32_t n_cigar = le16(p + 12);
    r.flag = le16(p + 14);
    int32_t l_seq = (int32_t)le32(p + 16);
    if (32 + (size_t)l_read_name + 4 * (size_t)n_cigar > block_size) { error = "BAM: record fields overrun block_size"; bad = true; return; }
    r.cigar.resize(n_cigar);
    if (n_cigar) memcpy(r.cigar.data(), p + 32 + l_read_name, 4 * (size_t)n_cigar);
    stats.n_records++;
    if (l_seq > 0) stats.n_nucleotides += (uint64_t)l_seq;

    const int32_t n_ref = (int32_t)chr_names.size();
    if (P.mate_join) {
        if (r.flag & 0x1) { on_paired_by_name(p + 32, l_read_name); return; }
    } else {
        if (have_pending_) {
            have_pending_ = false;
            pair_or_drop(pending_, r);
            return;
        }
        if (r.flag & 0x1) {           // paired: the NEXT record is taken as the mate
            std::swap(pending_, cur_);
            have_pending_ = true;
            return;
        }
    }
    if (r.flag & P.skip_flag_mask) { stats.n_skipped++; return; }
    if (r.ref_id < 0 || r.ref_id >= n_ref || r.pos < 0) { stats.n_error++; return; }
    emit_single(r);
}

// the two records of a pair, in arrival order: upstream's checks, then processPair on (left, right)
void BamReader::pair_or_drop(const Rec& a, const Rec& b) {
    const int32_t n_ref = (int32_t)chr_names.size();
    if ((a.flag | b.flag) & P.skip_flag_mask) { stats.n_skipped += 2; return; }
    if (a.ref_id != b.ref_id || a.ref_id < 0 || a.ref_id >= n_ref || a.pos < 0 || b.pos < 0) { stats.n_error += 2; return; }
    if (a.pos <= b.pos) emit_pair(a, b); else emit_pair(b, a);
}

// params.mate_join (upstream pairs ADJACENT records and therefore needs name-collated input; north_star
// feeds coordinate-sorted records).  A paired record waits under its read name until the record of the other mate
// (flag 0x40 / 0x80) arrives; the pair then goes through exactly the checks and the assembly of the adjacent case, in
// arrival order, so a sorted file yields the same multiset of fragments as its collated form.  A second record of the
// SAME mate under a name (secondary / supplementary lines) displaces the waiting one, which is counted as an error
// read, as is every record still waiting at the end of input.
void BamReader::on_paired_by_name(const uint8_t* name, uint32_t l_read_name) {
    std::string key((const char*)name, l_read_name ? l_read_name - 1 : 0);
    auto it = waiting_.find(key);
    if (it == waiting_.end()) { waiting_.emplace(std::move(key), cur_); return; }
    if (((it->second.flag ^ cur_.flag) & 0xc0) == 0) { stats.n_error++; it->second = cur_; return; }
    pair_or_drop(it->second, cur_);
    waiting_.erase(it);
}

int BamReader::finish() {
    if (!error.empty()) return IRF_E_PARSE;
    if (zoff_ != zbuf_.size()) { error = "BGZF: truncated block at end of input"; return IRF_E_PARSE; }
    if (!header_done_) { error = "BAM: truncated header"; return IRF_E_PARSE; }
    if (roff_ != raw_.size()) { error = "BAM: truncated record at end of input"; return IRF_E_PARSE; }
    if (have_pending_) { have_pending_ = false; stats.n_error++; }
    stats.n_error += waiting_.size();
    waiting_.clear();
    return 0;
}

// ⟨upstream⟩ BAM2blocks::processSingle
void BamReader::emit_single(const Rec& r) {
    uint32_t end = cigar_to_blocks((uint32_t)r.pos, r.cigar.data(), (uint32_t)r.cigar.size(), b1_);
    if (b1_.empty()) { stats.n_error++; return; }
    batch.add((uint32_t)r.ref_id, !(r.flag & 0x10), (uint32_t)r.pos, end, b1_.data(), b1_.size(), nullptr, 0, false);
    stats.n_singles++;
}

// ⟨upstream⟩ BAM2blocks::processPair.  a is the left mate (a.pos <= b.pos).
void BamReader::emit_pair(const Rec& a, const Rec& b) {
    const uint32_t pos1 = (uint32_t)a.pos, pos2 = (uint32_t)b.pos;
    const uint32_t e1 = cigar_to_blocks(pos1, a.cigar.data(), (uint32_t)a.cigar.size(), b1_);
    const uint32_t e2 = cigar_to_blocks(pos2, b.cigar.data(), (uint32_t)b.cigar.size(), b2_);
    if (b1_.empty() || b2_.empty()) { stats.n_error += 2; return; }
    // strand of read 1, read off the left record
    const bool dir = (a.flag & 0x40) ? !(a.flag & 0x10) : !(a.flag & 0x20);
    const uint32_t chr = (uint32_t)a.ref_id;
    const bool is_long = P.pair_long_inclusive ? (e1 <= pos2) : (e1 < pos2);
    stats.n_pairs++;
    bool gap_pair = false;
    if (!is_long) {
        const bool is_short = P.pair_short_inclusive ? (e1 >= e2) : (e1 > e2);
        if (is_short) {                      // read 2 inside read 1: read-1 blocks only
            stats.n_pairs_short++;
            batch.add(chr, dir, pos1, e1, b1_.data(), b1_.size(), nullptr, 0, false);
            return;
        }
        // Intersect: extend the read-1 block that holds pos2 by read 2's first block, then append the rest of read 2
        for (size_t i = 0; i < b1_.size(); i++) {
            const uint32_t bs = b1_[i].start, be = bs + b1_[i].len;
            if (bs <= pos2 && pos2 <= be) {
                merged_.assign(b1_.begin(), b1_.begin() + (ptrdiff_t)i + 1);
                merged_[i].len = pos2 - bs + b2_[0].len;
                merged_.insert(merged_.end(), b2_.begin() + 1, b2_.end());
                stats.n_pairs_intersect++;
                batch.add(chr, dir, pos1, std::max(e1, e2), merged_.data(), merged_.size(), nullptr, 0, false);
                return;
            }
        }
        // pos2 falls in an N gap of read 1 (OQ-3): handled as two separate reads; its class in the summary is a switch
        gap_pair = true;
    }
    if (gap_pair && P.gap_pair_is_intersect) stats.n_pairs_intersect++; else stats.n_pairs_long++;
    batch.add(chr, dir, pos1, std::max(e1, e2), b1_.data(), b1_.size(), b2_.data(), b2_.size(), true);
}

}   // namespace irf
