// dreader.hip — irf_dreader: the device reader (SURVEY §8 f N1) and its C ABI (include/irf.h, "Device reader").
//
// Stands where ⟨upstream⟩ bin/IRFinder pipes `gzip -cd sample.bam` into BAM2blocks::processAll: the host only locates
// the BGZF members (one header hop per <= 64 KiB) and parses the BAM header; inflate, CRC check, record location, mate
// pairing and fragment assembly run on the device and leave a DEVICE-resident irf_batch that goes straight into
// irf_submit_batch — the records never exist in host memory.  Semantics = the host reader's (host/bam_reader.cpp),
// params.mate_join = 0 (upstream's pairing: the next record is the mate).
//
// Per feed: [carry of the previous feed | inflated members] in one buffer; what is left behind the last complete
// fragment unit (a partial record, or a first mate whose mate has not arrived) is the next feed's carry.
#include <hip/hip_runtime.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/irf.h"
#include "../host/bam_reader.h"
#include "dreader_kernels.h"
#include "kernels.h"

namespace {

namespace dev = irf::dev;

template <class T>
struct DBuf {
    T* p = nullptr;
    size_t cap = 0;
    ~DBuf() { if (p) (void)hipFree(p); }
    // grows geometrically (never shrinks); contents are NOT preserved
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = std::max<size_t>(n + n / 4, 256);
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
};

// The compressed input of a feed: what the previous feeds left over (a partial member), then the caller's bytes —
// never copied into one host buffer; only what a feed leaves unconsumed is kept.
struct TwoSpan {
    const uint8_t* a; size_t na;
    const uint8_t* b; size_t nb;
    size_t size() const { return na + nb; }
    // [off, off + n) as contiguous bytes: inside a, inside b, or copied into tmp where it straddles the seam
    const uint8_t* view(size_t off, size_t n, std::vector<uint8_t>& tmp) const {
        if (off + n <= na) return a + off;
        if (off >= na) return b + (off - na);
        tmp.resize(n);
        memcpy(tmp.data(), a + off, na - off);
        memcpy(tmp.data() + (na - off), b, n - (na - off));
        return tmp.data();
    }
};

// IRF_DREADER_TRACE=1: wall-clock marks of the host side on stderr (diagnostic)
struct Trace {
    bool on = getenv("IRF_DREADER_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char* what) const {
        if (on) fprintf(stderr, "[dreader %9.3f ms] %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
    }
};
const Trace g_trace;

constexpr uint32_t kVerifyBatch = 8;         // chain_verify launches per host synchronisation (after the first three)
constexpr uint64_t kMinGap = 1ull << 20;    // bytes kept free in front of a slot's inflated members for the predecessor's carry

inline uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8; }
inline uint32_t rd32(const uint8_t* p) { return rd16(p) | rd16(p + 2) << 16; }

// host/bam_reader.cpp: bgzf_locate, over a TwoSpan (same checks, same messages)
std::string locate_members(const TwoSpan& in, size_t& scan, size_t& total_out, std::vector<irf::ZBlock>& out) {
    std::vector<uint8_t> tmp;
    const size_t size = in.size();
    while (true) {
        const size_t avail = size - scan;
        if (avail < 18) break;
        const uint8_t* h = in.view(scan, 12, tmp);
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return "BGZF: bad member header";
        const uint32_t xlen = rd16(h + 10);
        if (avail < 12 + (size_t)xlen) break;
        const uint8_t* x0 = in.view(scan + 12, xlen, tmp);
        uint32_t bsize = 0;
        bool found = false;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const uint8_t* sf = x0 + x;
            const uint32_t slen = rd16(sf + 2);
            if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) { bsize = rd16(sf + 4); found = true; }
            x += 4 + slen;
        }
        if (!found) return "BGZF: missing BC subfield";
        const size_t total = (size_t)bsize + 1;
        if (total < 12 + (size_t)xlen + 8) return "BGZF: bad BSIZE";
        if (avail < total) break;
        const uint8_t* t = in.view(scan + total - 8, 8, tmp);
        irf::ZBlock b;
        b.in_off = scan + 12 + xlen;
        b.clen = total - 12 - xlen - 8;
        b.crc = rd32(t);
        b.isize = rd32(t + 4);
        if (b.isize > 65536) return "BGZF: ISIZE > 64 KiB";
        b.out_off = total_out;
        total_out += b.isize;
        out.push_back(b);
        scan += total;
    }
    return std::string();
}

struct OutSlot {
    DBuf<uint32_t> frag_chr, frag_start, frag_end, frag_blk_off, blk_start, blk_len;
    DBuf<uint8_t> frag_flags, blk_read;
    uint32_t n_frag = 0, n_blk = 0;
};

}   // namespace

struct irf_dreader {
    irf_params P;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool failed = false;
    int failed_code = IRF_E_PARSE;

    // host side
    std::vector<uint8_t> pending;            // compressed bytes not consumed yet (a partial member, or an incomplete header)
    bool header_done = false;
    bool header_skipped = false;             // the device has stepped over the header bytes
    uint64_t header_len = 0;
    std::vector<std::string> chr_names;
    std::vector<uint32_t> chr_lens;
    std::vector<const char*> chr_name_ptrs;
    std::vector<irf::ZBlock> zblocks;
    std::vector<dev::BgzfMember> members;
    irf_reader_stats stats;
    irf_dreader_timing timing;

    // device side.  A feed is two stages: stage 1 (compressed bytes to the device on the copy stream, inflate + CRC on
    // the inflate stream) needs nothing from the feeds before it; stage 2 (record chain, pairing, assembly on `stream`)
    // needs the carry of its predecessor.  With the pipeline on, stage 1 of feed k runs before stage 2 of feed k - 1,
    // on a second host thread, so the copy of feed k + 1 overlaps the inflate of feed k and the parsing of feed k - 1.  The inflated members of a slot start at a fixed
    // offset `gap`; the carry of the predecessor is placed right in front of them.
    struct InSlot {
        DBuf<uint8_t> d_in, raw;
        DBuf<dev::BgzfMember> d_members;
        DBuf<uint32_t> d_status, d_first_bad;
        uint32_t n_members = 0;
        uint64_t total_out = 0, gap = 0;
        size_t consumed_in = 0;
        hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // copy begin / end, inflate begin / end, CRC end
    } slot[2];
    int next_slot = 0, inflight = -1;        // inflight: slot whose stage 2 is still to run
    bool pipeline = false, fed = false;
    hipStream_t s_copy = nullptr, s_inf = nullptr;
    DBuf<uint8_t> d_carry;                   // what the last stage 2 left unconsumed
    uint64_t carry_len = 0;
    bool carry_dangling = false;             // the carry starts with a complete first mate that waits for its mate
    bool carry_partial = false;              // ... and / or ends with an incomplete record
    DBuf<uint32_t> d_count, d_seg_base, d_rec_nf, d_tile_last, d_tile_carry, d_fr, d_bl, d_ws;
    // params.mate_join: name-joined pairing (bamparse.hip: join_*).  The records still waiting for their mate travel in
    // the carry, in front of the partial record (if any): they are the first n_old records of the next feed.
    DBuf<unsigned long long> d_jkeys[2];
    DBuf<uint32_t> d_jvals[2], d_mate, d_jsizes, d_joffs, d_ws_sort;
    uint32_t n_old = 0;
    DBuf<uint64_t> d_guess, d_exit, d_rec_off;
    DBuf<unsigned long long> d_scal;         // [0] record error, [1 .. 1 + RS_COUNT) reader statistics
    DBuf<uint32_t> d_flags;                  // [0] dangling, [8 .. 8 + kVerifyBatch) chain changed, one word per verify launch of a batch
    OutSlot out[2];
    int out_cur = 0;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};      // stage 2: begin, chain done, assembled

    std::mutex err_mutex;                    // stage 1 and stage 2 of neighbouring feeds run on two host threads
    int fail(int code, const std::string& msg) {
        std::lock_guard<std::mutex> g(err_mutex);
        if (!failed) { err = msg; failed_code = code; }      // the first error stays, text and code: later calls return it
        failed = true;
        return code;
    }
    int init();
    int feed(const uint8_t* data, size_t len);
    int stage1(InSlot& S, const TwoSpan& in, size_t consumed_in, uint64_t gap);
    int stage2(InSlot& S);
    int finish();
    int host_header(const TwoSpan& in, size_t n_members_avail);
};

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(IRF_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define MEM_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(IRF_E_NOMEM, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int irf_dreader::init() {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&s_copy, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&s_inf, hipStreamNonBlocking));
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    for (auto& S : slot) {
        for (auto& e : S.ev) HIP_TRY(hipEventCreate(&e));
        MEM_TRY(S.d_first_bad.reserve(4));
    }
    memset(&stats, 0, sizeof stats);
    memset(&timing, 0, sizeof timing);
    MEM_TRY(d_scal.reserve(1 + dev::RS_COUNT));
    MEM_TRY(d_flags.reserve(8 + kVerifyBatch));
    return 0;
}

// The BAM header is needed on the host (irf_chr_map_update wants names and lengths before any batch): the leading
// members are inflated with zlib until it is complete.  The device inflates them again with the rest and starts at
// header_len.
int irf_dreader::host_header(const TwoSpan& in, size_t n_avail) {
    std::vector<uint8_t> raw, tmp;
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) return fail(IRF_E_NOMEM, "zlib: inflateInit2 failed");
    int rc = 1;
    for (size_t i = 0; i < n_avail && rc == 1; i++) {
        const irf::ZBlock& b = zblocks[i];
        if (b.isize) {
            const size_t at = raw.size();
            raw.resize(at + b.isize);
            inflateReset(&zs);
            zs.next_in = const_cast<Bytef*>(in.view(b.in_off, b.clen, tmp));
            zs.avail_in = (uInt)b.clen;
            zs.next_out = raw.data() + at;
            zs.avail_out = b.isize;
            const int zr = inflate(&zs, Z_FINISH);
            if (zr != Z_STREAM_END || zs.avail_out != 0) { inflateEnd(&zs); return fail(IRF_E_PARSE, "BGZF: inflate failed"); }
        }
        size_t consumed = 0;
        std::string e;
        rc = irf::parse_bam_header(raw.data(), raw.size(), chr_names, chr_lens, consumed, e);
        if (rc < 0) { inflateEnd(&zs); return fail(IRF_E_PARSE, e); }
        if (rc == 0) header_len = consumed;
    }
    inflateEnd(&zs);
    if (rc == 0) {
        header_done = true;
        chr_name_ptrs.clear();
        for (auto& s : chr_names) chr_name_ptrs.push_back(s.c_str());
    }
    return 0;
}

int irf_dreader::feed(const uint8_t* data, size_t len) {
    if (failed) return failed_code;
    HIP_TRY(hipSetDevice(device));
    fed = true;
    const TwoSpan in{pending.data(), pending.size(), data, len};
    zblocks.clear();
    size_t scan = 0, total_out = 0;
    g_trace.mark("feed: begin");
    const std::string scan_error = locate_members(in, scan, total_out, zblocks);
    g_trace.mark("feed: members located");
    if (!scan_error.empty()) return fail(IRF_E_PARSE, scan_error);
    out[out_cur ^ 1].n_frag = out[out_cur ^ 1].n_blk = 0;          // the batch of this feed: empty unless a stage 2 fills it
    bool consume = !zblocks.empty();
    if (consume && !header_done) {
        int rc = host_header(in, zblocks.size());
        if (rc != 0) return rc;
        consume = header_done;                            // else: wait for more members (nothing consumed)
    }
    if (!consume) {
        pending.insert(pending.end(), data, data + len);
        out_cur ^= 1;
        return 0;
    }
    const int cur = next_slot;
    next_slot ^= 1;
    // room for the predecessor's carry in front of the inflated members: the carry is known when the pipeline is off;
    // with the pipeline on it is the one before it (stage 2 moves the members in the rare case that the carry does not fit)
    const uint64_t gap = std::max<uint64_t>(kMinGap, (carry_len + 4095) & ~4095ull);
    int rc = 0;
    if (pipeline && inflight >= 0) {
        // the copy of this feed (the calling thread is busy with it for its whole duration when the bytes are in pageable
        // memory) next to the parsing of the feed before it
        int rc1 = 0;
        std::thread copier;
        try {
            copier = std::thread([&] {
                rc1 = hipSetDevice(device) == hipSuccess ? stage1(slot[cur], in, scan, gap) : fail(IRF_E_DEVICE, "hipSetDevice");
            });
        } catch (...) {
            // no second thread to be had: one after the other (nothing may be thrown across the C ABI)
        }
        const int prev = inflight;
        rc = stage2(slot[prev]);
        if (copier.joinable()) copier.join();
        else rc1 = stage1(slot[cur], in, scan, gap);
        inflight = cur;
        if (rc == 0) rc = rc1;
        if (rc != 0) return rc;
    } else {
        rc = stage1(slot[cur], in, scan, gap);
        if (rc != 0) return rc;
        if (pipeline) inflight = cur;
        else if ((rc = stage2(slot[cur])) != 0) return rc;
    }
    // what is left behind the last complete member (stage 1 has read the caller's bytes)
    if (scan >= in.na) {
        pending.assign(data + (scan - in.na), data + len);
    } else {
        pending.erase(pending.begin(), pending.begin() + (ptrdiff_t)scan);
        pending.insert(pending.end(), data, data + len);
    }
    out_cur ^= 1;
    g_trace.mark("feed: end");
    return 0;
}

// Stage 1: compressed bytes + member table to the device, inflate, CRC.  Returns when the caller's bytes have been read;
// the kernels may still be running.
int irf_dreader::stage1(InSlot& S, const TwoSpan& in, size_t consumed_in, uint64_t gap) {
    const uint32_t n_members = (uint32_t)zblocks.size();
    members.resize(n_members);
    uint64_t total_out = 0;
    for (uint32_t i = 0; i < n_members; i++) {
        const irf::ZBlock& b = zblocks[i];
        members[i] = dev::BgzfMember{b.in_off, gap + b.out_off, (uint32_t)b.clen, b.isize, b.crc, 0};
        total_out += b.isize;
    }
    MEM_TRY(S.d_in.reserve(consumed_in + dev::kInflateSlack));
    MEM_TRY(S.d_members.reserve(n_members));
    MEM_TRY(S.d_status.reserve(n_members));
    MEM_TRY(S.raw.reserve(gap + total_out + 64));
    S.n_members = n_members; S.total_out = total_out; S.gap = gap; S.consumed_in = consumed_in;
    g_trace.mark("stage1: buffers reserved");

    HIP_TRY(hipEventRecord(S.ev[0], s_copy));
    {
        const size_t from_a = std::min(in.na, consumed_in);
        if (from_a) HIP_TRY(hipMemcpyAsync(S.d_in.p, in.a, from_a, hipMemcpyHostToDevice, s_copy));
        if (consumed_in > from_a) HIP_TRY(hipMemcpyAsync(S.d_in.p + from_a, in.b, consumed_in - from_a, hipMemcpyHostToDevice, s_copy));
    }
    HIP_TRY(hipMemsetAsync(S.d_in.p + consumed_in, 0, dev::kInflateSlack, s_copy));
    HIP_TRY(hipMemcpyAsync(S.d_members.p, members.data(), sizeof(dev::BgzfMember) * n_members, hipMemcpyHostToDevice, s_copy));
    HIP_TRY(hipMemsetAsync(S.d_first_bad.p, 0xff, 4, s_copy));
    HIP_TRY(hipEventRecord(S.ev[1], s_copy));
    HIP_TRY(hipStreamWaitEvent(s_inf, S.ev[1], 0));
    HIP_TRY(hipEventRecord(S.ev[2], s_inf));
    dev::launch_bgzf_inflate(S.d_in.p, S.d_members.p, n_members, S.raw.p, S.d_status.p, s_inf);
    HIP_TRY(hipEventRecord(S.ev[3], s_inf));
    dev::launch_bgzf_crc(S.raw.p, S.d_members.p, n_members, S.d_status.p, s_inf);
    dev::launch_first_bad(S.d_status.p, n_members, S.d_first_bad.p, s_inf);
    HIP_TRY(hipEventRecord(S.ev[4], s_inf));
    HIP_TRY(hipGetLastError());
    // the caller's bytes (and `members`) may change once this returns, whatever kind of host memory they are in
    g_trace.mark("stage1: launched");
    HIP_TRY(hipEventSynchronize(S.ev[1]));
    g_trace.mark("stage1: copied");
    return 0;
}

// Stage 2: [carry | inflated members] -> fragments of out[out_cur ^ 1], new carry.  Synchronous.
int irf_dreader::stage2(InSlot& S) {
    const uint32_t n_members = S.n_members;
    uint32_t* const f_changed = d_flags.p + 8, *const f_dangling = d_flags.p;
    g_trace.mark("stage2: begin");
    HIP_TRY(hipStreamWaitEvent(stream, S.ev[4], 0));
    uint32_t first_bad = 0;
    HIP_TRY(hipMemcpyAsync(&first_bad, S.d_first_bad.p, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    g_trace.mark("stage2: inflated");
#ifdef IRF_INFLATE_EXP
    { float ms_ = 0; (void)hipEventElapsedTime(&ms_, S.ev[2], S.ev[3]); fprintf(stderr, "[inflate_exp %d] %u members, inflate %.3f ms\n", IRF_INFLATE_EXP, n_members, ms_); }
#endif
    if (first_bad != 0xffffffffu) {
        uint32_t why = 0;
        HIP_TRY(hipMemcpy(&why, S.d_status.p + first_bad, 4, hipMemcpyDeviceToHost));
        return fail(IRF_E_PARSE, why == dev::BGZF_BAD_CRC ? "BGZF: CRC mismatch" : "BGZF: inflate failed");
    }
    if (carry_len > S.gap) {
        // (pipeline only) the carry is longer than the room left for it: move the members behind a larger gap
        const uint64_t gap = (carry_len + 4095) & ~4095ull;
        DBuf<uint8_t> bigger;
        MEM_TRY(bigger.reserve(gap + S.total_out + 64));
        if (S.total_out) HIP_TRY(hipMemcpyAsync(bigger.p + gap, S.raw.p + S.gap, S.total_out, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        std::swap(S.raw.p, bigger.p);
        std::swap(S.raw.cap, bigger.cap);
        S.gap = gap;
    }
    uint8_t* const raw = S.raw.p;
    HIP_TRY(hipEventRecord(ev[0], stream));
    if (carry_len) HIP_TRY(hipMemcpyAsync(raw + (S.gap - carry_len), d_carry.p, carry_len, hipMemcpyDeviceToDevice, stream));
    const uint64_t total = S.gap + S.total_out;

    // ---- record chain ----
    const uint64_t start = header_skipped ? S.gap - carry_len : S.gap + header_len;
    header_skipped = true;
    const int32_t n_ref = (int32_t)chr_names.size();
    uint32_t n_rec = 0;
    uint64_t tail_off = start;
    if (total < start) return fail(IRF_E_PARSE, "BAM: truncated header");
    const uint64_t n_seg64 = (total - start + dev::kChainSegment - 1) / dev::kChainSegment;
    if (n_seg64 >= 0xffffffffull) return fail(IRF_E_RANGE, "feed too large for one pass: feed the file in pieces");
    const uint32_t n_seg = (uint32_t)n_seg64;
    dev::ChainBufs C;
    memset(&C, 0, sizeof C);
    if (n_seg) {
        MEM_TRY(d_guess.reserve(n_seg));
        MEM_TRY(d_exit.reserve(n_seg));
        MEM_TRY(d_count.reserve((size_t)n_seg + 4));
        MEM_TRY(d_seg_base.reserve((size_t)n_seg + 4));
        MEM_TRY(d_ws.reserve(dev::scan_workspace_elems((uint64_t)n_seg + 1)));
        C.buf = raw; C.start = start; C.total = total; C.n_seg = n_seg; C.n_ref = n_ref;
        C.guess = d_guess.p; C.exit_ = d_exit.p; C.count = d_count.p; C.changed = f_changed;
        HIP_TRY(hipMemsetAsync(d_count.p + n_seg, 0, 4, stream));
        dev::launch_chain_guess(C, stream);
        // Verification rounds until one changes nothing.  A round is one launch; the host looks at the flags of a BATCH
        // of launches at a time (3 first: real data settles in 2-3 rounds = one synchronisation; then kVerifyBatch), so
        // input whose payload keeps forming plausible but wrong record chains costs launches, not round trips.
        uint64_t rounds = 0;
        for (bool settled = false; !settled;) {
            const uint32_t nb = rounds == 0 ? 3u : kVerifyBatch;
            uint32_t changed[kVerifyBatch];
            HIP_TRY(hipMemsetAsync(f_changed, 0, 4 * nb, stream));
            for (uint32_t k = 0; k < nb; k++) { C.changed = f_changed + k; dev::launch_chain_verify(C, stream); }
            HIP_TRY(hipMemcpyAsync(changed, f_changed, 4 * nb, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            for (uint32_t k = 0; k < nb && !settled; k++) { rounds++; settled = changed[k] == 0u; }
            if (!settled && rounds > (uint64_t)n_seg + 2 + kVerifyBatch) return fail(IRF_E_DEVICE, "record chain did not settle (internal error)");
        }
        if (rounds > timing.chain_rounds) timing.chain_rounds = (uint32_t)rounds;
        if (g_trace.on && rounds > 4) fprintf(stderr, "[dreader] record chain needed %llu verification rounds over %u segments\n", (unsigned long long)rounds, n_seg);
        dev::launch_exclusive_scan_u32(d_count.p, d_seg_base.p, (uint64_t)n_seg + 1, d_ws.p, stream);
        HIP_TRY(hipMemcpyAsync(&n_rec, d_seg_base.p + n_seg, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(&tail_off, d_exit.p + (n_seg - 1), 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    HIP_TRY(hipEventRecord(ev[1], stream));
    g_trace.mark("stage2: chain done");

    // ---- records -> fragments ----
    OutSlot& O = out[out_cur ^ 1];
    uint32_t n_frag = 0, n_blk = 0, dangling = 0;
    unsigned long long scal[1 + dev::RS_COUNT];
    memset(scal, 0, sizeof scal);
    scal[0] = ~0ull;
    HIP_TRY(hipMemcpyAsync(d_scal.p, scal, sizeof scal, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemsetAsync(f_dangling, 0, 4, stream));
    dev::AssembleArgs A;
    memset(&A, 0, sizeof A);
    dev::JoinArgs J;
    memset(&J, 0, sizeof J);
    uint32_t collision = 0;
    if (n_rec) {
        g_trace.mark("stage2: reserving record index");
        MEM_TRY(d_rec_off.reserve(n_rec));
        MEM_TRY(d_rec_nf.reserve(n_rec));
        MEM_TRY(d_fr.reserve((size_t)n_rec + 4));
        MEM_TRY(d_bl.reserve((size_t)n_rec + 4));
        MEM_TRY(d_tile_last.reserve(dev::pair_tiles(n_rec)));
        MEM_TRY(d_tile_carry.reserve(dev::pair_tiles(n_rec)));
        MEM_TRY(d_ws.reserve(dev::scan_workspace_elems((uint64_t)n_rec + 1)));
        g_trace.mark("stage2: record index reserved");
        dev::RecordIndex R{d_rec_off.p, d_rec_nf.p, d_scal.p};
        dev::launch_chain_emit(C, d_seg_base.p, R, stream);
        if (P.mate_join) {
            // pairs by read name: hash, stable sort of (hash, record index), one walk per group of equal hashes
            MEM_TRY(d_jkeys[0].reserve(n_rec)); MEM_TRY(d_jkeys[1].reserve(n_rec));
            MEM_TRY(d_jvals[0].reserve(n_rec)); MEM_TRY(d_jvals[1].reserve(n_rec));
            MEM_TRY(d_mate.reserve(n_rec)); MEM_TRY(d_jsizes.reserve((size_t)n_rec + 4)); MEM_TRY(d_joffs.reserve((size_t)n_rec + 4));
            MEM_TRY(d_ws_sort.reserve(std::max(dev::radix_workspace_elems(n_rec), dev::scan_workspace_elems((uint64_t)n_rec + 1))));
            J.buf = raw; J.n_rec = n_rec; J.n_old = std::min(n_old, n_rec); J.R = R;
            J.keys = d_jkeys[0].p; J.vals = d_jvals[0].p; J.mate = d_mate.p; J.stats = d_scal.p + 1; J.collision = f_dangling + 1; J.n_waiting = f_dangling + 2; J.sizes = d_jsizes.p;
            HIP_TRY(hipMemsetAsync(f_dangling + 1, 0, 8, stream));
            dev::launch_join_keys(J, stream);
            int from = 0;
            for (int shift = 0; shift < 64; shift += 8) {
                dev::launch_radix_pass((const uint64_t*)d_jkeys[from].p, (uint64_t*)d_jkeys[from ^ 1].p, d_jvals[from].p, d_jvals[from ^ 1].p, n_rec, shift,
                                       d_ws_sort.p, stream);
                from ^= 1;
            }
            J.keys = d_jkeys[from].p; J.vals = d_jvals[from].p;
            dev::launch_join_groups(J, stream);
            A.mate = d_mate.p;
        } else {
            dev::launch_pair_state(d_rec_nf.p, n_rec, d_tile_last.p, d_tile_carry.p, stream);
        }
        A.buf = raw; A.n_rec = n_rec; A.n_ref = n_ref; A.R = R; A.tile_carry = d_tile_carry.p;
        A.skip_flag_mask = P.skip_flag_mask; A.pair_long_inclusive = P.pair_long_inclusive; A.pair_short_inclusive = P.pair_short_inclusive; A.gap_pair_is_intersect = P.gap_pair_is_intersect;
        A.fr_cnt = d_fr.p; A.bl_cnt = d_bl.p; A.stats = d_scal.p + 1; A.dangling = f_dangling;
        HIP_TRY(hipMemsetAsync(d_fr.p + n_rec, 0, 4, stream));
        HIP_TRY(hipMemsetAsync(d_bl.p + n_rec, 0, 4, stream));
        dev::launch_assemble_count(A, stream);
        dev::launch_exclusive_scan_u32(d_fr.p, d_fr.p, (uint64_t)n_rec + 1, d_ws.p, stream);
        dev::launch_exclusive_scan_u32(d_bl.p, d_bl.p, (uint64_t)n_rec + 1, d_ws.p, stream);
        HIP_TRY(hipMemcpyAsync(&n_frag, d_fr.p + n_rec, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(&n_blk, d_bl.p + n_rec, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(&dangling, f_dangling, 4, hipMemcpyDeviceToHost, stream));
        if (P.mate_join) HIP_TRY(hipMemcpyAsync(&collision, f_dangling + 1, 4, hipMemcpyDeviceToHost, stream));
    } else if (n_seg) {
        // no complete record: a broken chain in the first segment still has to be reported
        dev::RecordIndex R{nullptr, nullptr, d_scal.p};
        dev::launch_chain_emit(C, d_seg_base.p, R, stream);
    }
    HIP_TRY(hipMemcpyAsync(scal, d_scal.p, sizeof scal, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    g_trace.mark("stage2: counted");
    if (scal[0] != ~0ull) return fail(IRF_E_PARSE, (scal[0] & 3ull) == 1ull ? "BAM: bad block_size" : "BAM: record fields overrun block_size");
    if (tail_off == dev::kNoRecord) return fail(IRF_E_DEVICE, "record chain broken without a bad record (internal error)");
    if (collision) return fail(IRF_E_RANGE, "two different read names share a 64-bit hash: read this file with the host reader (irf_reader_*)");

    MEM_TRY(O.frag_chr.reserve(n_frag + 4)); MEM_TRY(O.frag_flags.reserve(n_frag + 4)); MEM_TRY(O.frag_start.reserve(n_frag + 4));
    MEM_TRY(O.frag_end.reserve(n_frag + 4)); MEM_TRY(O.frag_blk_off.reserve((size_t)n_frag + 5));
    MEM_TRY(O.blk_start.reserve(n_blk + 4)); MEM_TRY(O.blk_len.reserve(n_blk + 4)); MEM_TRY(O.blk_read.reserve(n_blk + 4));
    g_trace.mark("stage2: out reserved");
    A.frag_chr = O.frag_chr.p; A.frag_flags = O.frag_flags.p; A.frag_start = O.frag_start.p; A.frag_end = O.frag_end.p;
    A.frag_blk_off = O.frag_blk_off.p; A.blk_start = O.blk_start.p; A.blk_len = O.blk_len.p; A.blk_read = O.blk_read.p;
    dev::launch_assemble_write(A, n_frag, n_blk, stream);
    HIP_TRY(hipEventRecord(ev[2], stream));
    O.n_frag = n_frag; O.n_blk = n_blk;

    // ---- carry: everything behind the last complete unit ----
    if (P.mate_join) {
        // the records that still wait for their mate (whole, in arrival order), then the incomplete record at the end
        uint32_t wait[2] = {0, 0};           // bytes, records
        if (n_rec) {
            dev::launch_join_sizes(J, stream);
            dev::launch_exclusive_scan_u32(d_jsizes.p, d_joffs.p, (uint64_t)n_rec + 1, d_ws_sort.p, stream);
            HIP_TRY(hipMemcpyAsync(&wait[0], d_joffs.p + n_rec, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(&wait[1], f_dangling + 2, 4, hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(hipStreamSynchronize(stream));
        g_trace.mark("stage2: written");
        const uint64_t tail_bytes = total - tail_off;
        const uint64_t new_carry = (uint64_t)wait[0] + tail_bytes;
        if (new_carry) {
            // (the carry buffer is a second copy: the next feed's stage 1 may already be writing the other slot's `raw`)
            MEM_TRY(d_carry.reserve(std::max<uint64_t>(new_carry + 64, kMinGap)));
            if (wait[0]) {
                dev::launch_join_carry(J, d_joffs.p, d_carry.p, stream);
                HIP_TRY(hipGetLastError());
            }
            if (tail_bytes) HIP_TRY(hipMemcpyAsync(d_carry.p + wait[0], raw + tail_off, tail_bytes, hipMemcpyDeviceToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        carry_len = new_carry;
        carry_dangling = false;
        carry_partial = tail_bytes != 0;
        n_old = wait[1];           // the next feed walks them again, as its first records
    } else {
    uint64_t tail_from = tail_off;
    if (dangling) HIP_TRY(hipMemcpyAsync(&tail_from, d_rec_off.p + (n_rec - 1), 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    g_trace.mark("stage2: written");
    const uint64_t new_carry = total - tail_from;
    if (new_carry) {
        // (never less than the gap: growing frees the old buffer, and hipFree waits for the next feed's inflate kernel)
        MEM_TRY(d_carry.reserve(std::max<uint64_t>(new_carry + 64, kMinGap)));
        HIP_TRY(hipMemcpyAsync(d_carry.p, raw + tail_from, new_carry, hipMemcpyDeviceToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    carry_len = new_carry;
    carry_dangling = dangling != 0;
    carry_partial = tail_off != total;
    }
    g_trace.mark("stage2: done");

    stats.n_records += scal[1 + dev::RS_RECORDS]; stats.n_nucleotides += scal[1 + dev::RS_NUCLEOTIDES];
    stats.n_singles += scal[1 + dev::RS_SINGLES]; stats.n_pairs += scal[1 + dev::RS_PAIRS];
    stats.n_pairs_short += scal[1 + dev::RS_SHORT]; stats.n_pairs_intersect += scal[1 + dev::RS_INTERSECT];
    stats.n_pairs_long += scal[1 + dev::RS_LONG]; stats.n_skipped += scal[1 + dev::RS_SKIPPED]; stats.n_error += scal[1 + dev::RS_ERROR];

    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, S.ev[0], S.ev[1])); timing.h2d_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, S.ev[2], S.ev[3])); timing.inflate_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, S.ev[3], S.ev[4])); timing.crc_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, ev[0], ev[1])); timing.chain_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, ev[1], ev[2])); timing.assemble_ms += ms;
    timing.bytes_in += S.consumed_in;
    timing.bytes_inflated += S.total_out;
    timing.n_members += n_members;
    return 0;
}

int irf_dreader::finish() {
    if (failed) return failed_code;
    if (inflight >= 0) {
        // pipeline: the last feed's stage 2; its fragments are the batch after this call
        HIP_TRY(hipSetDevice(device));
        out[out_cur ^ 1].n_frag = out[out_cur ^ 1].n_blk = 0;
        const int prev = inflight;
        inflight = -1;
        const int rc = stage2(slot[prev]);
        if (rc != 0) return rc;
        out_cur ^= 1;
    }
    if (!pending.empty()) return fail(IRF_E_PARSE, "BGZF: truncated block at end of input");
    if (!header_done) return fail(IRF_E_PARSE, "BAM: truncated header");
    if (carry_partial) return fail(IRF_E_PARSE, "BAM: truncated record at end of input");
    if (P.mate_join && n_old) {
        // records whose mate never arrived: error reads (host reader: every record still waiting at the end of input)
        stats.n_error += n_old;
        n_old = 0;
        carry_len = 0;
    }
    if (carry_dangling) {
        // the first mate is a record like any other: count it, and the read as an error (as the host reader does)
        int32_t l_seq = 0;
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipMemcpy(&l_seq, d_carry.p + 20, 4, hipMemcpyDeviceToHost));
        stats.n_records++;
        if (l_seq > 0) stats.n_nucleotides += (uint64_t)l_seq;
        stats.n_error++;
        carry_dangling = false;
        carry_len = 0;
    }
    return 0;
}

// ---- irf_gpu_bgzf_inflate_noalloc: device scratch that exists without any allocation call ----------------------------
// A file-scope device array: every GPU the library runs on holds its own copy (9 MiB) until the process ends.  Layout:
// [compressed bytes + kInflateSlack | member table | status words + first-bad word].
namespace {
constexpr int kNaMaxDevices = 64;
constexpr size_t kNaInOff = 0;
constexpr size_t kNaMemOff = ((size_t)IRF_NOALLOC_MAX_BYTES + dev::kInflateSlack + 255) / 256 * 256;
constexpr size_t kNaStatusOff = kNaMemOff + (size_t)IRF_NOALLOC_MAX_MEMBERS * sizeof(dev::BgzfMember);
constexpr size_t kNaBytes = (kNaStatusOff + ((size_t)IRF_NOALLOC_MAX_MEMBERS + 1) * sizeof(uint32_t) + 255) / 256 * 256;
static_assert(kNaStatusOff % 256 == 0, "scratch sections stay aligned");
}   // namespace
static __device__ __align__(256) uint8_t na_scratch[kNaBytes];
namespace {
struct NaDevice {
    std::mutex mu;
    uint8_t* scratch = nullptr;                            // hipGetSymbolAddress(na_scratch) with this device current
    dev::BgzfMember mem[IRF_NOALLOC_MAX_MEMBERS];          // host copy of the member table
};
NaDevice g_na[kNaMaxDevices];
}   // namespace

extern "C" {

int irf_dreader_create(const irf_params* params, int device, irf_dreader** out) {
    if (!out) return IRF_E_ARG;
    irf_params P;
    if (params) {
        if (params->struct_size != sizeof(irf_params)) return IRF_E_ARG;
        P = *params;
    } else {
        irf_params_default(&P);
    }

    irf_dreader* r = new (std::nothrow) irf_dreader();
    if (!r) return IRF_E_NOMEM;
    r->P = P;
    r->device = device;
    int rc = r->init();
    if (rc != 0) { delete r; return rc; }
    *out = r;
    return 0;
}

void irf_dreader_destroy(irf_dreader* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    for (hipStream_t st : {r->s_copy, r->s_inf, r->stream})
        if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (auto& e : r->ev) if (e) (void)hipEventDestroy(e);
    for (auto& S : r->slot) for (auto& e : S.ev) if (e) (void)hipEventDestroy(e);
    delete r;
}

const char* irf_dreader_last_error(const irf_dreader* r) { return r ? r->err.c_str() : "null reader"; }

int irf_dreader_feed(irf_dreader* r, const uint8_t* bgzf, size_t len) {
    if (!r || (!bgzf && len)) return IRF_E_ARG;
    return r->feed(bgzf, len);
}

int irf_dreader_set_pipeline(irf_dreader* r, int on) {
    if (!r) return IRF_E_ARG;
    if (r->fed) return IRF_E_STATE;
    r->pipeline = on != 0;
    return 0;
}

uint32_t irf_dreader_member_slots(const irf_dreader* r) {
    if (!r || hipSetDevice(r->device) != hipSuccess) return 0;
    return dev::bgzf_inflate_slots();
}

int irf_dreader_header(const irf_dreader* r, const char* const** names, const uint32_t** lens) {
    if (!r) return IRF_E_ARG;
    if (!r->header_done) return IRF_E_STATE;
    if (names) *names = r->chr_name_ptrs.data();
    if (lens) *lens = r->chr_lens.data();
    return (int)r->chr_names.size();
}

int irf_dreader_batch(const irf_dreader* r, irf_batch* out) {
    if (!r || !out) return IRF_E_ARG;
    const OutSlot& O = r->out[r->out_cur];
    memset(out, 0, sizeof *out);
    out->n_frag = O.n_frag; out->n_blk = O.n_blk;
    out->frag_chr = O.frag_chr.p; out->frag_flags = O.frag_flags.p; out->frag_start = O.frag_start.p; out->frag_end = O.frag_end.p;
    out->frag_blk_off = O.frag_blk_off.p; out->blk_start = O.blk_start.p; out->blk_len = O.blk_len.p; out->blk_read = O.blk_read.p;
    out->mem_kind = IRF_MEM_DEVICE;
    return 0;
}

int irf_dreader_finish(irf_dreader* r) { return r ? r->finish() : IRF_E_ARG; }

int irf_dreader_get_stats(const irf_dreader* r, irf_reader_stats* out) {
    if (!r || !out) return IRF_E_ARG;
    *out = r->stats;
    return 0;
}

int irf_dreader_get_timing(const irf_dreader* r, irf_dreader_timing* out) {
    if (!r || !out) return IRF_E_ARG;
    *out = r->timing;
    return 0;
}

int irf_copy_to_host(void* dst_host, const void* src_device, size_t bytes) {
    if (!bytes) return 0;
    if (!dst_host || !src_device) return IRF_E_ARG;
    return hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : IRF_E_DEVICE;
}

int irf_gpu_bgzf_inflate(int device, const uint8_t* bgzf, size_t len, uint8_t* out, size_t out_cap, size_t* out_len) {
    if ((!bgzf && len) || !out_len) return IRF_E_ARG;
    if (hipSetDevice(device) != hipSuccess) return IRF_E_DEVICE;
    std::vector<irf::ZBlock> zb;
    size_t scan = 0, total_out = 0;
    const std::string e = irf::bgzf_locate(bgzf, len, scan, total_out, zb);
    if (!e.empty() || scan != len) return IRF_E_PARSE;
    *out_len = total_out;
    if (total_out > out_cap) return IRF_E_RANGE;
    if (total_out == 0) return 0;          // no members, or only empty ones (the EOF marker)
    if (!out) return IRF_E_ARG;
    std::vector<dev::BgzfMember> mem(zb.size());
    for (size_t i = 0; i < zb.size(); i++) mem[i] = dev::BgzfMember{zb[i].in_off, zb[i].out_off, (uint32_t)zb[i].clen, zb[i].isize, zb[i].crc, 0};
    DBuf<uint8_t> d_in, d_out;
    DBuf<dev::BgzfMember> d_mem;
    DBuf<uint32_t> d_status;
    if (d_in.reserve(len + dev::kInflateSlack) != hipSuccess || d_out.reserve(total_out + 64) != hipSuccess ||
        d_mem.reserve(mem.size()) != hipSuccess || d_status.reserve(mem.size() + 1) != hipSuccess)
        return IRF_E_NOMEM;
    const uint32_t n = (uint32_t)mem.size();
    bool ok = hipMemcpy(d_in.p, bgzf, len, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(d_in.p + len, 0, dev::kInflateSlack) == hipSuccess;
    ok = ok && hipMemcpy(d_mem.p, mem.data(), sizeof(dev::BgzfMember) * n, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(d_status.p + n, 0xff, 4) == hipSuccess;
    if (!ok) return IRF_E_DEVICE;
    dev::launch_bgzf_inflate(d_in.p, d_mem.p, n, d_out.p, d_status.p, nullptr);
    dev::launch_bgzf_crc(d_out.p, d_mem.p, n, d_status.p, nullptr);
    dev::launch_first_bad(d_status.p, n, d_status.p + n, nullptr);
    uint32_t first_bad = 0;
    if (hipMemcpy(&first_bad, d_status.p + n, 4, hipMemcpyDeviceToHost) != hipSuccess) return IRF_E_DEVICE;
    if (first_bad != 0xffffffffu) return IRF_E_PARSE;
    if (hipMemcpy(out, d_out.p, total_out, hipMemcpyDeviceToHost) != hipSuccess) return IRF_E_DEVICE;
    return 0;
}

int irf_gpu_bgzf_inflate_noalloc(int device, const uint8_t* bgzf, size_t len, uint8_t* out_device, size_t out_cap, size_t* out_len) {
    if ((!bgzf && len) || !out_len) return IRF_E_ARG;
    *out_len = 0;
    if (device < 0 || device >= kNaMaxDevices) return IRF_E_ARG;
    if (len > IRF_NOALLOC_MAX_BYTES) return IRF_E_RANGE;          // beyond the scratch array's staging area
    if (hipSetDevice(device) != hipSuccess) return IRF_E_DEVICE;
    NaDevice& D = g_na[device];
    std::lock_guard<std::mutex> lock(D.mu);                       // held to the end of this synchronous call: callers on one GPU queue
    if (!D.scratch) {
        // this device's copy of the array; the name `na_scratch` itself is only a host-side handle
        uint8_t* ptr = nullptr;
        if (hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(na_scratch)) != hipSuccess || !ptr) return IRF_E_DEVICE;
        D.scratch = ptr;
    }
    // member table (host: a fixed per-device array)
    uint32_t n = 0;
    size_t scan = 0, total_out = 0;
    while (true) {
        irf::ZBlock b;
        size_t member_bytes = 0;
        const char* what = "";
        const int rc = irf::bgzf_next_member(bgzf, len, scan, total_out, b, member_bytes, &what);
        if (rc < 0) return IRF_E_PARSE;
        if (rc == 0) break;
        if (n == IRF_NOALLOC_MAX_MEMBERS) return IRF_E_RANGE;
        D.mem[n++] = dev::BgzfMember{b.in_off, b.out_off, (uint32_t)b.clen, b.isize, b.crc, 0};
        total_out += b.isize;
        scan += member_bytes;
    }
    if (scan != len) return IRF_E_PARSE;
    *out_len = total_out;
    if (total_out + 64 > out_cap) return IRF_E_RANGE;
    if (total_out == 0) return 0;
    if (!out_device) return IRF_E_ARG;
    uint8_t* const d_in = D.scratch + kNaInOff;
    dev::BgzfMember* const d_mem = reinterpret_cast<dev::BgzfMember*>(D.scratch + kNaMemOff);
    uint32_t* const d_status = reinterpret_cast<uint32_t*>(D.scratch + kNaStatusOff);
    // everything the kernels read from the array is written here first (it keeps the last call's contents)
    bool ok = hipMemcpy(d_in, bgzf, len, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(d_in + len, 0, dev::kInflateSlack) == hipSuccess;
    ok = ok && hipMemcpy(d_mem, D.mem, sizeof(dev::BgzfMember) * n, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemset(d_status, 0, sizeof(uint32_t) * n) == hipSuccess;
    ok = ok && hipMemset(d_status + n, 0xff, 4) == hipSuccess;
    if (!ok) return IRF_E_DEVICE;
    dev::launch_bgzf_inflate(d_in, d_mem, n, out_device, d_status, nullptr);
    dev::launch_bgzf_crc(out_device, d_mem, n, d_status, nullptr);
    dev::launch_first_bad(d_status, n, d_status + n, nullptr);
    if (hipGetLastError() != hipSuccess) return IRF_E_DEVICE;
    uint32_t first_bad = 0;
    if (hipMemcpy(&first_bad, d_status + n, 4, hipMemcpyDeviceToHost) != hipSuccess) return IRF_E_DEVICE;
    if (first_bad != 0xffffffffu) return IRF_E_PARSE;
    return 0;
}


}   // extern "C"
