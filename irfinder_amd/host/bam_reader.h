// bam_reader.h — streaming BAM reader + fragment assembler.
//
// Replaces ⟨upstream⟩ BAM2blocks.{h,cpp} (readBamHeader, processAll,
// processPair, processSingle, cigar2block) for callers that do not bring their
// own reader.  Upstream pulls from an std::istream and calls back once per
// fragment; this reader is push-driven (feed any chunking of raw or BGZF
// bytes) and assembles fragments straight into SoA batches for the device.
// Fragment semantics: SURVEY App. B.1 (recalled; OQ-1/2/3 are irf_params).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/irf.h"
#include "soa_batch.h"

namespace irf {

// cigar words -> absolute aligned blocks; returns the reference end of the read
uint32_t cigar_to_blocks(uint32_t pos, const uint32_t* cigar, uint32_t n_cigar, std::vector<BlockSpan>& out);

// One complete BGZF member inside a buffer: where its DEFLATE data is, where its inflated bytes go
struct ZBlock { size_t in_off, clen, out_off; uint32_t isize, crc; };
// One step of bgzf_locate, without allocation: the member at buf + scan.  1 = complete (b filled, out_off = total_out,
// member_bytes = its size in the compressed stream), 0 = not complete yet, -1 = malformed (*what says why).
int bgzf_next_member(const uint8_t* buf, size_t size, size_t scan, size_t total_out, ZBlock& b, size_t& member_bytes, const char** what);
// Locates the complete BGZF members of buf[scan, size): appends them to out (out_off continues from total_out), moves
// scan behind the last complete member.  Returns the empty string, or what is wrong with the member at scan.
std::string bgzf_locate(const uint8_t* buf, size_t size, size_t& scan, size_t& total_out, std::vector<ZBlock>& out);
// ⟨upstream⟩ BAM2blocks::readBamHeader on p[0, avail): 0 = parsed (consumed bytes, names, lens), 1 = needs more bytes,
// IRF_E_PARSE = malformed (error text set)
int parse_bam_header(const uint8_t* p, size_t avail, std::vector<std::string>& names, std::vector<uint32_t>& lens,
                     size_t& consumed, std::string& error);

// A few persistent worker threads that share a loop over item indices (BGZF blocks are independent deflate members:
// SURVEY §8 f N1).  run() returns when every index has been processed; the caller's thread works too.
class WorkerPool {
public:
    explicit WorkerPool(int n_threads);
    ~WorkerPool();
    int size() const { return (int)threads_.size() + 1; }
    void run(size_t n_items, const std::function<void(int worker, size_t item)>& fn) { start(0, n_items, fn); wait(); }
    // items [first, end) go to the worker threads; the caller may do other work before wait(), which helps to finish
    void start(size_t first, size_t end, const std::function<void(int worker, size_t item)>& fn);
    void wait();

private:
    void worker_main(int id);
    void drain(int id);
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    const std::function<void(int, size_t)>* fn_ = nullptr;
    size_t n_items_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t generation_ = 0;
    int busy_ = 0;
    bool stop_ = false;
};

class BamReader {
public:
    explicit BamReader(const irf_params& p);
    ~BamReader();
    int feed(const uint8_t* data, size_t len, bool bgzf);
    int finish();
    bool header_done() const { return header_done_; }
    // BGZF blocks of one feed() call are inflated by n threads (the record parser stays sequential: pairing is)
    void set_threads(int n);

    std::vector<std::string> chr_names;
    std::vector<uint32_t> chr_lens;
    std::vector<const char*> chr_name_ptrs;
    BatchBuilder batch;
    irf_reader_stats stats;
    std::string error;

private:
    struct Rec {
        int32_t ref_id = -1, pos = 0;
        uint16_t flag = 0;
        std::vector<uint32_t> cigar;
    };
    int inflate_blocks();
    int parse_raw(size_t limit, bool may_compact);
    int parse_header(bool& need_more, size_t limit);
    void on_record(const uint8_t* p, uint32_t block_size, bool& bad);
    void on_paired_by_name(const uint8_t* name, uint32_t l_read_name);
    void pair_or_drop(const Rec& a, const Rec& b);
    void emit_single(const Rec& r);
    void emit_pair(const Rec& a, const Rec& b);

    irf_params P;
    std::vector<uint8_t> zbuf_;
    size_t zoff_ = 0;
    std::vector<uint8_t> raw_;
    size_t roff_ = 0;
    bool header_done_ = false;
    bool have_pending_ = false;
    Rec pending_, cur_;
    // params.mate_join: records whose mate has not been seen yet, by read name (coordinate-sorted input)
    std::unordered_map<std::string, Rec> waiting_;
    std::vector<BlockSpan> b1_, b2_, merged_;
    std::vector<ZBlock> zblocks_;
    std::vector<void*> zs_;              // one z_stream per worker
    WorkerPool* pool_ = nullptr;
};

}   // namespace irf
