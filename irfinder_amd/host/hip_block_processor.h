// hip_block_processor.h — the drop-in: a ReadBlockProcessor that forwards to the
// MI355X engine through the C ABI (include/irf.h).
//
// Upstream (⟨upstream⟩ src/irfinder/irfinder.cpp main(); source not in
// /root/reference — SURVEY §0) registers five processors with BAM2blocks:
//
//     BB.registerCallbackChrMappingChange(std::bind(&X::ChrMapUpdate, &oX, _1));   x5
//     BB.registerCallbackProcessBlocks  (std::bind(&X::ProcessBlocks, &oX, _1));  x5
//
// With this header a maintainer registers ONE HipReadBlockProcessor instead; it
// batches the per-fragment calls into pinned structure-of-arrays buffers (two,
// used alternately) and submits them asynchronously.  Build inside upstream
// with -DIRF_USE_UPSTREAM_TYPES so that upstream's own FragmentBlocks.h /
// ReadBlockProcessor.h are used; otherwise the minimal declarations below (same
// member names) stand in.  Header-only, needs only libirfhip.so.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/irf.h"

#ifdef IRF_USE_UPSTREAM_TYPES
#include "FragmentBlocks.h"
#include "ReadBlockProcessor.h"
#else
// Same shape as ⟨upstream⟩ FragmentBlocks.h / ReadBlockProcessor.h (recalled).
class FragmentBlocks {
public:
    std::string readName;
    std::vector<int> rStarts[2];      // block starts relative to readStart[r]
    std::vector<int> rLens[2];
    unsigned int readStart[2] = {0, 0};
    unsigned int readEnd[2] = {0, 0};
    int readCount = 1;
    int chr_id = 0;
    bool direction = false;           // true: fragment from the + strand
};
class ReadBlockProcessor {
public:
    virtual ~ReadBlockProcessor() {}
    virtual void ProcessBlocks(const FragmentBlocks& fragblock) = 0;
    virtual void ChrMapUpdate(const std::vector<std::string>& chrmap) = 0;
};
#endif

class HipReadBlockProcessor : public ReadBlockProcessor {
public:
    // Loads the four reference files (what the five loadRef calls of main() did).
    HipReadBlockProcessor(const char* ref_cover_bed, const char* ref_sj_ref, const char* ref_read_continues_ref,
                          const char* ref_roi_bed, int device = 0, const irf_params* params = nullptr,
                          uint32_t batch_fragments = 1u << 20, uint32_t blocks_per_fragment_hint = 6,
                          uint32_t input_order = IRF_ORDER_ANY) {
        char err[256] = {0};
        irf_ref* ref = nullptr;
        int rc = irf_ref_load_files(ref_cover_bed, ref_sj_ref, ref_read_continues_ref, ref_roi_bed, &ref, err, sizeof err);
        if (rc != 0) throw std::runtime_error(std::string("reference tables: ") + err);
        rc = irf_create(params, ref, device, &ctx_);
        irf_ref_free(ref);
        if (rc != 0) throw std::runtime_error("irf_create failed (no HIP device?)");
        // BAM2blocks hands over fragments in the order of upstream's input: collated by read name, scattered over the
        // genome.  The engine bins every batch by coordinate first (reorder.hip) unless the caller knows better.
        if (irf_set_input_order(ctx_, input_order) != 0) {
            const std::string msg = std::string("irf_set_input_order: ") + irf_last_error(ctx_);
            irf_destroy(ctx_);
            throw std::runtime_error(msg);
        }
        cap_frag_ = batch_fragments;
        cap_blk_ = (uint64_t)batch_fragments * blocks_per_fragment_hint;
        for (int s = 0; s < 2; s++) alloc_slot(slot_[s]);
    }
    ~HipReadBlockProcessor() override {
        for (int s = 0; s < 2; s++) free_slot(slot_[s]);
        irf_destroy(ctx_);
    }
    HipReadBlockProcessor(const HipReadBlockProcessor&) = delete;
    HipReadBlockProcessor& operator=(const HipReadBlockProcessor&) = delete;

    // Upstream's ChrMapUpdate carries names only; the engine also needs the l_ref values of the BAM header,
    // which BAM2blocks has at hand when it fires the callback.  Call this first.
    void setChrLens(const std::vector<uint32_t>& lens) { lens_ = lens; }

    void ChrMapUpdate(const std::vector<std::string>& chrmap) override {
        if (lens_.size() != chrmap.size()) throw std::runtime_error("HipReadBlockProcessor: setChrLens() before ChrMapUpdate()");
        std::vector<const char*> names;
        for (auto& s : chrmap) names.push_back(s.c_str());
        check(irf_chr_map_update(ctx_, (uint32_t)chrmap.size(), names.data(), lens_.data(), nullptr));
    }

    void ProcessBlocks(const FragmentBlocks& f) override {
        const size_t nb = f.rLens[0].size() + (f.readCount == 2 ? f.rLens[1].size() : 0);
        Slot* s = &slot_[cur_];
        if (s->n_frag + 1 > cap_frag_ || s->n_blk + nb > cap_blk_) {
            if (nb > cap_blk_) throw std::runtime_error("HipReadBlockProcessor: fragment has more blocks than a batch holds");
            flush();
            s = &slot_[cur_];
        }
        const uint32_t i = s->n_frag++;
        s->frag_chr[i] = (uint32_t)f.chr_id;
        s->frag_flags[i] = (uint8_t)((f.direction ? IRF_FRAG_DIR : 0u) | (f.readCount == 2 ? IRF_FRAG_TWO : 0u));
        s->frag_start[i] = f.readStart[0];
        uint32_t fe = f.readEnd[0];
        if (f.readCount == 2 && f.readEnd[1] > fe) fe = f.readEnd[1];
        s->frag_end[i] = fe;
        for (int r = 0; r < f.readCount; r++)
            for (size_t j = 0; j < f.rLens[r].size(); j++) {
                const uint32_t k = s->n_blk++;
                s->blk_start[k] = f.readStart[r] + (uint32_t)f.rStarts[r][j];
                s->blk_len[k] = (uint32_t)f.rLens[r][j];
                s->blk_read[k] = (uint8_t)r;
            }
        s->frag_blk_off[i + 1] = s->n_blk;
    }

    // End of input (after BAM2blocks::processAll returns): counts the last batch, computes the IR tables.
    void finish() {
        flush();
        check(irf_finalize(ctx_));
    }
    // The five / six IRFinder-*.txt files (what the tail of main() wrote).
    void writeOutputs(const std::string& outdir) { check(irf_write_outputs(ctx_, outdir.c_str())); }
    irf_ctx* context() { return ctx_; }

private:
    struct Slot {
        uint32_t *frag_chr = nullptr, *frag_start = nullptr, *frag_end = nullptr, *frag_blk_off = nullptr;
        uint32_t *blk_start = nullptr, *blk_len = nullptr;
        uint8_t *frag_flags = nullptr, *blk_read = nullptr;
        uint32_t n_frag = 0, n_blk = 0;
        uint64_t batch_id = 0;
        bool in_flight = false;
    };
    template <class T>
    T* pinned(size_t n) {
        void* p = nullptr;
        if (irf_host_alloc(n * sizeof(T), &p) != 0) throw std::bad_alloc();
        return (T*)p;
    }
    void alloc_slot(Slot& s) {
        s.frag_chr = pinned<uint32_t>(cap_frag_); s.frag_start = pinned<uint32_t>(cap_frag_); s.frag_end = pinned<uint32_t>(cap_frag_);
        s.frag_blk_off = pinned<uint32_t>((size_t)cap_frag_ + 1); s.frag_flags = pinned<uint8_t>(cap_frag_);
        s.blk_start = pinned<uint32_t>(cap_blk_); s.blk_len = pinned<uint32_t>(cap_blk_); s.blk_read = pinned<uint8_t>(cap_blk_);
        s.frag_blk_off[0] = 0;
    }
    void free_slot(Slot& s) {
        void* p[] = {s.frag_chr, s.frag_start, s.frag_end, s.frag_blk_off, s.frag_flags, s.blk_start, s.blk_len, s.blk_read};
        for (void* q : p) if (q) irf_host_free(q);
    }
    // submit the current slot, switch to the other one (waiting until the engine has copied it out)
    void flush() {
        Slot& s = slot_[cur_];
        if (s.n_frag) {
            irf_batch b;
            memset(&b, 0, sizeof b);
            b.n_frag = s.n_frag; b.n_blk = s.n_blk;
            b.frag_chr = s.frag_chr; b.frag_flags = s.frag_flags; b.frag_start = s.frag_start; b.frag_end = s.frag_end;
            b.frag_blk_off = s.frag_blk_off; b.blk_start = s.blk_start; b.blk_len = s.blk_len; b.blk_read = s.blk_read;
            b.mem_kind = IRF_MEM_HOST;
            check(irf_submit_batch(ctx_, &b, &s.batch_id));
            s.in_flight = true;
        }
        cur_ ^= 1;
        Slot& n = slot_[cur_];
        if (n.in_flight) { check(irf_wait(ctx_, n.batch_id)); n.in_flight = false; }
        n.n_frag = 0; n.n_blk = 0; n.frag_blk_off[0] = 0;
    }
    void check(int rc) {
        if (rc != 0) throw std::runtime_error(std::string("irf: ") + irf_last_error(ctx_));
    }
    irf_ctx* ctx_ = nullptr;
    std::vector<uint32_t> lens_;
    Slot slot_[2];
    int cur_ = 0;
    uint32_t cap_frag_ = 0;
    uint64_t cap_blk_ = 0;
};
