// soa_batch.h — growable host-side structure-of-arrays fragment batch (the
// payload of irf_submit_batch).  One fragment = one upstream FragmentBlocks.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/irf.h"

namespace irf {

struct BlockSpan { uint32_t start, len; };

struct BatchBuilder {
    std::vector<uint32_t> frag_chr, frag_start, frag_end, frag_blk_off;
    std::vector<uint8_t> frag_flags;
    std::vector<uint32_t> blk_start, blk_len;
    std::vector<uint8_t> blk_read;

    BatchBuilder() { frag_blk_off.push_back(0); }

    size_t n_frag() const { return frag_chr.size(); }
    size_t n_blk() const { return blk_start.size(); }

    void clear() {
        frag_chr.clear(); frag_start.clear(); frag_end.clear(); frag_flags.clear();
        frag_blk_off.clear(); frag_blk_off.push_back(0);
        blk_start.clear(); blk_len.clear(); blk_read.clear();
    }
    void reserve(size_t nf, size_t nb) {
        frag_chr.reserve(nf); frag_start.reserve(nf); frag_end.reserve(nf); frag_flags.reserve(nf);
        frag_blk_off.reserve(nf + 1);
        blk_start.reserve(nb); blk_len.reserve(nb); blk_read.reserve(nb);
    }
    // blocks of read 0 then (if two) read 1, absolute coordinates
    void add(uint32_t chr, bool dir, uint32_t frag_s, uint32_t frag_e, const BlockSpan* b0, size_t n0,
             const BlockSpan* b1, size_t n1, bool two_reads) {
        frag_chr.push_back(chr);
        frag_flags.push_back((uint8_t)((dir ? IRF_FRAG_DIR : 0u) | (two_reads ? IRF_FRAG_TWO : 0u)));
        frag_start.push_back(frag_s);
        frag_end.push_back(frag_e);
        for (size_t i = 0; i < n0; i++) { blk_start.push_back(b0[i].start); blk_len.push_back(b0[i].len); blk_read.push_back(0); }
        for (size_t i = 0; i < n1; i++) { blk_start.push_back(b1[i].start); blk_len.push_back(b1[i].len); blk_read.push_back(1); }
        frag_blk_off.push_back((uint32_t)blk_start.size());
    }
    void view(irf_batch* out) const {
        memset(out, 0, sizeof *out);
        out->n_frag = (uint32_t)frag_chr.size();
        out->n_blk = (uint32_t)blk_start.size();
        out->frag_chr = frag_chr.data(); out->frag_flags = frag_flags.data();
        out->frag_start = frag_start.data(); out->frag_end = frag_end.data();
        out->frag_blk_off = frag_blk_off.data();
        out->blk_start = blk_start.data(); out->blk_len = blk_len.data(); out->blk_read = blk_read.data();
        out->mem_kind = IRF_MEM_HOST;
    }
};

}   // namespace irf
