// reader_abi.cpp — C ABI of the host BAM reader (include/irf.h, "Host reader").
#include <cstring>
#include <new>
#include <string>

#include "../../include/irf.h"
#include "bam_reader.h"

struct irf_reader {
    irf::BamReader impl;
    explicit irf_reader(const irf_params& p) : impl(p) {}
};

extern "C" {

int irf_reader_create(const irf_params* params, irf_reader** out) {
    if (!out) return IRF_E_ARG;
    irf_params P;
    if (params) {
        if (params->struct_size != sizeof(irf_params)) return IRF_E_ARG;
        P = *params;
    } else {
        irf_params_default(&P);
    }
    irf_reader* r = new (std::nothrow) irf_reader(P);
    if (!r) return IRF_E_NOMEM;
    *out = r;
    return 0;
}

void irf_reader_destroy(irf_reader* r) { delete r; }

const char* irf_reader_last_error(const irf_reader* r) { return r ? r->impl.error.c_str() : "null reader"; }

int irf_reader_feed(irf_reader* r, const uint8_t* data, size_t len, int is_bgzf) {
    if (!r || (!data && len)) return IRF_E_ARG;
    return r->impl.feed(data, len, is_bgzf != 0);
}

int irf_reader_set_threads(irf_reader* r, int n_threads) {
    if (!r || n_threads < 1 || n_threads > 64) return IRF_E_ARG;
    r->impl.set_threads(n_threads);
    return 0;
}

int irf_reader_header(const irf_reader* r, const char* const** names, const uint32_t** lens) {
    if (!r) return IRF_E_ARG;
    if (!r->impl.header_done()) return IRF_E_STATE;
    if (names) *names = r->impl.chr_name_ptrs.data();
    if (lens) *lens = r->impl.chr_lens.data();
    return (int)r->impl.chr_names.size();
}

int irf_reader_batch(const irf_reader* r, irf_batch* out) {
    if (!r || !out) return IRF_E_ARG;
    r->impl.batch.view(out);
    return 0;
}

void irf_reader_clear_batch(irf_reader* r) { if (r) r->impl.batch.clear(); }

int irf_reader_finish(irf_reader* r) { return r ? r->impl.finish() : IRF_E_ARG; }

int irf_reader_get_stats(const irf_reader* r, irf_reader_stats* out) {
    if (!r || !out) return IRF_E_ARG;
    *out = r->impl.stats;
    return 0;
}

int irf_cigar2block(const uint32_t* cigar, uint32_t n_cigar, uint32_t* starts, uint32_t* lens, uint32_t cap,
                    uint32_t* n_blocks, uint32_t* genome_len) {
    if ((!cigar && n_cigar) || !n_blocks || !genome_len) return IRF_E_ARG;
    std::vector<irf::BlockSpan> b;
    uint32_t end = irf::cigar_to_blocks(0, cigar, n_cigar, b);
    *n_blocks = (uint32_t)b.size();
    *genome_len = end;
    if (b.size() > cap) return IRF_E_RANGE;
    for (size_t i = 0; i < b.size(); i++) { starts[i] = b[i].start; lens[i] = b[i].len; }
    return 0;
}

}   // extern "C"
