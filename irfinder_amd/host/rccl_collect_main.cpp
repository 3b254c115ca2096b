// rccl_collect_main.cpp — irf_rccl_collect: one process per GPU, chromosome shards, results collected over RCCL.
//
//   IRF_RANK=r IRF_WORLD=n IRF_ID_FILE=path irf_rccl_collect OUTDIR ref-cover.bed ref-sj.ref ref-read-continues.ref ref-ROI.bed < raw.bam
//
// The C/C++ host side of SURVEY.md §8 e as a program (rccl_collect.h is the part a host would reuse): every rank reads
// the same BAM stream (raw, as `gzip -cd x.bam` yields it), owns the chromosomes the longest-first deal gives it
// (foreign fragments are ignored by the count kernel), finalizes, and irf_rccl_collect brings the tables into rank 0's
// context, which writes the IRFinder-*.txt files and the reader summary.  IRF_WORLD=1 (the default) runs the same calls
// with a one-rank communicator.  The ncclUniqueId travels through IRF_ID_FILE (rank 0 writes it, the others wait for it).
// Device = IRF_DEVICE, else the rank.  No upstream counterpart: upstream is one process.
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rccl_collect.h"

static int die(const char* what, const char* msg) {
    fprintf(stderr, "irf_rccl_collect: %s: %s\n", what, msg);
    return 1;
}
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: [IRF_RANK=r IRF_WORLD=n IRF_ID_FILE=path] %s OUTDIR ref-cover.bed ref-sj.ref ref-read-continues.ref ref-ROI.bed < raw.bam\n", argv[0]);
        return 2;
    }
    const int rank = env_int("IRF_RANK", 0), world = env_int("IRF_WORLD", 1), device = env_int("IRF_DEVICE", rank);
    if (world < 1 || rank < 0 || rank >= world) return die("IRF_RANK / IRF_WORLD", "out of range");
    if (hipSetDevice(device) != hipSuccess) return die("hipSetDevice", "no such device");

    // ---- the communicator ----
    ncclUniqueId id;
    const char* id_file = getenv("IRF_ID_FILE");
    if (rank == 0) {
        if (ncclGetUniqueId(&id) != ncclSuccess) return die("ncclGetUniqueId", "failed");
        if (world > 1) {
            if (!id_file) return die("IRF_ID_FILE", "needed with IRF_WORLD > 1");
            const std::string tmp = std::string(id_file) + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&id, sizeof id, 1, f) != 1) return die("IRF_ID_FILE", "cannot write");
            fclose(f);
            if (rename(tmp.c_str(), id_file) != 0) return die("IRF_ID_FILE", "cannot rename");
        }
    } else {
        if (!id_file) return die("IRF_ID_FILE", "needed with IRF_WORLD > 1");
        FILE* f = nullptr;
        for (int tries = 0; tries < 600 && !(f = fopen(id_file, "rb")); tries++) usleep(100000);
        if (!f || fread(&id, sizeof id, 1, f) != 1) return die("IRF_ID_FILE", "rank 0 did not write it");
        fclose(f);
    }
    ncclComm_t comm;
    if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) return die("ncclCommInitRank", "failed");
    hipStream_t stream;
    if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return die("hipStreamCreate", "failed");

    // ---- this rank's context ----
    irf_params P;
    irf_params_default(&P);
    char err[256] = {0};
    irf_ref* ref = nullptr;
    if (irf_ref_load_files(argv[2], argv[3], argv[4], argv[5], &ref, err, sizeof err) != 0) return die("reference tables", err);
    irf_ctx* ctx = nullptr;
    int rc = irf_create(&P, ref, device, &ctx);
    irf_ref_free(ref);
    if (rc != 0) return die("irf_create", "no HIP device or bad parameters");
    if (irf_set_input_order(ctx, IRF_ORDER_ANY) != 0) return die("irf_set_input_order", irf_last_error(ctx));
    irf_reader* rd = nullptr;
    if (irf_reader_create(&P, &rd) != 0) return die("reader", "cannot create");

    bool mapped = false;
    auto drain = [&](bool all) -> int {
        if (!mapped) {
            const char* const* names = nullptr;
            const uint32_t* lens = nullptr;
            const int n = irf_reader_header(rd, &names, &lens);
            if (n < 0) return 0;
            // chromosomes dealt longest first to the rank with the least bases so far (every rank computes the same deal)
            std::vector<uint8_t> owned((size_t)n, 0);
            std::vector<int> order((size_t)n);
            for (int i = 0; i < n; i++) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] > lens[b]; });
            std::vector<unsigned long long> load((size_t)world, 0ull);
            for (int i : order) {
                const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
                load[r] += lens[i];
                if (r == rank) owned[i] = 1;
            }
            if (irf_chr_map_update(ctx, (uint32_t)n, names, lens, world > 1 ? owned.data() : nullptr) != 0) return die("irf_chr_map_update", irf_last_error(ctx));
            mapped = true;
        }
        irf_batch b;
        irf_reader_batch(rd, &b);
        if (b.n_frag == 0 || (!all && b.n_frag < (1u << 20))) return 0;
        uint64_t bid = 0;
        if (irf_submit_batch(ctx, &b, &bid) != 0 || irf_wait(ctx, bid) != 0) return die("irf_submit_batch", irf_last_error(ctx));
        irf_reader_clear_batch(rd);
        return 0;
    };
    std::vector<uint8_t> buf(8u << 20);
    size_t n;
    while ((n = fread(buf.data(), 1, buf.size(), stdin)) > 0) {
        if (irf_reader_feed(rd, buf.data(), n, 0) != 0) return die("BAM", irf_reader_last_error(rd));
        if (int r = drain(false)) return r;
    }
    if (irf_reader_finish(rd) != 0) return die("BAM", irf_reader_last_error(rd));
    if (int r = drain(true)) return r;
    if (!mapped) return die("BAM", "no header");
    if (irf_finalize(ctx) != 0) return die("irf_finalize", irf_last_error(ctx));

    // ---- collection ----
    rc = irf_rccl_collect(ctx, comm, rank, world, 0, P.dir_ratio, stream);
    if (rc != 0) {
        fprintf(stderr, "irf_rccl_collect: rank %d: collection failed with %d (%s)\n", rank, rc, rc == IRF_RCCL_E_NCCL ? "RCCL" : rc == IRF_RCCL_E_HIP ? "HIP" : irf_last_error(ctx));
        return 1;
    }
    if (rank == 0) {
        if (irf_write_outputs(ctx, argv[1]) != 0) return die("irf_write_outputs", irf_last_error(ctx));
        irf_reader_stats st;
        irf_reader_get_stats(rd, &st);
        printf("Total reads processed\t%llu\nTotal nucleotides\t%llu\nTotal singles processed\t%llu\nTotal pairs processed\t%llu\n"
               "Short pairs\t%llu\nIntersect pairs\t%llu\nLong pairs\t%llu\nSkipped reads\t%llu\nError reads\t%llu\n",
               (unsigned long long)st.n_records, (unsigned long long)st.n_nucleotides, (unsigned long long)st.n_singles,
               (unsigned long long)st.n_pairs, (unsigned long long)st.n_pairs_short, (unsigned long long)st.n_pairs_intersect,
               (unsigned long long)st.n_pairs_long, (unsigned long long)st.n_skipped, (unsigned long long)st.n_error);
    }
    irf_reader_destroy(rd);
    irf_destroy(ctx);
    (void)hipStreamDestroy(stream);
    ncclCommDestroy(comm);
    return 0;
}
