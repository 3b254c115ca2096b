// irfinder_main.cpp — command-line drop-in for the upstream engine binary.
//
//   irfinder_hip OUTDIR ref-cover.bed ref-sj.ref ref-read-continues.ref ref-ROI.bed [--bgzf] [--device N]
//                [--via-adapter] [--sorted] [--threads N] [--device-reader [--feed-mb N]] < input.bam
//
// Same positional argv as ⟨upstream⟩ irfinder.cpp main(): reads the
// BAM stream from stdin (raw, as `gzip -cd x.bam` yields it, or BGZF with
// --bgzf), writes the IRFinder-*.txt files into OUTDIR and the reader summary
// to stdout.  --sorted: the input is coordinate-sorted, mates are joined by read name (params.mate_join; upstream
// needs name-collated input because it pairs adjacent records).  --threads N: BGZF members are inflated by N threads
// (upstream's pipe has one gzip process).  --via-adapter routes every fragment through the
// ReadBlockProcessor-shaped adapter (hip_block_processor.h) instead of handing
// the reader's SoA batches to the engine directly; both give identical files.
// --device-reader: the input is BGZF and is inflated, parsed and assembled on the GPU (irf_dreader_*, SURVEY §8 f N1):
// the host only reads stdin in --feed-mb pieces (default 0: sized from the members seen so far to what the device
// inflates at once, irf_dreader_member_slots: the inflate kernel is one lane per BGZF member and needs ~49 k members
// in flight to fill the GPU) and locates the members; same files, same summary.
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/irf.h"
#include "hip_block_processor.h"

static int die(const char* what, const char* msg) {
    fprintf(stderr, "irfinder_hip: %s: %s\n", what, msg);
    return 1;
}

static void print_summary(const irf_reader_stats& st) {
    printf("Total reads processed\t%llu\nTotal nucleotides\t%llu\nTotal singles processed\t%llu\nTotal pairs processed\t%llu\n"
           "Short pairs\t%llu\nIntersect pairs\t%llu\nLong pairs\t%llu\nSkipped reads\t%llu\nError reads\t%llu\n",
           (unsigned long long)st.n_records, (unsigned long long)st.n_nucleotides, (unsigned long long)st.n_singles,
           (unsigned long long)st.n_pairs, (unsigned long long)st.n_pairs_short, (unsigned long long)st.n_pairs_intersect,
           (unsigned long long)st.n_pairs_long, (unsigned long long)st.n_skipped, (unsigned long long)st.n_error);
}

// stdin (BGZF) -> irf_dreader -> device-resident batches -> irf_submit_batch: no record ever exists in host memory
static int run_device_reader(const std::vector<const char*>& pos, const irf_params& P, int device, size_t feed_bytes) {
    char err[256] = {0};
    irf_ref* ref = nullptr;
    if (irf_ref_load_files(pos[1], pos[2], pos[3], pos[4], &ref, err, sizeof err) != 0) return die("reference tables", err);
    irf_ctx* ctx = nullptr;
    int rc = irf_create(&P, ref, device, &ctx);
    irf_ref_free(ref);
    if (rc != 0) return die("irf_create", "no HIP device or bad parameters");
    // without --sorted the file is name-collated: its fragments arrive in no coordinate order, and the engine bins every
    // batch by coordinate before it counts it (same tables, a third of the count time)
    if (!P.mate_join && irf_set_input_order(ctx, IRF_ORDER_ANY) != 0) return die("irf_set_input_order", irf_last_error(ctx));
    irf_dreader* dr = nullptr;
    if (irf_dreader_create(&P, device, &dr) != 0) return die("irf_dreader_create", "no HIP device or bad parameters");
    // pipelined: while a piece is copied and inflated, the fragments of the piece before it are assembled and counted,
    // and stdin is read for the next one
    if (irf_dreader_set_pipeline(dr, 1) != 0) return die("irf_dreader_set_pipeline", irf_dreader_last_error(dr));
    // feed_bytes == 0: pieces sized so that a feed holds a little less than the members the device inflates at once
    // (the inflate kernel is one lane per member; its time does not depend on how many of its lanes are used)
    const bool adaptive = feed_bytes == 0;
    const uint32_t slots = irf_dreader_member_slots(dr);
    if (adaptive) feed_bytes = (size_t)256 << 20;
    bool mapped = false;
    auto take_batch = [&]() -> int {
        if (!mapped) {
            const char* const* names = nullptr;
            const uint32_t* lens = nullptr;
            const int nc = irf_dreader_header(dr, &names, &lens);
            if (nc >= 0) {
                if (irf_chr_map_update(ctx, (uint32_t)nc, names, lens, nullptr) != 0) return die("irf_chr_map_update", irf_last_error(ctx));
                mapped = true;
            }
        }
        irf_batch b;
        irf_dreader_batch(dr, &b);
        if (b.n_frag) {
            uint64_t id = 0;
            if (!mapped) return die("BAM", "fragments before the header");
            if (irf_submit_batch(ctx, &b, &id) != 0) return die("irf_submit_batch", irf_last_error(ctx));
        }
        return 0;
    };
    // stdin redirected from a regular file: the pieces are fed straight from a mapping of the file (no read() copy)
    const uint8_t* map = nullptr;
    size_t map_len = 0, map_pos = 0;
    {
        struct stat st;
        if (fstat(0, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            const off_t at = lseek(0, 0, SEEK_CUR);
            if (at >= 0 && at < st.st_size) {
                void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, 0, 0);
                if (m != MAP_FAILED) {
                    (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                    map = static_cast<const uint8_t*>(m); map_len = (size_t)st.st_size; map_pos = (size_t)at;
                }
            }
        }
    }
    size_t piece = feed_bytes;
    std::vector<uint8_t> buf(map ? 0 : piece);
    while (true) {
        const uint8_t* data = nullptr;
        size_t have = 0, n;
        if (map) {
            data = map + map_pos;
            have = std::min(piece, map_len - map_pos);
            map_pos += have;
        } else {
            if (buf.size() != piece) buf.resize(piece);
            while (have < buf.size() && (n = fread(buf.data() + have, 1, buf.size() - have, stdin)) > 0) have += n;
            data = buf.data();
        }
        if (!have) break;
        // the batch submitted last lives in one of the reader's two slots: this feed fills the other one, so the
        // count kernel of that batch must have finished only before the feed after this one
        if (irf_sync(ctx) != 0) return die("irf_sync", irf_last_error(ctx));
        if (irf_dreader_feed(dr, data, have) != 0) return die("BAM", irf_dreader_last_error(dr));
        if (int rc = take_batch()) return rc;
        if (adaptive && slots) {
            irf_dreader_timing tm;
            if (irf_dreader_get_timing(dr, &tm) == 0 && tm.n_members > 64 && tm.bytes_in) {
                const double per_member = (double)tm.bytes_in / (double)tm.n_members;
                double want = 0.97 * (double)slots * per_member;
                want = std::min(std::max(want, 64.0 * 1048576.0), 4096.0 * 1048576.0);
                piece = (size_t)want;
            }
        }
    }
    if (map) (void)munmap(const_cast<uint8_t*>(map), map_len);
    if (irf_sync(ctx) != 0) return die("irf_sync", irf_last_error(ctx));
    if (irf_dreader_finish(dr) != 0) return die("BAM", irf_dreader_last_error(dr));
    if (int rc = take_batch()) return rc;
    if (!mapped) return die("BAM", "no header");
    if (irf_finalize(ctx) != 0) return die("irf_finalize", irf_last_error(ctx));
    if (irf_write_outputs(ctx, pos[0]) != 0) return die("irf_write_outputs", irf_last_error(ctx));
    irf_reader_stats st;
    irf_dreader_get_stats(dr, &st);
    print_summary(st);
    irf_dreader_destroy(dr);
    irf_destroy(ctx);
    return 0;
}

int main(int argc, char** argv) {
    std::vector<const char*> pos;
    bool bgzf = false, via_adapter = false, sorted_input = false, device_reader = false;
    int device = 0, threads = 1, feed_mb = 0;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--bgzf")) bgzf = true;
        else if (!strcmp(argv[i], "--via-adapter")) via_adapter = true;
        else if (!strcmp(argv[i], "--sorted")) sorted_input = true;
        else if (!strcmp(argv[i], "--device-reader")) device_reader = true;
        else if (!strcmp(argv[i], "--feed-mb") && i + 1 < argc) feed_mb = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else pos.push_back(argv[i]);
    }
    if (pos.size() < 5) {
        fprintf(stderr, "usage: %s OUTDIR ref-cover.bed ref-sj.ref ref-read-continues.ref ref-ROI.bed [--bgzf] [--sorted] [--threads N] [--device N] < bam\n", argv[0]);
        return 2;
    }
    irf_params P;
    irf_params_default(&P);
    P.mate_join = sorted_input ? 1u : 0u;
    if (device_reader) {
        if (feed_mb < 0 || feed_mb > 16384) return die("--feed-mb", "0 (sized by the reader) or 1..16384");
        return run_device_reader(pos, P, device, (size_t)feed_mb << 20);
    }
    irf_reader* rd = nullptr;
    if (irf_reader_create(&P, &rd) != 0) return die("reader", "cannot create");
    if (threads != 1 && irf_reader_set_threads(rd, threads) != 0) return die("--threads", "1..64");

    const size_t kChunk = 8u << 20, kBatchFrags = 1u << 20;
    std::vector<uint8_t> buf(kChunk);
    irf_ctx* ctx = nullptr;
    HipReadBlockProcessor* adapter = nullptr;
    bool mapped = false;
    try {
        if (via_adapter) {
            adapter = new HipReadBlockProcessor(pos[1], pos[2], pos[3], pos[4], device, &P, (uint32_t)kBatchFrags, 6,
                                                sorted_input ? IRF_ORDER_SORTED : IRF_ORDER_ANY);
            ctx = adapter->context();
        } else {
            char err[256] = {0};
            irf_ref* ref = nullptr;
            if (irf_ref_load_files(pos[1], pos[2], pos[3], pos[4], &ref, err, sizeof err) != 0) return die("reference tables", err);
            int rc = irf_create(&P, ref, device, &ctx);
            irf_ref_free(ref);
            if (rc != 0) return die("irf_create", "no HIP device or bad parameters");
            if (!sorted_input && irf_set_input_order(ctx, IRF_ORDER_ANY) != 0) return die("irf_set_input_order", irf_last_error(ctx));
        }
        auto drain = [&](bool all) -> int {
            if (!mapped) {
                const char* const* names = nullptr;
                const uint32_t* lens = nullptr;
                int n = irf_reader_header(rd, &names, &lens);
                if (n < 0) return 0;
                if (adapter) {
                    std::vector<std::string> nm;
                    for (int i = 0; i < n; i++) nm.emplace_back(names[i]);
                    adapter->setChrLens(std::vector<uint32_t>(lens, lens + n));
                    adapter->ChrMapUpdate(nm);
                } else if (irf_chr_map_update(ctx, (uint32_t)n, names, lens, nullptr) != 0) {
                    return die("irf_chr_map_update", irf_last_error(ctx));
                }
                mapped = true;
            }
            irf_batch b;
            irf_reader_batch(rd, &b);
            if (b.n_frag == 0 || (!all && b.n_frag < kBatchFrags)) return 0;
            if (adapter) {
                FragmentBlocks f;
                for (uint32_t i = 0; i < b.n_frag; i++) {
                    f.chr_id = (int)b.frag_chr[i];
                    f.direction = (b.frag_flags[i] & IRF_FRAG_DIR) != 0;
                    f.readCount = (b.frag_flags[i] & IRF_FRAG_TWO) ? 2 : 1;
                    for (int r = 0; r < 2; r++) { f.rStarts[r].clear(); f.rLens[r].clear(); f.readStart[r] = f.readEnd[r] = 0; }
                    f.readStart[0] = b.frag_start[i];
                    bool seen1 = false;
                    for (uint32_t k = b.frag_blk_off[i]; k < b.frag_blk_off[i + 1]; k++) {
                        int r = b.blk_read[k] ? 1 : 0;
                        if (r == 1 && !seen1) { f.readStart[1] = b.blk_start[k]; seen1 = true; }
                        f.rStarts[r].push_back((int)(b.blk_start[k] - f.readStart[r]));
                        f.rLens[r].push_back((int)b.blk_len[k]);
                        f.readEnd[r] = b.blk_start[k] + b.blk_len[k];
                    }
                    if (f.readCount == 2) f.readEnd[1] = b.frag_end[i]; else f.readEnd[0] = b.frag_end[i];
                    adapter->ProcessBlocks(f);
                }
            } else {
                uint64_t id = 0;
                if (irf_submit_batch(ctx, &b, &id) != 0 || irf_wait(ctx, id) != 0) return die("irf_submit_batch", irf_last_error(ctx));
            }
            irf_reader_clear_batch(rd);
            return 0;
        };
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), stdin)) > 0) {
            if (irf_reader_feed(rd, buf.data(), n, bgzf ? 1 : 0) != 0) return die("BAM", irf_reader_last_error(rd));
            if (int rc = drain(false)) return rc;
        }
        if (irf_reader_finish(rd) != 0) return die("BAM", irf_reader_last_error(rd));
        if (int rc = drain(true)) return rc;
        if (!mapped) return die("BAM", "no header");
        if (adapter) {
            adapter->finish();
            adapter->writeOutputs(pos[0]);
        } else {
            if (irf_finalize(ctx) != 0) return die("irf_finalize", irf_last_error(ctx));
            if (irf_write_outputs(ctx, pos[0]) != 0) return die("irf_write_outputs", irf_last_error(ctx));
        }
    } catch (const std::exception& e) {
        return die("error", e.what());
    }
    irf_reader_stats st;
    irf_reader_get_stats(rd, &st);
    print_summary(st);
    irf_reader_destroy(rd);
    if (adapter) delete adapter; else irf_destroy(ctx);
    return 0;
}
