/*
 * irf.h — C ABI of the MI355X-native IRFinder read-block counting engine.
 *
 * This is the drop-in boundary (SURVEY.md §8 b).  Upstream IRFinder has no FFI
 * for this path: the boundary there is the C++ interface
 *
 *     class ReadBlockProcessor {
 *         virtual void ProcessBlocks(const FragmentBlocks&) = 0;
 *         virtual void ChrMapUpdate(const std::vector<std::string>&) = 0;
 *     };
 *
 * (⟨upstream⟩ src/irfinder/ReadBlockProcessor.h — the source is NOT in
 * /root/reference, which holds only README.md:1-7; every "replaces" note
 * below therefore cites the upstream file by name, without line numbers.
 * See SURVEY.md §0.)  The C++ adapter that implements that interface on top
 * of this ABI is irfinder_amd/host/hip_block_processor.h.
 *
 * Conventions: every function returns 0 on success, a negative irf_status on
 * failure; no exception crosses the ABI; irf_last_error() gives the text.
 * A context is thread-compatible (one host thread per context), not thread-safe.
 * Coordinates are 0-based half-open (BAM).  Strand index: [0] = neg, [1] = pos.
 */
#ifndef IRF_H
#define IRF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IRF_ABI_VERSION 1

typedef enum irf_status {
    IRF_OK = 0,
    IRF_E_ARG = -1,      /* bad argument / NULL / inconsistent sizes          */
    IRF_E_PARSE = -2,    /* malformed reference file or BAM stream             */
    IRF_E_RANGE = -3,    /* value outside what the packed device layout holds  */
    IRF_E_STATE = -4,    /* call out of order (e.g. submit before chr map)     */
    IRF_E_DEVICE = -5,   /* HIP runtime error / no device                      */
    IRF_E_NOMEM = -6,
    IRF_E_IO = -7
} irf_status;

/* Every recalled-behaviour uncertainty of SURVEY.md App. B.5 (OQ-n) is one
 * field here; defaults (irf_params_default) are the survey's stated guesses. */
typedef struct irf_params {
    uint32_t struct_size;           /* = sizeof(irf_params), ABI check          */
    /* ---- reader (BAM2blocks) ---- */
    uint32_t skip_flag_mask;        /* OQ-2  records with any of these flag bits are skipped (0x4) */
    uint32_t pair_long_inclusive;   /* OQ-1  0: Long iff e1 <  pos2 ; 1: e1 <= pos2               */
    uint32_t pair_short_inclusive;  /* OQ-1  1: Short iff e1 >= e2  ; 0: e1 >  e2                 */
    /* ---- counters ---- */
    uint32_t junc_min_flank;        /* OQ-4  both flanking blocks >= this (5)                      */
    uint32_t sp_overhang_left;      /* SpansPoint::setSpanLength(5,4)                              */
    uint32_t sp_overhang_right;
    uint32_t sp_inclusive;          /* OQ-6  0: bs+L < p < be-R ; 1: bs+L <= p <= be-R             */
    /* ---- per-intron statistics ---- */
    uint32_t trim_centre_pct;       /* OQ-8  trimmed mean keeps the centre 40 %                    */
    uint32_t pct_interpolate;       /* OQ-7  1: (N+1)p/100 with linear interpolation; 0: no interp */
    uint32_t flank_offset;          /* OQ-9  first/last window starts 5 nt inside the intron       */
    uint32_t flank_len;             /* OQ-9  ... and is 50 nt long                                 */
    uint32_t flank_use_measured;    /* OQ-9  1: window ∩ measured blocks ; 0: raw coordinates      */
    /* ---- directionality vote ---- */
    uint32_t dir_ratio;             /* OQ-5  same > ratio*opp => +1, opp > ratio*same => -1 (10)   */
    /* ---- output ---- */
    uint32_t out_precision;         /* OQ-13 significant digits of doubles in the text files (6)   */
    /* ---- warnings (OQ-12) ---- */
    double warn_lowcover_min;       /* LowCover   if SpliceExact + IntronDepth < 10                */
    double warn_lowsplice_min;      /* LowSplicing if SpliceExact < 4                              */
    double warn_minor_ratio;        /* MinorIsoform if SpliceExact*1.33 < max(SpliceL,SpliceR)     */
    double warn_nonuniform_rel;     /* NonUniformIntronCover if max flank > depth*1.5 + 4          */
    double warn_nonuniform_abs;
    /* ---- engine sizing (no upstream counterpart) ---- */
    uint64_t junc_event_capacity;   /* device junction-event buffer, in events, before compaction  */
    uint32_t keep_depth;            /* 1: keep whole-chromosome depth after finalize (FragmentsMap)*/
    /* ---- reader (no upstream counterpart: SURVEY App. D-1) ---- */
    uint32_t mate_join;             /* 0: upstream's pairing, the NEXT record is the mate (name-collated input);
                                       1: coordinate-sorted input, mates are joined by read name            */
    /* ---- further OQ switches (appended: ABI order of the fields above is unchanged) ---- */
    uint32_t warn_nonuniform_and;   /* OQ-12 0: NonUniformIntronCover iff max flank > depth*rel + abs (one mixed tolerance);
                                             1: iff max flank > depth + abs AND max flank > depth*rel (two conditions)   */
    uint32_t gap_pair_is_intersect; /* OQ-3  an overlapping pair whose read 2 starts in an N gap of read 1 is counted as two
                                             reads either way; the stdout summary lists it under 0: long pairs, 1: intersect pairs */
} irf_params;

void irf_params_default(irf_params* p);

/* ------------------------------------------------------------------------ */
/* Reference tables.  Replaces JunctionCount::loadRef, SpansPoint::loadRef,
 * FragmentsInROI::loadRef (⟨upstream⟩ ReadBlockProcessor.cpp) and
 * CoverageBlocks::loadRef (⟨upstream⟩ ReadBlockProcessor_CoverageBlocks.cpp). */
typedef struct irf_ref irf_ref;

int irf_ref_load_files(const char* ref_cover_bed, const char* ref_sj_ref,
                       const char* ref_read_continues_ref, const char* ref_roi_bed,
                       irf_ref** out, char* errbuf, size_t errbuf_len);
/* Same, from memory.  Order of bufs: cover, sj, read-continues, ROI.        */
int irf_ref_load_mem(const char* const bufs[4], const size_t lens[4],
                     irf_ref** out, char* errbuf, size_t errbuf_len);
void irf_ref_free(irf_ref* ref);
uint32_t irf_ref_n_intron(const irf_ref* ref);
uint32_t irf_ref_n_junction(const irf_ref* ref);
uint32_t irf_ref_n_point(const irf_ref* ref);
uint32_t irf_ref_n_roi(const irf_ref* ref);

/* ------------------------------------------------------------------------ */
/* One batch of fragments, structure-of-arrays.  One entry of the frag_*
 * arrays is one upstream FragmentBlocks (⟨upstream⟩ FragmentBlocks.h) as it
 * is handed to ProcessBlocks; blocks of fragment i are
 * [frag_blk_off[i], frag_blk_off[i+1]) of the blk_* arrays, read 0 first.    */
typedef struct irf_batch {
    uint32_t n_frag;
    uint32_t n_blk;
    const uint32_t* frag_chr;      /* BAM refID (FragmentBlocks::chr_id)                     */
    const uint8_t*  frag_flags;    /* bit0 direction (1 = +), bit1 readCount == 2            */
    const uint32_t* frag_start;    /* readStart[0]                                           */
    const uint32_t* frag_end;      /* readCount==2 ? max(readEnd[0],readEnd[1]) : readEnd[0] */
    const uint32_t* frag_blk_off;  /* n_frag + 1 entries                                     */
    const uint32_t* blk_start;     /* absolute: readStart[r] + rStarts[r][j]                 */
    const uint32_t* blk_len;       /* rLens[r][j]                                            */
    const uint8_t*  blk_read;      /* r: 0 or 1 (junctions are only within one read)         */
    uint32_t mem_kind;             /* IRF_MEM_HOST or IRF_MEM_DEVICE                         */
    uint32_t reserved;
} irf_batch;

#define IRF_MEM_HOST   0u   /* pageable or pinned host memory (pinned => async copy) */
#define IRF_MEM_DEVICE 1u   /* already resident on the context's device              */

#define IRF_FRAG_DIR   0x1u
#define IRF_FRAG_TWO   0x2u

typedef enum irf_warning {
    IRF_WARN_NONE = 0,          /* "-" */
    IRF_WARN_LOWCOVER = 1,
    IRF_WARN_LOWSPLICING = 2,
    IRF_WARN_MINORISOFORM = 3,
    IRF_WARN_NONUNIFORM = 4
} irf_warning;

/* One IRFinder-IR-{nondir,dir}.txt table as columns; rows are the records of
 * ref-cover.bed in file order.  Replaces the per-row computation of
 * CoverageBlocksIRFinder::WriteOutput (⟨upstream⟩
 * ReadBlockProcessor_CoverageBlocks.cpp).                                    */
typedef struct irf_ir_table {
    const double* coverage;
    const double* intron_depth;
    const double* p25;
    const double* p50;
    const double* p75;
    const double* first50;
    const double* last50;
    const double* irratio;
    const uint32_t* sp_left;       /* ExonToIntronReadsLeft  */
    const uint32_t* sp_right;      /* ExonToIntronReadsRight */
    const uint32_t* splice_left;
    const uint32_t* splice_right;
    const uint32_t* splice_exact;
    const uint8_t*  warning;       /* irf_warning */
} irf_ir_table;

typedef struct irf_results {
    int32_t  directionality;       /* JunctionCount::Directional(): -1, 0, +1               */
    uint32_t n_chr;                /* BAM references                                         */
    const uint64_t* chr_cnt;       /* [n_chr][2]  FragmentsInChr, (neg,pos)                  */
    uint32_t n_roi;
    const uint64_t* roi_cnt;       /* [2][n_roi]  FragmentsInROI, ref-ROI.bed file order     */
    uint32_t n_sp;
    const uint32_t* sp_chr;        /* chromosome slot (irf_chr_name)                          */
    const uint32_t* sp_pos;
    const uint32_t* sp_cnt;        /* [2][n_sp]   SpansPoint, sorted by (slot,pos)           */
    uint64_t n_junc;               /* known ∪ observed junctions, sorted (slot,left,right)   */
    const uint32_t* junc_chr;
    const uint32_t* junc_left;
    const uint32_t* junc_right;
    const uint32_t* junc_cnt;      /* [2][n_junc]                                            */
    const uint8_t*  junc_refdir;   /* 0 unknown, 1 '-', 2 '+', 3 both                        */
    uint32_t n_intron;
    uint32_t has_dir;              /* dir table valid (directionality != 0)                  */
    irf_ir_table nondir;
    irf_ir_table dir;
    uint64_t n_fragments;          /* fragments submitted                                    */
    uint64_t n_blocks;
    uint64_t n_junction_events;
} irf_results;

/* ------------------------------------------------------------------------ */
typedef struct irf_ctx irf_ctx;

/* Builds the engine for one device.  `ref` is copied; it may be freed after. */
int irf_create(const irf_params* params, const irf_ref* ref, int device, irf_ctx** out);
void irf_destroy(irf_ctx* ctx);
const char* irf_last_error(const irf_ctx* ctx);

/* Replaces ReadBlockProcessor::ChrMapUpdate for all five processors at once.
 * `lens` are the l_ref values of the BAM header (the depth arrays need them).
 * `owned` (may be NULL = all) marks the references this context counts:
 * fragments on other references are ignored (chromosome sharding, §8 e).    */
int irf_chr_map_update(irf_ctx* ctx, uint32_t n_chr, const char* const* names,
                       const uint32_t* lens, const uint8_t* owned);

/* Multi-GPU, a reference split between contexts (no upstream counterpart: upstream is one process; SURVEY.md §8 e).
 * As irf_chr_map_update, plus for every OWNED reference the part [own_start[i], own_end[i]) this context owns (NULL, NULL:
 * whole references; an end beyond the reference's length = to its end).  The cut must lie where no intron (with its
 * measured blocks) spans it: an intergenic gap (ROIs may span it — ref-ROI.bed lists the intergenic regions themselves).  Rules:
 *   - introns, SpansPoints and ref-sj junctions belong to the context that owns their START (left end, position);
 *     the others do not exist for this context (their rows / counters stay zero, so the shards' tables still add up);
 *   - a fragment is counted by every context whose part it reaches into (so a fragment across the cut adds its depth and
 *     its junctions on both sides, no exchange needed); fragments elsewhere on the reference are ignored;
 *   - FragmentsInChr and FragmentsInROI count a fragment where its start lies (every context holds all ROIs of the
 *     reference); a novel junction is reported by the owner of its left end
 *     (irf_export_junctions / irf_get_results; the packed IRF_COL_JUNC_* columns are refused for such a context).
 * irf_gather and the RCCL recipe work unchanged.                                                                     */
int irf_chr_map_update_ranges(irf_ctx* ctx, uint32_t n_chr, const char* const* names, const uint32_t* lens, const uint8_t* owned,
                              const uint32_t* own_start, const uint32_t* own_end);

/* Replaces n_frag calls of ReadBlockProcessor::ProcessBlocks on each of
 * FragmentsInChr, FragmentsInROI, JunctionCount, SpansPoint, CoverageBlocks
 * and FragmentsMap.  Asynchronous: host arrays must stay valid and unmodified
 * until irf_wait(batch_id) returns.                                          */
int irf_submit_batch(irf_ctx* ctx, const irf_batch* batch, uint64_t* batch_id);
int irf_wait(irf_ctx* ctx, uint64_t batch_id);

/* End of input: scan depth, per-intron statistics, junction sort-reduce,
 * directionality vote, IR tables.  Results stay on the device until
 * irf_get_results.  Replaces the end-of-run half of ⟨upstream⟩ irfinder.cpp
 * main() up to (not including) the file writers.
 * Returns when the input errors of the sample are known (it waits for the count
 * kernels) and all of its own work is QUEUED on the context's stream; it does not
 * wait for the statistics kernels.  Every call that reads a result waits first
 * (irf_get_results, irf_write_outputs, irf_get_depth, irf_get_timing, an
 * irf_export_device without IRF_COL_NO_SYNC, irf_finalize_directional), and so do
 * irf_sync and irf_reset; exports with IRF_COL_NO_SYNC queue behind it.  A caller
 * that counts sample after sample can thus prepare the next submission while this
 * sample's statistics kernels still run.                                        */
int irf_finalize(irf_ctx* ctx);
/* Multi-GPU (no upstream counterpart; upstream is one process): JunctionCount::Directional() is a vote over ALL
 * known junctions, so with chromosome shards the votes {same, opposite} of every rank are summed by the host
 * (one all-reduce of two integers), the decision (same > ratio*opp => +1, opp > ratio*same => -1, else 0) is taken
 * once and imposed on every rank; the directional table is recomputed where the local decision differed.        */
int irf_get_votes(irf_ctx* ctx, uint64_t votes[2]);
int irf_finalize_directional(irf_ctx* ctx, int32_t directionality);
/* Host view of the results (pointers owned by ctx, valid until reset/destroy). */
int irf_get_results(irf_ctx* ctx, irf_results* out);
/* Chromosome slot -> name.  Slots 0..n_chr-1 are the BAM references, further
 * slots are chromosomes that occur only in the reference tables.            */
const char* irf_chr_name(const irf_ctx* ctx, uint32_t slot);
uint32_t irf_n_chr_slots(const irf_ctx* ctx);

/* Writes IRFinder-IR-nondir.txt, IRFinder-IR-dir.txt (if directional),
 * IRFinder-JuncCount.txt, IRFinder-SpansPoint.txt, IRFinder-ChrCoverage.txt,
 * IRFinder-ROI.txt into outdir.  Replaces the writers of ⟨upstream⟩
 * irfinder.cpp main() and the WriteOutput methods.                           */
int irf_write_outputs(irf_ctx* ctx, const char* outdir);

/* FragmentsMap (north_star; SURVEY OQ-14): per-base depth of strand `dir`
 * (0 neg, 1 pos, 2 both) over [start,end) of BAM reference `chr`, copied to
 * host `out` (end-start uint32).  Needs params.keep_depth = 1.               */
int irf_get_depth(irf_ctx* ctx, uint32_t chr, uint32_t start, uint32_t end,
                  uint32_t dir, uint32_t* out);

/* Zero all counters for the next sample; reference tables and chr map stay. */
int irf_reset(irf_ctx* ctx);

/* Device-resident result export for the multi-GPU gather (§8 e): copies one
 * named column into caller-owned DEVICE memory (no host round trip).  `what`
 * is an irf_column; returns the element count in *n (call with dst == NULL
 * to query the size).                                                        */
typedef enum irf_column {
    IRF_COL_IR_NONDIR_F64 = 0,   /* [8][n_intron] doubles, order of irf_ir_table   */
    IRF_COL_IR_NONDIR_U32 = 1,   /* [6][n_intron] uint32 (5 counts + warning)      */
    IRF_COL_IR_DIR_F64 = 2,
    IRF_COL_IR_DIR_U32 = 3,
    IRF_COL_JUNC_KEYS = 4,       /* NOVEL junctions: uint64 packed keys, sorted    */
    IRF_COL_JUNC_COUNTS = 5,     /* ... and their uint32 counts                    */
    IRF_COL_SP_CNT = 6,          /* [2][n_sp] uint32                               */
    IRF_COL_ROI_CNT = 7,         /* [2][n_roi] uint32                              */
    IRF_COL_CHR_CNT = 8,         /* [n_chr][2] uint32                              */
    IRF_COL_KNOWN_CNT = 9        /* [2][n_ref_junction] uint32, ref-sj rows sorted by (slot,left,right) */
} irf_column;
int irf_export_device(irf_ctx* ctx, uint32_t what, void* dst_device, uint64_t* n);
/* IR tables of a context that does not own every BAM reference leave irf_export_device with the rows of introns on
 * references it does not own set to zero, so that the shards' columns can simply be added (RCCL reduce(SUM), or
 * irf_import_device).  Introns on no BAM reference at all belong to no shard: OR this flag into `what` on ONE rank
 * (the root) to keep its rows for them (all-zero depth, LowCover — what a single-GPU run reports).                  */
#define IRF_COL_KEEP_ORPHANS 0x100u
/* `dst_device` of irf_export_device may also be PAGE-LOCKED HOST memory (irf_host_alloc): the column then crosses PCIe
 * once, without a device-side copy (not for the IR tables of a sharded context).  With IRF_COL_NO_SYNC OR-ed into `what`
 * the copy is only queued: call irf_sync before reading `dst` (several columns, one wait).  The queued copies of the IR
 * tables of an unsharded context run on a stream of their own: irf_reset / irf_submit_batch of the NEXT sample may be
 * called before irf_sync, and its count kernel then runs while the tables cross PCIe (the library orders the next
 * write of its tables behind the copy; give every sample in flight its own `dst`).                                    */
#define IRF_COL_NO_SYNC 0x200u

/* ------------------------------------------------------------------------ */
/* Multi-GPU collection for C / C++ hosts (SURVEY.md §8 e; no upstream counterpart: upstream is one process).
 *
 * One process per GPU (north_star): every rank finalizes its chromosomes, then
 *   1. irf_get_votes -> ncclAllReduce(sum) of the two integers -> irf_finalize_directional on every rank;
 *   2. per fixed-shape column (IRF_COL_IR_*, SP_CNT, ROI_CNT, CHR_CNT, KNOWN_CNT): irf_export_device into a device
 *      buffer, ncclReduce(sum) to the root (or ncclSend / ncclRecv + irf_import_device on the root), root: irf_import_device
 *      of the reduced peers' sum — or, simpler, reduce straight into a buffer and import it once;
 *   3. novel junctions (their packed keys depend on the shard's layout): irf_export_junctions on every rank, the rows
 *      sent as plain uint32 arrays (ncclSend / ncclRecv of 5 x n words), irf_import_junctions on the root;
 *   4. irf_get_results / irf_write_outputs on the root give the whole sample's tables and files.
 * One process driving several contexts (threads, or several contexts on one GPU): irf_gather does all four steps with
 * peer copies.  INTEGRATION.md §5 shows both.                                                                      */

/* Adds a column exported by another shard (already on this context's device) into this context's table.  `what` is
 * one of the fixed-shape columns; n must equal this context's element count.                                      */
int irf_import_device(irf_ctx* ctx, uint32_t what, const void* src_device, uint64_t n);
/* Novel junctions of this context as coordinates, sorted by (BAM refID, left, right), host arrays.  Call with all
 * arrays NULL to get the row count in *n; else *n is the capacity on entry and the row count on return.             */
int irf_export_junctions(irf_ctx* ctx, uint64_t* n, uint32_t* chr, uint32_t* left, uint32_t* right, uint32_t* cnt_neg, uint32_t* cnt_pos);
int irf_import_junctions(irf_ctx* ctx, uint64_t n, const uint32_t* chr, const uint32_t* left, const uint32_t* right,
                         const uint32_t* cnt_neg, const uint32_t* cnt_pos);
/* In-process collection over n finalized contexts with the same reference tables and chromosome map and disjoint
 * `owned` masks: global strandedness vote, then every table of the others is added into ctxs[root].                */
int irf_gather(irf_ctx* const* ctxs, uint32_t n, uint32_t root);

/* HBM layout facts for roofline accounting (no upstream counterpart).       */
typedef struct irf_layout {
    uint64_t plane_positions;   /* positions per strand plane of the difference array (compact unless keep_depth) */
    uint64_t full_positions;    /* size of the full coordinate space (owned references)                          */
    uint64_t measured_positions;/* sum of measured bases over the introns this context finalises                 */
    uint64_t junction_capacity; /* novel-junction event buffer, in events                                        */
} irf_layout;
int irf_get_layout(irf_ctx* ctx, irf_layout* out);

/* The order the fragments of the batches arrive in (a performance hint: every result is a commutative sum, the tables are
 * the same either way — tests/test_gpu_parity.py, P1).  IRF_ORDER_SORTED (default): coordinate-sorted, the order the count
 * kernel's per-chunk window of the depth planes is built for.  IRF_ORDER_ANY: no order (a name-collated BAM, upstream's
 * own input order; what irf_dreader_* delivers without params.mate_join): every batch is binned by coordinate on the
 * device (radix passes over (chromosome base + start) / 256, then a gather of the columns) before it is counted —
 * about 2.5x faster than counting scattered fragments.  No upstream counterpart (upstream's counters are hash maps on
 * the host; order does not matter to them).                                                                            */
#define IRF_ORDER_SORTED 0u
#define IRF_ORDER_ANY 1u
int irf_set_input_order(irf_ctx* ctx, uint32_t order);

/* Time (ms) the device spent in the named stage during the last
 * submit/finalize cycle, measured with HIP events on the engine's stream.   */
typedef struct irf_timing {
    float count_ms;      /* sum over batches of the fused count kernel       */
    float scan_ms;       /* known-junction depth events + scan of the depth carries                             */
    float stats_ms;      /* the per-intron statistics kernels                                                   */
    float junction_ms;   /* known-junction lookups + vote, then the sort-reduce of the novel junctions: the
                            latter runs on its own stream NEXT TO the statistics kernels, so this interval
                            overlaps stats_ms (the stages' sum is more than total_finalize_ms)                  */
    float table_ms;      /* the junction-dependent columns of the IR rows, written when both have finished      */
    float total_finalize_ms;
    uint32_t n_count_launches;
    uint32_t n_radix_passes;   /* 8-bit digit passes the novel-junction sort ran (digits equal in every key are skipped) */
    uint64_t n_sort_records;   /* (key, weight) records the novel-junction radix sort-reduce ran over (the count kernel
                                  merges a chunk's events on one junction before it appends them)                        */
} irf_timing;
/* Kernel times of the last finalized sample (waits for it).  Between irf_reset and the next irf_finalize: of the
 * sample before the reset, so that a loop over samples can read them without an extra wait of its own.            */
int irf_get_timing(irf_ctx* ctx, irf_timing* out);

/* Page-locked host memory for batch arrays (async H2D copies need it).  No
 * upstream counterpart: upstream never leaves the host.                     */
int irf_host_alloc(size_t bytes, void** out);
void irf_host_free(void* p);

/* ------------------------------------------------------------------------ */
/* Host reader.  Replaces BAM2blocks (⟨upstream⟩ BAM2blocks.cpp:
 * readBamHeader, processAll, processPair, processSingle, cigar2block) for
 * callers that do not bring their own.  Parses an uncompressed BAM stream
 * (what `gzip -cd x.bam` yields) or BGZF, pairs adjacent records, assembles
 * fragments and appends them to growable SoA arrays.                        */
typedef struct irf_reader irf_reader;

typedef struct irf_reader_stats {
    uint64_t n_records, n_nucleotides;
    uint64_t n_singles, n_pairs, n_pairs_short, n_pairs_intersect, n_pairs_long;
    uint64_t n_skipped, n_error;
} irf_reader_stats;

int irf_reader_create(const irf_params* params, irf_reader** out);
void irf_reader_destroy(irf_reader* r);
const char* irf_reader_last_error(const irf_reader* r);
/* Feed bytes (any chunking).  is_bgzf: 1 = BGZF-compressed, 0 = raw BAM.  Errors are sticky.  After an error the
 * batch and the counters hold the records in front of the damaged member / record (the same for any thread count); the
 * device reader (irf_dreader_feed) refuses the whole feed instead — discard the partial results either way.          */
int irf_reader_feed(irf_reader* r, const uint8_t* data, size_t len, int is_bgzf);
/* BGZF members of one irf_reader_feed call are inflated and CRC-checked by n_threads threads (1..64; default 1).
 * No upstream counterpart: ⟨upstream⟩ bin/IRFinder pipes `gzip -cd` (one thread) into the engine.  The record
 * parser stays sequential (mate pairing is), so feed in large pieces (MiB) for the threads to have work.        */
int irf_reader_set_threads(irf_reader* r, int n_threads);
/* Header (valid once parsed): returns n_ref, or a negative status if the
 * header is not complete yet.                                                */
int irf_reader_header(const irf_reader* r, const char* const** names, const uint32_t** lens);
/* Fragments assembled so far, as a host batch (pointers owned by the reader,
 * invalidated by the next feed/clear).                                       */
int irf_reader_batch(const irf_reader* r, irf_batch* out);
void irf_reader_clear_batch(irf_reader* r);
int irf_reader_finish(irf_reader* r);       /* EOF: flags a dangling mate     */
int irf_reader_get_stats(const irf_reader* r, irf_reader_stats* out);

/* Pure functions of the reader, exposed for parity tests.                   */
/* cigar2block: BAM cigar words -> relative block starts/lengths.            */
int irf_cigar2block(const uint32_t* cigar, uint32_t n_cigar,
                    uint32_t* starts, uint32_t* lens, uint32_t cap,
                    uint32_t* n_blocks, uint32_t* genome_len);

/* ------------------------------------------------------------------------ */
/* Device reader (SURVEY.md §8 f, row N1).  Replaces the `gzip -cd sample.bam |`
 * stage of ⟨upstream⟩ bin/IRFinder together with BAM2blocks (⟨upstream⟩
 * BAM2blocks.cpp: processAll, processPair, processSingle, cigar2block): the
 * caller feeds COMPRESSED BGZF bytes (host memory, any chunking); members are
 * inflated and CRC-checked, records located, mates paired and fragments
 * assembled on the device.  The result of a feed is a DEVICE-resident
 * irf_batch — equal, column by column, to what irf_reader_* yields for the
 * same bytes — that goes straight into irf_submit_batch of a context on the
 * same device.  params.mate_join = 0: upstream's pairing, the next record is the
 * mate (name-collated input; the fragments then arrive in no coordinate order:
 * see irf_set_input_order); 1: coordinate-sorted input, mates are joined by read
 * name on the device (records whose mate has not arrived wait for the next feed).
 * The BAM header must be complete before the first batch appears.
 * The batch of a feed stays valid until the next-but-one feed (two slots): call
 * irf_sync on the consuming context before feeding twice more.                */
typedef struct irf_dreader irf_dreader;

typedef struct irf_dreader_timing {     /* sums over all feeds, HIP events on the reader's stream */
    float h2d_ms;         /* compressed bytes + member table to the device                  */
    float inflate_ms;     /* bgzf_inflate_kernel                                              */
    float crc_ms;         /* bgzf_crc_kernel                                                  */
    float chain_ms;       /* record starts: guess, verify rounds, scan (includes host syncs)  */
    float assemble_ms;    /* record index, pairing state, count pass, scans, write pass       */
    uint32_t chain_rounds;/* most verification rounds a feed's record chain needed (2-3 on real data) */
    uint64_t bytes_in;        /* compressed bytes consumed                                    */
    uint64_t bytes_inflated;
    uint64_t n_members;
} irf_dreader_timing;

int irf_dreader_create(const irf_params* params, int device, irf_dreader** out);
void irf_dreader_destroy(irf_dreader* r);
const char* irf_dreader_last_error(const irf_dreader* r);
int irf_dreader_feed(irf_dreader* r, const uint8_t* bgzf, size_t len);
/* Pipeline (off by default; set before the first feed, else IRF_E_STATE).  On: a
 * feed copies and inflates ITS bytes and then parses the bytes of the feed
 * BEFORE it, so the host-to-device copy of one feed overlaps the inflate of the
 * previous one.  irf_dreader_batch after feed k then holds the fragments of
 * feed k - 1 (none after the first), irf_dreader_finish parses the last feed,
 * and its fragments are the batch after finish; errors of a feed surface one
 * call later.  Same fragments in the same order as with the pipeline off.  The
 * bytes passed to a feed may be reused as soon as it returns, in both modes.
 * No upstream counterpart (upstream reads a pipe synchronously).              */
int irf_dreader_set_pipeline(irf_dreader* r, int on);
/* BGZF members the device inflates at once (one lane per member: CUs x resident
 * workgroups x 64; 49152 on an MI355X), 0 on error.  The inflate kernel of a
 * feed takes about the same time for any number of members up to this and
 * twice as long just above it, so feeds sized to a little less than a multiple
 * of it are the efficient ones (irf_dreader_timing.n_members / bytes_in gives
 * the members per byte seen so far).  No upstream counterpart.                */
uint32_t irf_dreader_member_slots(const irf_dreader* r);
int irf_dreader_header(const irf_dreader* r, const char* const** names, const uint32_t** lens);
/* Fragments of the LAST feed (possibly none), device-resident.               */
int irf_dreader_batch(const irf_dreader* r, irf_batch* out);
int irf_dreader_finish(irf_dreader* r);      /* EOF: truncated member / record, dangling mate */
int irf_dreader_get_stats(const irf_dreader* r, irf_reader_stats* out);
int irf_dreader_get_timing(const irf_dreader* r, irf_dreader_timing* out);

/* Waits until everything submitted to the context has been processed (its
 * compute stream is idle): device-resident batches may then be overwritten.
 * No upstream counterpart (upstream is synchronous).                         */
int irf_sync(irf_ctx* ctx);
/* Device -> host copy for callers without a HIP runtime binding of their own
 * (parity tests read the columns of a device-resident batch with it).         */
int irf_copy_to_host(void* dst_host, const void* src_device, size_t bytes);
/* Pure function of the device reader, for parity tests: inflates a buffer of
 * whole BGZF members (CRC-checked) on the device into host memory.
 * IRF_E_PARSE: malformed / corrupt member; IRF_E_RANGE: out_cap too small
 * (*out_len is set to the size needed).                                       */
int irf_gpu_bgzf_inflate(int device, const uint8_t* bgzf, size_t len, uint8_t* out, size_t out_cap, size_t* out_len);

/* Variant of irf_gpu_bgzf_inflate that allocates nothing.  It has no counterpart in the reference: it exists for
 * callers that cannot call hipMalloc (tight serving loops, phases in which the caller's allocator is frozen, harnesses
 * that forbid allocation).  It is NOT itself capturable into a HIP graph: it copies synchronously on the null stream
 * and reads a status word back.  Same members in (host memory), same checks and status codes, but the inflated bytes
 * are left in CALLER-owned DEVICE memory `out_device` (out_cap >= inflated size + 64: the inflate kernel stores whole
 * words), and the device staging the other entry point allocates per call (compressed bytes, member table, per-member
 * status) lives in a file-scope device array of the library (9 MiB per GPU, held until the process ends).  Limits:
 * len <= IRF_NOALLOC_MAX_BYTES and at most IRF_NOALLOC_MAX_MEMBERS members, else IRF_E_RANGE (*out_len is still the size
 * needed when only out_cap is too small).  Calls no hipMalloc / hipFree.  Every call on a GPU shares that GPU's array,
 * which keeps the last call's contents; calls from several host threads on one GPU are serialised by a per-device
 * mutex held for the whole (synchronous) call, so they queue rather than race.  Synchronous (returns after the CRC check).               */
#define IRF_NOALLOC_MAX_BYTES   (8u << 20)
#define IRF_NOALLOC_MAX_MEMBERS 8192u
int irf_gpu_bgzf_inflate_noalloc(int device, const uint8_t* bgzf, size_t len, uint8_t* out_device, size_t out_cap, size_t* out_len);

#ifdef __cplusplus
}
#endif
#endif /* IRF_H */
