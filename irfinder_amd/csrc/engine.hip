// engine.hip — irf_ctx: device state, batch staging on HIP streams, the
// finalize pipeline and the C ABI of include/irf.h.
//
// Stands where ⟨upstream⟩ irfinder.cpp main() wires five ReadBlockProcessor
// objects to BAM2blocks: here one context owns all counter state in HBM and
// every ProcessBlocks call becomes one row of a structure-of-arrays batch.
//
// HBM layout (per context; G = sum over owned BAM references of len+1 rounded
// up to 64):
//   diff        int32 [2 strands][G]      difference events (+1 block start, -1 block end)
//   carry       uint32 [2][G/256]         depth just before each 256-position segment
//   junc_keys   uint64 [capacity] x2      packed junction events + sort buffer
//   sp_pos/cnt, roi_*, chr_*              reference tables and their counters
//   intron tables                         global coordinates of measured blocks
//   staging     2 x batch arrays          double buffer for host batches
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <queue>
#include <vector>

#include "../../include/irf.h"
#include "../host/ref_tables.h"
#include "kernels.h"

struct irf_ref {
    irf::RefTables t;
};

namespace {

using irf::RefTables;
namespace dev = irf::dev;

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(IRF_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;      // elements
    bool owned = true;   // false: a view into another buffer (irf_ctx: the counter pool)
    ~DevBuf() { release(); }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; cap = 0; owned = true; }
    void view(void* base, size_t n) { release(); p = static_cast<T*>(base); cap = n; owned = false; }
    void swap(DevBuf& o) { std::swap(p, o.p); std::swap(cap, o.cap); std::swap(owned, o.owned); }
    // grows (never shrinks); contents are NOT preserved
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        release();
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
};

struct Staging {
    DevBuf<uint32_t> frag_chr, frag_start, frag_end, frag_blk_off, blk_start, blk_len;
    DevBuf<uint8_t> frag_flags, blk_read;
    hipEvent_t copied = nullptr;     // H2D copies of the batch in this slot are done
    hipEvent_t consumed = nullptr;   // count kernel that reads this slot is done
    uint64_t batch_id = 0;
    bool in_use = false;
};

}   // namespace

struct irf_ctx {
    irf_params P;
    RefTables ref;
    int device = 0;
    hipStream_t stream = nullptr, copy_stream = nullptr;
    // IRF_COL_NO_SYNC exports of the IR tables run on their own stream: the tables of sample k cross PCIe while the count
    // kernel of sample k+1 runs (whatever writes the tables next waits for them: order_after_exports)
    hipStream_t export_stream = nullptr;
    hipEvent_t ev_tables_ready = nullptr, ev_export_done = nullptr;
    bool exports_pending = false;
    // IR-table exports that were only queued (IRF_COL_NO_SYNC) are HELD BACK until the next sample's count kernel has been
    // launched (or until anything waits for them or writes the tables): while a device-to-host copy kernel is in flight, no
    // other kernel completes before it does (seen in the kernel trace), so copies issued right away stood between the
    // small kernels in front of the next count kernel and cost the step their whole duration.
    struct HeldExport { void* dst; const void* src; size_t bytes; };
    std::vector<HeldExport> held_exports;
    int flush_exports() {
        if (held_exports.empty()) return 0;
        HIP_TRY(hipStreamWaitEvent(export_stream, ev_tables_ready, 0));
        for (const HeldExport& h : held_exports) HIP_TRY(hipMemcpyAsync(h.dst, h.src, h.bytes, hipMemcpyDefault, export_stream));
        HIP_TRY(hipEventRecord(ev_export_done, export_stream));
        held_exports.clear();
        return 0;
    }
    int order_after_exports() {
        if (int rc = flush_exports()) return rc;
        if (exports_pending) { HIP_TRY(hipStreamWaitEvent(stream, ev_export_done, 0)); exports_pending = false; }
        return 0;
    }
    hipStream_t junc_stream = nullptr;           // the novel-junction sort of irf_finalize, next to the statistics kernels
    hipEvent_t ev_scalars = nullptr;             // irf_finalize: the scalars and error flags have reached h_scal
    unsigned long long* h_scal = nullptr;        // page-locked: d_scalars[0..8), then the error flags
    std::string err;
    bool have_map = false, finalized = false, results_ready = false;
    bool final_pending = false;                  // irf_finalize has queued its work; nobody has waited for it yet (settle())
    bool gathered = false;                       // irf_gather added the peers' columns to this context's (once per sample)
    bool kj_depth_applied = false;               // the known-junction difference events of this sample are in the planes (finalize may be called again after a failure)

    // ---- chromosome map ----
    uint32_t n_chr = 0;                          // BAM references
    std::vector<std::string> slot_names;         // BAM references, then reference-only names (sorted)
    std::vector<uint32_t> chr_len;
    std::vector<uint8_t> chr_owned;
    // an owned reference may be owned in part: features (by their start / left / position) and FragmentsInChr (by the
    // fragment's start) inside [own_start, own_end) are this context's; irf_chr_map_update_ranges, DESIGN.md §6
    std::vector<uint32_t> own_start, own_end;
    bool any_split = false;
    bool in_own(uint32_t sl, uint32_t pos) const { return pos >= own_start[sl] && pos < own_end[sl]; }
    std::vector<uint64_t> chr_base;              // per BAM reference, offset inside a plane
    std::vector<uint32_t> refchr_slot;           // reference chr id -> slot
    uint64_t plane_stride = 0;                   // positions per strand plane (compact unless keep_depth)
    uint64_t full_size = 0;                      // full coordinate space: sum of owned (len+1, rounded to 64)
    std::vector<uint32_t> pmap;                  // full -> plane, per 512-bucket; empty = identity
    DevBuf<uint32_t> d_pmap;

    // ---- device state ----
    DevBuf<int32_t> d_diff;
    DevBuf<uint32_t> d_seg_sum;                  // [2][plane_stride/256] sum of each segment of d_diff (count kernel)
    // A second set of planes for the NEXT sample: cleared on a side stream once this sample's statistics kernels are done
    // (while the host fetches the tables the GPU has nothing else to do), so irf_reset swaps instead of clearing in line.
    // irf_set_input_order(IRF_ORDER_ANY): every batch is binned by coordinate before it is counted (reorder.hip)
    uint32_t input_order = 0;
    DevBuf<uint64_t> ro_key_a, ro_key_b;
    DevBuf<uint32_t> ro_nblk, ro_chr, ro_start, ro_end, ro_off, ro_bs, ro_bl;
    DevBuf<uint8_t> ro_flags, ro_rd, ro_frag_recs;
    int reorder(dev::BatchDev& B);
    DevBuf<uint8_t> d_dirty;                     // which 64-byte lines of d_diff hold something (kernels.h: CountTables.dirty)
    // Spare plane sets for the following samples.  A set that irf_reset swapped out is dirty (its line marks say where) and is
    // cleared on the side stream behind the statistics kernels of the sample that follows; with TWO spare sets the next
    // sample takes the one cleared a whole sample ago and never waits for the clear that was just queued (which then
    // runs next to the table export and the start of the next count kernel).  The second set is allocated with the first
    // one (behind the first sample's statistics) if the device has plenty of room for it.
    struct PlaneSet {
        DevBuf<int32_t> diff;
        DevBuf<uint32_t> seg_sum;
        DevBuf<uint8_t> dirty;
        hipEvent_t cleared = nullptr;
        bool clean = false;                      // a clear is queued (or done): `cleared` says when
        uint64_t seq = 0;                        // order in which the clears were queued
        bool allocated(uint64_t stride) const { return stride && diff.cap >= 2 * stride && seg_sum.cap >= 2 * stride / dev::kDepthSegment && dirty.cap >= dev::dirty_map_bytes(stride); }
        void release() { diff.release(); seg_sum.release(); dirty.release(); clean = false; }
    } spare[2];
    uint64_t clear_seq = 0;
    hipStream_t clear_stream = nullptr;
    hipEvent_t ev_stats_done = nullptr;
    DevBuf<uint32_t> d_carry;                    // [2][plane_stride/256] depth before each segment (after finalize)
    DevBuf<uint64_t> d_chr_base;
    DevBuf<uint32_t> d_chr_len;
    DevBuf<uint8_t> d_chr_owned;
    DevBuf<uint2> d_chr_own;                     // [n_chr] own_start, own_end (read by the count kernel for references owned in part)
    DevBuf<uint32_t> d_chr_cnt;
    DevBuf<uint32_t> d_roi_idx, d_roi_cnt;
    DevBuf<uint4> d_roi_bkt;                     // kernels.h: CountTables::roi_bkt
    DevBuf<uint4> d_chr_rec;                     // kernels.h: CountTables::chr_rec
    DevBuf<uint64_t> d_roi_gstart, d_roi_gend;
    uint32_t n_roi_dev = 0;
    size_t n_sp_dev = 0;                         // SpansPoint entries on owned chromosomes (device tables)
    DevBuf<uint32_t> d_sp_row, d_sp_cnt;
    DevBuf<uint4> d_cell_s, d_cell_e;            // kernels.h: cell tables (plane coordinate + SpansPoint rank per 32-position cell)
    DevBuf<uint64_t> d_sp_gpos;                  // SpansPoint: the points' full coordinates, sorted (input of the cell tables)
    DevBuf<uint4> d_jhash;                       // kernels.h: CountTables::jhash
    DevBuf<unsigned long long> d_chunk_origin;   // kernels.h: CountTables::chunk_origin (sized to the largest batch so far)
    DevBuf<dev::CountTables> d_ctab;             // the count kernel's table block (kernels.h: launch_count) ...
    DevBuf<dev::IntronTables> d_itab;            // the statistics kernels' table blocks: [0] one table, [1] both (irf_finalize queues both shapes), [2] irf_finalize_directional
    dev::CountTables ctab_shadow;                // ... and what it holds
    uint32_t jh_mask = 0;
    DevBuf<uint64_t> d_junc_a, d_junc_b;          // novel-junction records: packed keys (+ sort buffer)
    DevBuf<uint32_t> d_juncw_a, d_juncw_b;        // ... their weights; d_juncw_a.cap follows d_junc_a.cap
    DevBuf<uint32_t> d_wex;                       // exclusive scan of the sorted weights
    DevBuf<unsigned long long> d_scalars;        // [0] junc_count, [1] key OR, [2] key AND, [3],[4] dir votes
    DevBuf<uint32_t> d_err;
    DevBuf<uint8_t> d_counter_pool;              // d_scalars, d_err, d_chr_cnt, d_roi_cnt, d_sp_cnt, d_known_cnt are views into it
    size_t counter_pool_bytes = 0;
    DevBuf<uint32_t> d_ws;                       // workspace of the carry scan
    DevBuf<uint32_t> d_ws_sort;                  // workspace of the junction radix sort / compaction
    DevBuf<uint32_t> d_flags, d_uidx, d_ustart, d_ucount;
    DevBuf<uint64_t> d_ukeys;
    uint64_t n_unique = 0, n_events = 0;
    const uint64_t* d_sorted = nullptr;
    // intron tables
    DevBuf<dev::IntronRec> d_in_recs;              // one record per intron, in launch order (size classes)
    DevBuf<dev::BlkRec> d_blk_recs;
    dev::StatsClasses stats_classes;
    DevBuf<uint32_t> d_stats_wg_off[2];
    DevBuf<uint64_t> d_us_pos, d_ue_pos, d_known_keys;
    DevBuf<uint8_t> d_known_dir;
    DevBuf<int32_t> d_known_us_idx, d_known_ue_idx; // per known key: position of its ends among the intron starts / ends
    std::vector<uint64_t> known_keys_h;
    DevBuf<uint32_t> d_known_kjidx, d_known_cnt;   // known-junction rows and their [2][n_kj] counters
    uint32_t n_kj = 0;
    unsigned long long votes[2] = {0, 0};
    uint64_t n_junction_events_total = 0;
    uint64_t measured_valid = 0;
    DevBuf<uint32_t> d_left_cnt, d_right_cnt;
    uint32_t n_us = 0, n_ue = 0, n_known_dev = 0;
    DevBuf<double> d_out_f64[2];
    DevBuf<uint32_t> d_out_u32[2];
    // multi-GPU collection (kernels.h: kRowForeign / kRowMine / kRowOrphan per IR row)
    DevBuf<uint8_t> d_row_code;
    bool sharded = false;                          // some BAM reference is not owned by this context
    bool ir_prepared[2] = {false, false};          // foreign rows of the IR table zeroed for an import
    DevBuf<double> d_tmp_f64;                      // masked copy of an IR table for the export
    DevBuf<uint32_t> d_tmp_u32;
    struct ImpJunc { uint32_t chr, left, right, c0, c1; };
    std::vector<ImpJunc> imp_junc;                 // novel junctions imported from other shards (irf_import_junctions)

    // host-side permutations of the reference tables
    uint32_t n_sp = 0, n_roi = 0, n_intron = 0;
    std::vector<uint32_t> sp_slot, sp_pos_sorted;          // (slot,pos) order = device order
    std::vector<uint32_t> kj_slot, kj_left, kj_right;      // known junctions sorted by (slot,left,right)
    std::vector<uint8_t> kj_dir;

    // ---- batches ----
    Staging stage[2];
    uint64_t next_batch_id = 1;
    uint64_t n_fragments = 0, n_blocks = 0, junc_upper = 0;
    int32_t directionality = 0;

    // ---- timing ----
    // Timing events of a sample.  Two sets, used alternately (irf_reset switches): the events of the sample that just ended
    // stay readable while the next sample records its own, so that irf_get_timing can work the times out (a dozen
    // hipEventElapsedTime calls: not free) when it is asked, not between two samples while the device waits.
    struct EvSet {
        std::vector<std::pair<hipEvent_t, hipEvent_t>> count_events;
        size_t count_events_used = 0;
        hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        hipEvent_t junc_done = nullptr;
        bool finalized = false, timed = false;   // irf_finalize recorded its events / `timing` holds the times read off them
        irf_timing timing;
    } evset[2];
    int cur_set = 0;
    void collect_timing(EvSet& E);

    // ---- host results ----
    std::vector<uint64_t> h_chr_cnt, h_roi_cnt;
    std::vector<uint32_t> h_sp_chr, h_sp_pos, h_sp_cnt;
    std::vector<uint32_t> h_j_chr, h_j_left, h_j_right, h_j_cnt;
    std::vector<uint8_t> h_j_refdir;
    std::vector<double> h_f64[2];
    std::vector<uint32_t> h_u32[2];
    std::vector<uint8_t> h_warn[2];

    int fail(int code, const std::string& msg) { err = msg; return code; }

    int init();
    void free_events();
    int build_map(uint32_t n, const char* const* names, const uint32_t* lens, const uint8_t* owned, const uint32_t* r_start, const uint32_t* r_end);
    int zero_counters();
    dev::CountTables count_tables();
    int ensure_junc_capacity(uint64_t need);
    int submit(const irf_batch* b, uint64_t* id);
    int wait(uint64_t id);
    void launch_stats(int t, int dirn, bool dual = false);
    dev::IntronTables intron_tables(int t, int dirn, bool dual);
    int finalize();
    int finalize_directional(int32_t dirn);
    int settle();
    int queue_spare_clears();
    int fetch_results();
    int export_col(uint32_t what, void* dst, uint64_t* n_out);
    int import_col(uint32_t what, const void* src, uint64_t n);
    int own_novel_junctions(std::vector<ImpJunc>& out);
};

namespace {

template <class T>
int upload(irf_ctx* c, DevBuf<T>& buf, const std::vector<T>& v) {
    hipError_t e = buf.reserve(v.size());
    if (e != hipSuccess) return c->fail(IRF_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    if (!v.empty()) {
        e = hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) return c->fail(IRF_E_DEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e));
        e = hipStreamSynchronize(c->stream);      // v may be a temporary
        if (e != hipSuccess) return c->fail(IRF_E_DEVICE, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    return 0;
}

template <class T>
int alloc_zero(irf_ctx* c, DevBuf<T>& buf, size_t n) {
    hipError_t e = buf.reserve(n);
    if (e != hipSuccess) return c->fail(IRF_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e));
    if (n) {
        e = hipMemsetAsync(buf.p, 0, n * sizeof(T), c->stream);
        if (e != hipSuccess) return c->fail(IRF_E_DEVICE, std::string("hipMemset: ") + hipGetErrorString(e));
    }
    return 0;
}

#define TRY(expr) do { int rc_ = (expr); if (rc_ != 0) return rc_; } while (0)

// Diagnostic switches (kernel ablations, tuning knobs) exist only in builds with -DIRF_DIAG (tools/build_variant.sh):
// the shipped library never reads the environment on the counting path, so a stray variable cannot change a table.
#ifdef IRF_DIAG
uint64_t env_u64(const char* name, uint64_t dflt) {
    const char* v = getenv(name);
    return v && *v ? strtoull(v, nullptr, 10) : dflt;
}
#else
inline uint64_t env_u64(const char*, uint64_t dflt) { return dflt; }
#endif

}   // namespace

int irf_ctx::init() {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    {
        // the sort's many small kernels run while the statistics kernels fill the device: highest priority, so that they
        // get the next free slot
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&junc_stream, hipStreamNonBlocking, greatest));
        HIP_TRY(hipStreamCreateWithPriority(&clear_stream, hipStreamNonBlocking, least));      // (yields to the export's copy kernels)
        for (auto& E : evset) HIP_TRY(hipEventCreate(&E.junc_done));
        HIP_TRY(hipEventCreateWithFlags(&ev_scalars, hipEventDisableTiming));
        HIP_TRY(hipHostMalloc((void**)&h_scal, 16 * sizeof(unsigned long long), hipHostMallocDefault));
        memset(h_scal, 0, 16 * sizeof(unsigned long long));
        // The queued exports of the IR tables use the junction stream: it is idle from the end of one sample's sort to the
        // next sample's, and a fifth stream would share a hardware queue with one of the others (four per process: the
        // copies then hold back the next sample's count kernel, seen in the kernel trace).
        export_stream = junc_stream;
        HIP_TRY(hipEventCreateWithFlags(&ev_tables_ready, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_export_done, hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&ev_stats_done, hipEventDisableTiming));
    for (auto& sp : spare) HIP_TRY(hipEventCreateWithFlags(&sp.cleared, hipEventDisableTiming));
    for (auto& s : stage) {
        HIP_TRY(hipEventCreateWithFlags(&s.copied, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.consumed, hipEventDisableTiming));
    }
    for (auto& E : evset) {
        for (auto& e : E.ev) HIP_TRY(hipEventCreate(&e));
        memset(&E.timing, 0, sizeof E.timing);
    }
    return 0;
}

void irf_ctx::free_events() {
    for (auto& s : stage) {
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.consumed) (void)hipEventDestroy(s.consumed);
        s.copied = s.consumed = nullptr;
    }
    for (auto& E : evset) {
        for (auto& e : E.ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        for (auto& p : E.count_events) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
        E.count_events.clear();
        if (E.junc_done) { (void)hipEventDestroy(E.junc_done); E.junc_done = nullptr; }
    }
    if (ev_stats_done) { (void)hipEventDestroy(ev_stats_done); ev_stats_done = nullptr; }
    for (auto& sp : spare) if (sp.cleared) { (void)hipEventDestroy(sp.cleared); sp.cleared = nullptr; }
    if (clear_stream) { (void)hipStreamDestroy(clear_stream); clear_stream = nullptr; }
    if (stream) { (void)hipStreamDestroy(stream); stream = nullptr; }
    if (copy_stream) { (void)hipStreamDestroy(copy_stream); copy_stream = nullptr; }
    if (ev_scalars) { (void)hipEventDestroy(ev_scalars); ev_scalars = nullptr; }
    if (h_scal) { (void)hipHostFree(h_scal); h_scal = nullptr; }
    export_stream = nullptr;         // (an alias of the junction stream)
    if (ev_tables_ready) { (void)hipEventDestroy(ev_tables_ready); ev_tables_ready = nullptr; }
    if (ev_export_done) { (void)hipEventDestroy(ev_export_done); ev_export_done = nullptr; }
    if (junc_stream) { (void)hipStreamDestroy(junc_stream); junc_stream = nullptr; }
}

// Replaces ChrMapUpdate of all processors: binds reference-table chromosomes to
// BAM refIDs and lays every table out in device order.
int irf_ctx::build_map(uint32_t n, const char* const* names, const uint32_t* lens, const uint8_t* owned, const uint32_t* r_start, const uint32_t* r_end) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipStreamSynchronize(copy_stream));
    TRY(settle());
    // exports of the last sample may still be reading the tables that are reallocated below
    TRY(flush_exports());
    if (export_stream) HIP_TRY(hipStreamSynchronize(export_stream));
    exports_pending = false;
    n_chr = n;
    slot_names.clear();
    chr_len.assign(lens, lens + n);
    chr_owned.assign(n, 1);
    if (owned) chr_owned.assign(owned, owned + n);
    own_start.assign(n, 0u);
    own_end.assign(n, 0xffffffffu);
    any_split = false;
    for (uint32_t i = 0; i < n && r_start && r_end; i++) {
        if (!chr_owned[i]) continue;
        if (r_start[i] >= r_end[i]) return fail(IRF_E_ARG, "irf_chr_map_update_ranges: empty or reversed range of an owned reference");
        own_start[i] = r_start[i];
        own_end[i] = r_end[i] > lens[i] ? 0xffffffffu : r_end[i];        // (an end beyond the reference = to its end)
        if (own_start[i] != 0u || own_end[i] != 0xffffffffu) any_split = true;
    }
    std::map<std::string, uint32_t> slot_of;
    for (uint32_t i = 0; i < n; i++) {
        slot_names.emplace_back(names[i]);
        slot_of.emplace(slot_names.back(), i);        // first occurrence wins
    }
    {
        std::vector<std::string> extra;
        for (auto& nm : ref.chr_names) if (!slot_of.count(nm)) extra.push_back(nm);
        std::sort(extra.begin(), extra.end());
        for (auto& nm : extra) { slot_of.emplace(nm, (uint32_t)slot_names.size()); slot_names.push_back(nm); }
    }
    refchr_slot.resize(ref.chr_names.size());
    for (size_t i = 0; i < ref.chr_names.size(); i++) refchr_slot[i] = slot_of[ref.chr_names[i]];

    // planes: owned references only
    chr_base.assign(n, 0);
    uint64_t off = 0;
    for (uint32_t i = 0; i < n; i++) {
        chr_base[i] = off;
        if (chr_owned[i]) off += (((uint64_t)chr_len[i] + 1 + 63) / 64) * 64;
    }
    full_size = ((off + dev::kScanTile - 1) / dev::kScanTile) * dev::kScanTile;
    if (full_size >= dev::kJuncMaxGlobal) return fail(IRF_E_RANGE, "genome too large for the packed junction key (>= 2^34 bases)");
    // which introns can be measured at all (chromosome in the BAM header, owned, coordinates inside it)
    n_intron = (uint32_t)ref.n_intron();
    std::vector<uint8_t> in_valid(n_intron);
    for (uint32_t i = 0; i < n_intron; i++) {
        uint32_t sl = refchr_slot[ref.in_chr[i]];
        bool ok = sl < n && chr_owned[sl] && in_own(sl, ref.in_start[i]) && ref.in_end[i] <= chr_len[sl] && ref.in_start[i] <= ref.in_end[i];
        for (uint32_t b = ref.in_blk_off[i]; ok && b < ref.in_blk_off[i + 1]; b++)
            if (ref.blk_end[b] > chr_len[sl] || ref.blk_end[b] < ref.blk_start[b]) ok = false;
        in_valid[i] = ok;
    }
    measured_valid = 0;
    for (uint32_t i = 0; i < n_intron; i++) if (in_valid[i]) measured_valid += ref.in_measured[i];
    // plane layout: the whole coordinate space (keep_depth: FragmentsMap), or only the 512-position buckets that hold
    // measured intron bases (kernels.h: plane_coord)
    pmap.clear();
    plane_stride = full_size;
    if (!P.keep_depth) {
        const uint64_t nbk = (full_size >> dev::kBucketShift) + 2;
        std::vector<uint8_t> keep(nbk, 0);
        auto mark = [&](uint64_t g0, uint64_t g1) {          // [g0, g1)
            if (g0 >= g1) return;
            for (uint64_t b = g0 >> dev::kBucketShift; b <= (g1 - 1) >> dev::kBucketShift; b++) keep[b] = 1;
        };
        for (uint32_t i = 0; i < n_intron; i++) {
            if (!in_valid[i]) continue;
            const uint64_t base = chr_base[refchr_slot[ref.in_chr[i]]];
            for (uint32_t b = ref.in_blk_off[i]; b < ref.in_blk_off[i + 1]; b++) mark(base + ref.blk_start[b], base + ref.blk_end[b]);
            if (!P.flank_use_measured) {
                const uint64_t st = ref.in_start[i], en = ref.in_end[i], il = en - st, fo = P.flank_offset, fl = P.flank_len;
                mark(base + st + std::min(fo, il), base + st + std::min(fo + fl, il));
                mark(base + en - std::min(fo + fl, il), base + en - std::min(fo, il));
            }
        }
        pmap.resize(nbk);
        uint64_t kept = 0;
        for (uint64_t b = 0; b < nbk; b++) { pmap[b] = (uint32_t)((kept << 1) | keep[b]); kept += keep[b]; }
        if (kept >= (1ull << 30)) { pmap.clear(); }          // does not fit the map encoding: fall back to full planes
        else plane_stride = (((kept << dev::kBucketShift) + 1 + dev::kScanTile - 1) / dev::kScanTile) * dev::kScanTile;
    }
    TRY(upload(this, d_pmap, pmap));
    TRY(upload(this, d_chr_base, chr_base));
    TRY(upload(this, d_chr_len, chr_len));
    TRY(upload(this, d_chr_owned, chr_owned));
    {
        std::vector<uint4> rec(std::max<uint32_t>(n, 1u), make_uint4(0u, 0u, 0u, 0u));
        for (uint32_t i = 0; i < n; i++) rec[i] = make_uint4((uint32_t)chr_base[i], (uint32_t)(chr_base[i] >> 32), chr_len[i],
                                                             !chr_owned[i] ? 0u : ((own_start[i] != 0u || own_end[i] != 0xffffffffu) ? 2u : 1u));
        TRY(upload(this, d_chr_rec, rec));
        std::vector<uint2> own(std::max<uint32_t>(n, 1u), make_uint2(0u, 0xffffffffu));
        for (uint32_t i = 0; i < n; i++) own[i] = make_uint2(own_start[i], own_end[i]);
        TRY(upload(this, d_chr_own, own));
    }
    {
        if (clear_stream) HIP_TRY(hipStreamSynchronize(clear_stream));
        for (auto& sp : spare) sp.release();      // (a new map: the spare planes are allocated again when first needed)
        hipError_t e = d_diff.reserve(2 * plane_stride);
        if (e == hipSuccess) e = d_dirty.reserve(dev::dirty_map_bytes(plane_stride));
        if (e != hipSuccess) return fail(IRF_E_NOMEM, "hipMalloc of the depth planes failed");
    }
    {
        const uint64_t n_seg = 2 * plane_stride / dev::kDepthSegment;
        hipError_t e = d_carry.reserve(n_seg);
        if (e == hipSuccess) e = d_seg_sum.reserve(n_seg);
        if (e == hipSuccess) e = d_ws.reserve(dev::scan_workspace_elems(n_seg));
        if (e != hipSuccess) return fail(IRF_E_NOMEM, "hipMalloc of the depth carries / scan workspace failed");
    }

    // ---- ROI: owned chromosomes only, sorted by (slot,end) = by end in full coordinates; bucket index on the end ----
    n_roi = (uint32_t)ref.roi_start.size();
    {
        std::vector<uint32_t> idx;
        for (uint32_t i = 0; i < n_roi; i++) {
            const uint32_t sl = refchr_slot[ref.roi_chr[i]];
            if (sl < n && chr_owned[sl] && ref.roi_start[i] <= ref.roi_end[i] && ref.roi_end[i] <= chr_len[sl]) idx.push_back(i);
        }
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
            uint32_t sa = refchr_slot[ref.roi_chr[a]], sb = refchr_slot[ref.roi_chr[b]];
            if (sa != sb) return sa < sb;
            return ref.roi_end[a] < ref.roi_end[b];
        });
        std::vector<uint64_t> gs, ge;
        std::vector<uint32_t> fi;
        for (uint32_t i : idx) {
            const uint64_t base = chr_base[refchr_slot[ref.roi_chr[i]]];
            gs.push_back(base + ref.roi_start[i]); ge.push_back(base + ref.roi_end[i]); fi.push_back(i);
        }
        n_roi_dev = (uint32_t)fi.size();
        if (fi.size() >= 0x0fffffffull) return fail(IRF_E_RANGE, "too many ROIs");
        // per bucket: the first ROI that ends at or after the bucket's start, with its extent inline (kernels.h: roi_bkt)
        const uint64_t nb = (full_size >> dev::kBucketShift) + 2;
        std::vector<uint4> bucket(nb);
        size_t p = 0;
        for (uint64_t b = 0; b < nb; b++) {
            while (p < ge.size() && ge[p] < (b << dev::kBucketShift)) p++;
            if (p < ge.size()) bucket[b] = make_uint4((uint32_t)gs[p], (uint32_t)ge[p], (uint32_t)p | (uint32_t)((gs[p] >> 32) & 3u) << 28 | (uint32_t)((ge[p] >> 32) & 3u) << 30, fi[p]);
            else bucket[b] = make_uint4(0u, 0u, 0x0fffffffu, 0u);
        }
        TRY(upload(this, d_roi_gstart, gs));
        TRY(upload(this, d_roi_gend, ge));
        TRY(upload(this, d_roi_idx, fi));
        TRY(upload(this, d_roi_bkt, bucket));
    }
    // ---- SpansPoint: (slot,pos)-sorted; reference-only chromosomes at the tail ----
    n_sp = (uint32_t)ref.sp_pos.size();
    if (P.sp_overhang_left > (1u << 20) || P.sp_overhang_right > (1u << 20)) return fail(IRF_E_ARG, "SpansPoint overhang beyond 2^20");
    n_sp_dev = 0;
    {
        std::vector<uint64_t> key(n_sp);
        for (uint32_t i = 0; i < n_sp; i++) key[i] = ((uint64_t)refchr_slot[ref.sp_chr[i]] << 32) | ref.sp_pos[i];
        std::sort(key.begin(), key.end());
        sp_slot.resize(n_sp);
        sp_pos_sorted.resize(n_sp);
        std::vector<uint64_t> gpos;
        std::vector<uint32_t> row;
        for (uint32_t i = 0; i < n_sp; i++) {
            sp_slot[i] = (uint32_t)(key[i] >> 32);
            sp_pos_sorted[i] = (uint32_t)key[i];
            const uint32_t sl = sp_slot[i];
            if (sl < n && chr_owned[sl] && in_own(sl, sp_pos_sorted[i]) && sp_pos_sorted[i] <= chr_len[sl]) { gpos.push_back(chr_base[sl] + sp_pos_sorted[i]); row.push_back(i); }
        }
        n_sp_dev = gpos.size();
        if (n_sp_dev >= 0xffffffffull) return fail(IRF_E_RANGE, "too many SpansPoint positions");
        TRY(upload(this, d_sp_row, row));
        // the cell tables: plane word + point rank / bits per 32-position cell, built on the device
        {
            std::vector<uint64_t> gp(gpos);
            if (gp.empty()) gp.push_back(0);
            TRY(upload(this, d_sp_gpos, gp));
            const uint64_t n_cells = (full_size >> dev::kCellShift) + 2;
            hipError_t e = d_cell_s.reserve(n_cells);
            if (e == hipSuccess) e = d_cell_e.reserve(n_cells);
            if (e != hipSuccess) return fail(IRF_E_NOMEM, "hipMalloc of the cell tables failed");
            const int32_t lo_add = (int32_t)(P.sp_overhang_left + (P.sp_inclusive ? 0u : 1u));
            const int32_t hi_sub = (int32_t)(P.sp_overhang_right + (P.sp_inclusive ? 0u : 1u));
            dev::launch_build_cells(d_sp_gpos.p, d_sp_row.p, (uint32_t)n_sp_dev, pmap.empty() ? nullptr : d_pmap.p, n_cells, lo_add, 1 - hi_sub, d_cell_s.p,
                                    d_cell_e.p, stream);
            HIP_TRY(hipGetLastError());
        }
    }
    // ---- known junctions sorted by (slot,left,right) ----
    {
        size_t nk = ref.kj_left.size();
        std::vector<uint32_t> idx(nk);
        for (size_t i = 0; i < nk; i++) idx[i] = (uint32_t)i;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) {
            return refchr_slot[ref.kj_chr[a]] < refchr_slot[ref.kj_chr[b]];
        });
        kj_slot.clear(); kj_left.clear(); kj_right.clear(); kj_dir.clear();
        std::vector<uint64_t> kkeys;
        std::vector<uint8_t> kdir;
        std::vector<uint32_t> krow;
        for (uint32_t i : idx) {
            uint32_t sl = refchr_slot[ref.kj_chr[i]];
            kj_slot.push_back(sl); kj_left.push_back(ref.kj_left[i]); kj_right.push_back(ref.kj_right[i]); kj_dir.push_back(ref.kj_dir[i]);
            if (sl < n && chr_owned[sl] && in_own(sl, ref.kj_left[i]) && ref.kj_right[i] >= ref.kj_left[i]) {
                uint64_t span = (uint64_t)ref.kj_right[i] - ref.kj_left[i];
                if (span < (1ull << dev::kJuncSpanBits)) {
                    kkeys.push_back(((chr_base[sl] + ref.kj_left[i]) << dev::kJuncSpanBits) | span);
                    kdir.push_back(ref.kj_dir[i]);
                    krow.push_back((uint32_t)kj_left.size() - 1);
                }
            }
        }
        n_known_dev = (uint32_t)kkeys.size();
        n_kj = (uint32_t)kj_left.size();
        {
            // the count kernel's junction table: open addressing on junction_hash, load <= 1/32 (kernels.h: jhash)
            uint64_t slots = 1024;
            while (slots < 32ull * kkeys.size()) slots <<= 1;
            while (slots > (1ull << 27) && slots > 4ull * kkeys.size()) slots >>= 1;
            if (slots > (1ull << 31)) return fail(IRF_E_RANGE, "too many known junctions");
            jh_mask = (uint32_t)(slots - 1);
            std::vector<uint4> tab(slots, make_uint4(0xffffffffu, 0xffffffffu, 0u, 0u));
            for (size_t k = 0; k < kkeys.size(); k++) {
                if (k && kkeys[k] == kkeys[k - 1]) continue;          // a junction listed twice in ref-sj.ref: its first row counts
                const uint32_t lo = (uint32_t)kkeys[k], hi = (uint32_t)(kkeys[k] >> 32);
                uint32_t sl = dev::junction_hash(lo, hi) & jh_mask;
                while ((tab[sl].x & tab[sl].y) != 0xffffffffu) sl = (sl + 1u) & jh_mask;
                tab[sl] = make_uint4(lo, hi, krow[k], 0u);
            }
            TRY(upload(this, d_jhash, tab));
        }
        known_keys_h = kkeys;
        TRY(upload(this, d_known_keys, kkeys));
        TRY(upload(this, d_known_dir, kdir));
        TRY(upload(this, d_known_kjidx, krow));
    }
    // ---- introns ----
    {
        std::vector<uint8_t> valid(in_valid), plus(n_intron);
        std::vector<uint64_t> bcs(ref.blk_start.size());
        const uint32_t* pm = pmap.empty() ? nullptr : pmap.data();
        std::vector<uint64_t> gs(n_intron), ge(n_intron), bgs(ref.blk_start.size());
        std::vector<uint32_t> bl(ref.blk_start.size());
        std::vector<int32_t> sis(n_intron, -1), sie(n_intron, -1), usi(n_intron, -1), uei(n_intron, -1), kji(n_intron, -1);
        std::vector<uint64_t> us, ue;
        for (uint32_t i = 0; i < n_intron; i++) {
            uint32_t sl = refchr_slot[ref.in_chr[i]];
            const bool ok = valid[i] != 0;
            plus[i] = ref.in_strand[i] != '-';
            uint64_t base = ok ? chr_base[sl] : 0;
            gs[i] = base + ref.in_start[i];
            ge[i] = base + ref.in_end[i];
            for (uint32_t b = ref.in_blk_off[i]; b < ref.in_blk_off[i + 1]; b++) {
                bgs[b] = base + ref.blk_start[b];
                bcs[b] = ok ? dev::plane_coord(pm, bgs[b]) : bgs[b];
                bl[b] = ref.blk_end[b] >= ref.blk_start[b] ? ref.blk_end[b] - ref.blk_start[b] : 0;
            }
            if (ok) {
                us.push_back(gs[i]); ue.push_back(ge[i]);
                // SpansPoint index of Start / End
                auto find_sp = [&](uint32_t pos) -> int32_t {
                    uint64_t k = ((uint64_t)sl << 32) | pos;
                    size_t lo = 0, hi = n_sp;
                    while (lo < hi) {
                        size_t mid = (lo + hi) / 2;
                        uint64_t km = ((uint64_t)sp_slot[mid] << 32) | sp_pos_sorted[mid];
                        if (km < k) lo = mid + 1; else hi = mid;
                    }
                    if (lo < n_sp && sp_slot[lo] == sl && sp_pos_sorted[lo] == pos) return (int32_t)lo;
                    return -1;
                };
                sis[i] = find_sp(ref.in_start[i]);
                sie[i] = find_sp(ref.in_end[i]);
                // is (Start,End) a known junction?
                size_t lo = 0, hi = kj_left.size();
                const uint32_t st = ref.in_start[i], en = ref.in_end[i];
                while (lo < hi) {
                    size_t mid = (lo + hi) / 2;
                    bool less = kj_slot[mid] != sl ? kj_slot[mid] < sl : (kj_left[mid] != st ? kj_left[mid] < st : kj_right[mid] < en);
                    if (less) lo = mid + 1; else hi = mid;
                }
                if (lo < kj_left.size() && kj_slot[lo] == sl && kj_left[lo] == st && kj_right[lo] == en &&
                    (uint64_t)en - st < (1ull << dev::kJuncSpanBits)) kji[i] = (int32_t)lo;
            }
        }
        std::sort(us.begin(), us.end()); us.erase(std::unique(us.begin(), us.end()), us.end());
        std::sort(ue.begin(), ue.end()); ue.erase(std::unique(ue.begin(), ue.end()), ue.end());
        for (uint32_t i = 0; i < n_intron; i++) if (valid[i]) {
            usi[i] = (int32_t)(std::lower_bound(us.begin(), us.end(), gs[i]) - us.begin());
            uei[i] = (int32_t)(std::lower_bound(ue.begin(), ue.end(), ge[i]) - ue.begin());
        }
        n_us = (uint32_t)us.size(); n_ue = (uint32_t)ue.size();
        {
            std::vector<int32_t> kus(known_keys_h.size(), -1), kue(known_keys_h.size(), -1);
            for (size_t k = 0; k < known_keys_h.size(); k++) {
                const uint64_t gl = known_keys_h[k] >> dev::kJuncSpanBits, gr = gl + (known_keys_h[k] & ((1ull << dev::kJuncSpanBits) - 1ull));
                auto a = std::lower_bound(us.begin(), us.end(), gl);
                if (a != us.end() && *a == gl) kus[k] = (int32_t)(a - us.begin());
                auto b = std::lower_bound(ue.begin(), ue.end(), gr);
                if (b != ue.end() && *b == gr) kue[k] = (int32_t)(b - ue.begin());
            }
            TRY(upload(this, d_known_us_idx, kus)); TRY(upload(this, d_known_ue_idx, kue));
            std::vector<uint64_t>().swap(known_keys_h);
        }
        // one record per intron, laid out by size class and workgroup (kernels.h: StatsClasses): inside a class the
        // introns go, longest first, to the workgroup with the least work so far (cost = segments + a fixed part)
        {
            const uint64_t small_max = P.flank_len <= dev::kStatsSmallFlank ? env_u64("IRF_STATS_SMALL_MAX", dev::kStatsSmallMax) : 0ull;
            int n_cu = 256;
            (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device);
            if (n_cu < 1) n_cu = 1;
            // (five resident 256-thread workgroups per CU, not the six that fit: the sixth slot is what the junction stage's small
            //  kernels run in next to them — with six the stage waits for the class to drain and part of it ends up behind the
            //  statistics: step 4.28 -> 4.16 ms, the statistics themselves 1.79 -> 1.70 ms)
            const uint32_t wgs_per_cu[2] = {(uint32_t)env_u64("IRF_STATS_WGS_BIG", 5), (uint32_t)env_u64("IRF_STATS_WGS_SMALL", 192)};
            std::vector<uint32_t> members[2];
            for (uint32_t i = 0; i < n_intron; i++)
                if (ref.in_measured[i] >= (1ull << 28)) return fail(IRF_E_RANGE, "an intron of ref-cover.bed has 2^28 or more measured bases");
            for (uint32_t i = 0; i < n_intron; i++) members[ref.in_measured[i] <= small_max ? 1 : 0].push_back(i);
            std::vector<dev::IntronRec> recs;
            recs.reserve(n_intron + 1);
            std::vector<uint32_t> wg_off[2];
            for (int c = 0; c < 2; c++) {
                auto& m = members[c];
                std::stable_sort(m.begin(), m.end(), [&](uint32_t a, uint32_t b) { return ref.in_measured[a] > ref.in_measured[b]; });
                // many short lists for the hardware to deal out (the one-wave class: 192 per CU), but about four introns per
                // list at least — a list's first intron pays for the start of the header pipeline (a shard of a reference
                // split over several GPUs has fewer introns than lists) — and never fewer lists than fill the device
                const uint64_t most = (uint64_t)n_cu * std::max(wgs_per_cu[c], 1u);
                const uint64_t fill = std::min<uint64_t>(most, (uint64_t)n_cu * (c == 0 ? 5u : 24u));
                uint32_t n_wg = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(most, std::max<uint64_t>(m.size() / 4, fill)), m.size());
                stats_classes.n_wg[c] = n_wg;
                std::vector<std::vector<uint32_t>> lists(n_wg);
                typedef std::pair<uint64_t, uint32_t> Load;          // (cost so far, workgroup)
                std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
                for (uint32_t w = 0; w < n_wg; w++) heap.push(Load(0, w));
                for (uint32_t i : m) {
                    Load l = heap.top(); heap.pop();
                    lists[l.second].push_back(i);
                    const uint64_t nb = ref.in_blk_off[i + 1] - ref.in_blk_off[i];
                    l.first += ref.in_measured[i] / dev::kDepthSegment + 2 * nb + 8;
                    heap.push(l);
                }
                wg_off[c].assign(1, (uint32_t)recs.size());
                for (uint32_t w = 0; w < n_wg; w++) {
                    for (uint32_t i : lists[w]) {
                        dev::IntronRec r;
                        memset(&r, 0, sizeof r);
                        r.g_start = gs[i]; r.g_end = ge[i]; r.measured = ref.in_measured[i];
                        r.blk_off = ref.in_blk_off[i]; r.n_blk = ref.in_blk_off[i + 1] - ref.in_blk_off[i];
                        r.sp_s = sis[i]; r.sp_e = sie[i]; r.us = usi[i]; r.ue = uei[i]; r.kj = kji[i];
                        r.row = i;
                        uint32_t need = 0, prem = 0;
                        dev::stats_queries(r.measured, P.trim_centre_pct, P.pct_interpolate != 0, r.qrank, &need, &prem);
                        r.flags = (valid[i] ? 1u : 0u) | (plus[i] ? 2u : 0u) | (need << 8);
                        r.prem = prem;
                        recs.push_back(r);
                    }
                    wg_off[c].push_back((uint32_t)recs.size());
                }
            }
            if (recs.empty()) recs.resize(1);
            std::vector<dev::BlkRec> blks(ref.blk_start.size() + 1);      // + one spare entry: the kernel's clamped descriptor load
            for (size_t b = 0; b + 1 < blks.size(); b++) { blks[b].gstart = bgs[b]; blks[b].cstart = bcs[b]; blks[b].len = bl[b]; blks[b].pad = 0; }
            memset(&blks.back(), 0, sizeof(dev::BlkRec));
            TRY(upload(this, d_in_recs, recs)); TRY(upload(this, d_blk_recs, blks));
            TRY(upload(this, d_stats_wg_off[0], wg_off[0])); TRY(upload(this, d_stats_wg_off[1], wg_off[1]));
            stats_classes.wg_off[0] = d_stats_wg_off[0].p; stats_classes.wg_off[1] = d_stats_wg_off[1].p;
        }
        TRY(upload(this, d_us_pos, us)); TRY(upload(this, d_ue_pos, ue));
        TRY(alloc_zero(this, d_left_cnt, (size_t)n_us * 2));
        TRY(alloc_zero(this, d_right_cnt, (size_t)n_ue * 2));
        for (int t = 0; t < 2; t++) {
            TRY(alloc_zero(this, d_out_f64[t], (size_t)n_intron * 8));
            TRY(alloc_zero(this, d_out_u32[t], (size_t)n_intron * 6));
        }
    }
    {
        std::vector<uint8_t> code(std::max<uint32_t>(n_intron, 1u), (uint8_t)dev::kRowOrphan);
        for (uint32_t i = 0; i < n_intron; i++) {
            const uint32_t sl = refchr_slot[ref.in_chr[i]];
            if (sl < n) code[i] = chr_owned[sl] && in_own(sl, ref.in_start[i]) ? (uint8_t)dev::kRowMine : (uint8_t)dev::kRowForeign;
        }
        TRY(upload(this, d_row_code, code));
        sharded = false;
        for (uint32_t i = 0; i < n; i++) if (!chr_owned[i]) sharded = true;
        if (any_split) sharded = true;
    }
    {
        // every counter irf_reset zeroes (except the planes) in ONE allocation: one memset per sample instead of six
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
        const size_t o_scal = take(8 * sizeof(unsigned long long)), o_err = take(sizeof(uint32_t)), o_chr = take((size_t)n_chr * 2 * 4),
                     o_roi = take((size_t)n_roi * 2 * 4), o_sp = take((size_t)n_sp * 2 * 4), o_kj = take((size_t)n_kj * 2 * 4);
        d_scalars.release(); d_err.release(); d_chr_cnt.release(); d_roi_cnt.release(); d_sp_cnt.release(); d_known_cnt.release();
        if (d_counter_pool.reserve(off) != hipSuccess) return fail(IRF_E_NOMEM, "hipMalloc of the counter tables failed");
        counter_pool_bytes = off;
        uint8_t* base = d_counter_pool.p;
        d_scalars.view(base + o_scal, 8); d_err.view(base + o_err, 1); d_chr_cnt.view(base + o_chr, (size_t)n_chr * 2);
        d_roi_cnt.view(base + o_roi, (size_t)n_roi * 2); d_sp_cnt.view(base + o_sp, (size_t)n_sp * 2); d_known_cnt.view(base + o_kj, (size_t)n_kj * 2);
    }
    {
        uint64_t cap = std::max<uint64_t>(P.junc_event_capacity, 1024);
        hipError_t e = d_junc_a.reserve(cap);
        if (e == hipSuccess) e = d_juncw_a.reserve(d_junc_a.cap);
        if (e != hipSuccess) return fail(IRF_E_NOMEM, "hipMalloc of the junction event buffer failed");
    }
    if (d_ctab.reserve(1) != hipSuccess || d_itab.reserve(3) != hipSuccess) return fail(IRF_E_NOMEM, "hipMalloc of the table block failed");
    memset(&ctab_shadow, 0xff, sizeof ctab_shadow);          // (nothing uploaded yet)
    have_map = true;
    return zero_counters();
}

int irf_ctx::zero_counters() {
    HIP_TRY(hipSetDevice(device));
    TRY(settle());                   // (the events of the sample that ends here are recorded again below)
    // (a spare set of planes, cleared behind an earlier sample's statistics kernels, is swapped in; without one the planes
    //  are cleared in line)
    PlaneSet* take = nullptr;        // the spare set whose clear was queued first
    for (auto& sp : spare) if (sp.clean && sp.allocated(plane_stride) && (!take || sp.seq < take->seq)) take = &sp;
    if (take) {
        HIP_TRY(hipStreamWaitEvent(stream, take->cleared, 0));
        d_diff.swap(take->diff);
        d_seg_sum.swap(take->seg_sum);
        d_dirty.swap(take->dirty);
        take->clean = false;         // (it now holds the planes of the sample that just ended: dirty, marks valid)
    } else {
        if (plane_stride) HIP_TRY(hipMemsetAsync(d_diff.p, 0, 2 * plane_stride * sizeof(int32_t), stream));
        if (plane_stride) HIP_TRY(hipMemsetAsync(d_seg_sum.p, 0, 2 * plane_stride / dev::kDepthSegment * sizeof(uint32_t), stream));
        if (plane_stride) HIP_TRY(hipMemsetAsync(d_dirty.p, 0, dev::dirty_map_bytes(plane_stride), stream));
    }
    // chromosome / ROI / SpansPoint / known-junction counters, the scalars and the error flags: one pool
    HIP_TRY(hipMemsetAsync(d_counter_pool.p, 0, counter_pool_bytes, stream));
    n_fragments = n_blocks = junc_upper = 0;
    n_unique = n_events = 0;
    finalized = results_ready = false;
    gathered = false;
    kj_depth_applied = false;
    ir_prepared[0] = ir_prepared[1] = false;
    imp_junc.clear();
    directionality = 0;
    cur_set ^= 1;                    // (the other set's events are those of the sample before the last: long complete)
    EvSet& E = evset[cur_set];
    E.count_events_used = 0;
    E.finalized = E.timed = false;
    memset(&E.timing, 0, sizeof E.timing);
    return 0;
}

dev::CountTables irf_ctx::count_tables() {
    dev::CountTables T;
    memset(&T, 0, sizeof T);
    T.n_chr = n_chr; T.n_roi = n_roi; T.n_sp = n_sp;
    T.junc_min_flank = P.junc_min_flank;
    T.sp_min_len = P.sp_overhang_left + P.sp_overhang_right;
    T.plane_stride = plane_stride; T.pmap = pmap.empty() ? nullptr : d_pmap.p;
    T.junc_capacity = d_junc_a.cap;
    T.chr_rec = d_chr_rec.p; T.chr_own = d_chr_own.p; T.cell_s = d_cell_s.p; T.cell_e = d_cell_e.p; T.jhash = d_jhash.p; T.jh_mask = jh_mask;
    T.diff = d_diff.p; T.seg_sum = d_seg_sum.p; T.dirty = d_dirty.p; T.chr_cnt = d_chr_cnt.p;
    T.roi_bkt = d_roi_bkt.p; T.roi_gstart = d_roi_gstart.p; T.roi_gend = d_roi_gend.p; T.roi_idx = d_roi_idx.p;
    T.n_roi_dev = n_roi_dev; T.roi_cnt = d_roi_cnt.p;
    T.sp_cnt = d_sp_cnt.p;
    T.known_keys = d_known_keys.p; T.known_kjidx = d_known_kjidx.p; T.n_known = n_known_dev; T.n_kj = n_kj; T.known_cnt = d_known_cnt.p;
    T.known_dir = d_known_dir.p; T.dir_votes = d_scalars.p + 3;
    T.junc_keys = d_junc_a.p; T.junc_w = d_juncw_a.p; T.junc_count = d_scalars.p; T.err_flags = d_err.p;
    T.chunk_origin = d_chunk_origin.p;
    return T;
}

// The event buffer must hold every junction a batch can emit (< n_blk).  Uses a host-side upper bound and
// only synchronises to read the true count when the bound is exceeded; grows by doubling (HBM is plentiful).
int irf_ctx::ensure_junc_capacity(uint64_t n_blk) {
    if (junc_upper + n_blk <= d_junc_a.cap) { junc_upper += n_blk; return 0; }
    HIP_TRY(hipStreamSynchronize(stream));
    unsigned long long cur = 0;
    HIP_TRY(hipMemcpy(&cur, d_scalars.p, sizeof cur, hipMemcpyDeviceToHost));
    if (cur > d_junc_a.cap) return fail(IRF_E_RANGE, "junction event buffer overflowed");
    if (cur + n_blk > d_junc_a.cap) {
        uint64_t ncap = std::max<uint64_t>(d_junc_a.cap * 2, cur + n_blk);
        uint64_t* np = nullptr;
        uint32_t* nw = nullptr;
        HIP_TRY(hipMalloc((void**)&np, ncap * sizeof(uint64_t)));
        if (hipMalloc((void**)&nw, ncap * sizeof(uint32_t)) != hipSuccess) { (void)hipFree(np); return fail(IRF_E_NOMEM, "hipMalloc of the junction event buffer failed"); }
        if (cur) {
            hipError_t e1 = hipMemcpy(np, d_junc_a.p, cur * sizeof(uint64_t), hipMemcpyDeviceToDevice);
            hipError_t e2 = hipMemcpy(nw, d_juncw_a.p, cur * sizeof(uint32_t), hipMemcpyDeviceToDevice);
            if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipFree(np); (void)hipFree(nw); return fail(IRF_E_DEVICE, "hipMemcpy of the junction event buffer failed"); }
        }
        d_junc_a.release(); d_juncw_a.release();
        d_junc_a.p = np; d_junc_a.cap = ncap;
        d_juncw_a.p = nw; d_juncw_a.cap = ncap;
    }
    junc_upper = cur + n_blk;
    return 0;
}

int irf_ctx::submit(const irf_batch* b, uint64_t* id) {
    if (!have_map) return fail(IRF_E_STATE, "irf_submit_batch before irf_chr_map_update");
    if (finalized) return fail(IRF_E_STATE, "irf_submit_batch after irf_finalize (call irf_reset first)");
    if (!b) return fail(IRF_E_ARG, "null batch");
    if (b->n_frag && (!b->frag_chr || !b->frag_flags || !b->frag_start || !b->frag_end || !b->frag_blk_off))
        return fail(IRF_E_ARG, "null fragment array");
    if (b->n_blk && (!b->blk_start || !b->blk_len || !b->blk_read)) return fail(IRF_E_ARG, "null block array");
    if (b->mem_kind != IRF_MEM_HOST && b->mem_kind != IRF_MEM_DEVICE) return fail(IRF_E_ARG, "bad mem_kind");
    HIP_TRY(hipSetDevice(device));
    const uint64_t bid = next_batch_id++;
    if (id) *id = bid;
    if (b->n_frag == 0) return 0;
    if (n_fragments + b->n_frag > 0xffffffffull) return fail(IRF_E_RANGE, "more than 2^32-1 fragments in one context (uint32 counters)");
    if (b->mem_kind == IRF_MEM_HOST) {
        // the CSR offsets are the one thing the kernel trusts for addressing: check them on the host
        if (b->frag_blk_off[0] != 0 || b->frag_blk_off[b->n_frag] != b->n_blk) return fail(IRF_E_ARG, "frag_blk_off does not span [0, n_blk]");
        // (a branch-free OR over blocks of 64 K offsets: the compiler vectorises it; an early exit per element kept this
        //  loop at 50 M scalar iterations per sample — the kernel checks the same thing again for device-resident batches)
        const uint32_t* off = b->frag_blk_off;
        for (uint32_t i0 = 0; i0 < b->n_frag; i0 += 65536u) {
            const uint32_t i1 = b->n_frag - i0 < 65536u ? b->n_frag : i0 + 65536u;
            uint32_t bad = 0;
            for (uint32_t i = i0; i < i1; i++) bad |= (uint32_t)(off[i] > off[i + 1]);
            if (bad) return fail(IRF_E_ARG, "frag_blk_off is not non-decreasing");
        }
    }
    TRY(ensure_junc_capacity(b->n_blk));
    dev::BatchDev B;
    memset(&B, 0, sizeof B);
    B.n_frag = b->n_frag; B.n_blk = b->n_blk;
    Staging* st = nullptr;
    if (b->mem_kind == IRF_MEM_DEVICE) {
        // the count kernel reads the columns with 16-byte (uint32 columns) and 4-byte (uint8 columns) vector loads
        auto mis = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) != 0; };
        if (mis(b->frag_chr, 16) || mis(b->frag_start, 16) || mis(b->frag_end, 16) || mis(b->frag_blk_off, 16) || mis(b->frag_flags, 4) ||
            (b->n_blk && (mis(b->blk_start, 16) || mis(b->blk_len, 16) || mis(b->blk_read, 4))))
            return fail(IRF_E_ARG, "device-resident batch columns must be 16-byte aligned (uint32 columns) / 4-byte aligned (uint8 columns)");
        B.frag_chr = b->frag_chr; B.frag_flags = b->frag_flags; B.frag_start = b->frag_start; B.frag_end = b->frag_end;
        B.frag_blk_off = b->frag_blk_off; B.blk_start = b->blk_start; B.blk_len = b->blk_len; B.blk_read = b->blk_read;
    } else {
        st = &stage[bid & 1];
        if (st->in_use) HIP_TRY(hipEventSynchronize(st->consumed));      // the kernel reading this slot has finished
        size_t nf = b->n_frag, nb = std::max<uint32_t>(b->n_blk, 1);
        hipError_t e = hipSuccess;
        if (e == hipSuccess) e = st->frag_chr.reserve(nf);
        if (e == hipSuccess) e = st->frag_flags.reserve(nf);
        if (e == hipSuccess) e = st->frag_start.reserve(nf);
        if (e == hipSuccess) e = st->frag_end.reserve(nf);
        if (e == hipSuccess) e = st->frag_blk_off.reserve(nf + 1);
        if (e == hipSuccess) e = st->blk_start.reserve(nb);
        if (e == hipSuccess) e = st->blk_len.reserve(nb);
        if (e == hipSuccess) e = st->blk_read.reserve(nb);
        if (e != hipSuccess) return fail(IRF_E_NOMEM, "hipMalloc of the staging buffers failed");
        HIP_TRY(hipMemcpyAsync(st->frag_chr.p, b->frag_chr, nf * 4, hipMemcpyHostToDevice, copy_stream));
        HIP_TRY(hipMemcpyAsync(st->frag_flags.p, b->frag_flags, nf, hipMemcpyHostToDevice, copy_stream));
        HIP_TRY(hipMemcpyAsync(st->frag_start.p, b->frag_start, nf * 4, hipMemcpyHostToDevice, copy_stream));
        HIP_TRY(hipMemcpyAsync(st->frag_end.p, b->frag_end, nf * 4, hipMemcpyHostToDevice, copy_stream));
        HIP_TRY(hipMemcpyAsync(st->frag_blk_off.p, b->frag_blk_off, (nf + 1) * 4, hipMemcpyHostToDevice, copy_stream));
        if (b->n_blk) {
            HIP_TRY(hipMemcpyAsync(st->blk_start.p, b->blk_start, (size_t)b->n_blk * 4, hipMemcpyHostToDevice, copy_stream));
            HIP_TRY(hipMemcpyAsync(st->blk_len.p, b->blk_len, (size_t)b->n_blk * 4, hipMemcpyHostToDevice, copy_stream));
            HIP_TRY(hipMemcpyAsync(st->blk_read.p, b->blk_read, (size_t)b->n_blk, hipMemcpyHostToDevice, copy_stream));
        }
        HIP_TRY(hipEventRecord(st->copied, copy_stream));
        HIP_TRY(hipStreamWaitEvent(stream, st->copied, 0));
        st->batch_id = bid; st->in_use = true;
        B.frag_chr = st->frag_chr.p; B.frag_flags = st->frag_flags.p; B.frag_start = st->frag_start.p; B.frag_end = st->frag_end.p;
        B.frag_blk_off = st->frag_blk_off.p; B.blk_start = st->blk_start.p; B.blk_len = st->blk_len.p; B.blk_read = st->blk_read.p;
    }
    if (input_order == IRF_ORDER_ANY && B.n_frag > 1024u) TRY(reorder(B));       // (a batch of a few chunks has nothing to gain)
    {
        // (grown on the stream's own time line: a kernel of an earlier batch may still read the old array)
        const size_t need = (size_t)B.n_frag / dev::count_chunk_fragments() + 2;      // (+ the slice queue word behind the last chunk)
        if (need > d_chunk_origin.cap) {
            HIP_TRY(hipStreamSynchronize(stream));
            if (d_chunk_origin.reserve(need + need / 4) != hipSuccess) return fail(IRF_E_NOMEM, "hipMalloc of the chunk table failed");
        }
    }
    EvSet& E = evset[cur_set];
    if (E.count_events_used == E.count_events.size()) {
        hipEvent_t a, z;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&z));
        E.count_events.emplace_back(a, z);
    }
    auto& evp = E.count_events[E.count_events_used++];
    dev::launch_count_prepare(count_tables(), d_ctab.p, &ctab_shadow, B, stream);
    HIP_TRY(hipEventRecord(evp.first, stream));        // (count_ms = the count kernel itself, as rocprofv3 times it)
    dev::launch_count(d_ctab.p, B, stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(evp.second, stream));
    TRY(flush_exports());            // (the last sample's held-back table copies run under this count kernel)
    if (st) HIP_TRY(hipEventRecord(st->consumed, stream));
    n_fragments += b->n_frag;
    n_blocks += b->n_blk;
    return 0;
}

// B (device-resident columns, any order) -> the same fragments binned by coordinate, in this context's buffers.  Runs on
// the engine's stream in front of the batch's count kernel; the buffers are reused by the next batch (same stream).
int irf_ctx::reorder(dev::BatchDev& B) {
    const size_t n = B.n_frag, nb = std::max<uint32_t>(B.n_blk, 1);
    hipError_t e = ro_key_a.reserve(n);
    if (e == hipSuccess) e = ro_key_b.reserve(n);
    if (e == hipSuccess) e = ro_nblk.reserve(n + 1);
    if (e == hipSuccess) e = ro_off.reserve(n + 4);
    if (e == hipSuccess) e = ro_chr.reserve(n + 4);
    if (e == hipSuccess) e = ro_start.reserve(n + 4);
    if (e == hipSuccess) e = ro_end.reserve(n + 4);
    if (e == hipSuccess) e = ro_flags.reserve(n + 4);
    if (e == hipSuccess) e = ro_bs.reserve(nb + 4);
    if (e == hipSuccess) e = ro_bl.reserve(nb + 4);
    if (e == hipSuccess) e = ro_rd.reserve(nb + 4);
    if (e == hipSuccess) e = ro_frag_recs.reserve(n * dev::reorder_frag_rec_bytes());
    const uint64_t ws_need = std::max(dev::radix_workspace_elems(n), dev::scan_workspace_elems(n + 1));
    if (e == hipSuccess && ws_need > d_ws_sort.cap) e = d_ws_sort.reserve(ws_need);
    if (e != hipSuccess) return fail(IRF_E_NOMEM, "hipMalloc of the batch reorder buffers failed");
    const uint64_t top = full_size >> dev::kReorderShift;        // (the bin sits in bits kReorderKeyBinShift and up)
    if (top >> (64 - dev::kReorderKeyBinShift)) return fail(IRF_E_RANGE, "IRF_ORDER_ANY: the chromosomes of this context span more than 2^34 positions");
    dev::launch_reorder_keys(B, d_chr_base.p, n_chr, ro_key_a.p, ro_frag_recs.p, d_err.p, stream);
    uint64_t* ks = ro_key_a.p; uint64_t* kd = ro_key_b.p;
    for (int shift = 0; shift < 64 - dev::kReorderKeyBinShift && (top >> shift) != 0; shift += 8) {
        dev::launch_radix_pass(ks, kd, nullptr, nullptr, n, dev::kReorderKeyBinShift + shift, d_ws_sort.p, stream);
        std::swap(ks, kd);
    }
    dev::launch_reorder_counts(B.n_frag, ks, ro_frag_recs.p, ro_nblk.p, stream);
    dev::launch_exclusive_scan_u32(ro_nblk.p, ro_off.p, n + 1, d_ws_sort.p, stream);
    dev::launch_reorder_gather(B, ro_frag_recs.p, ks, ro_off.p, ro_chr.p, ro_flags.p, ro_start.p, ro_end.p, ro_bs.p, ro_bl.p, ro_rd.p, stream);
    HIP_TRY(hipGetLastError());
    B.frag_chr = ro_chr.p; B.frag_flags = ro_flags.p; B.frag_start = ro_start.p; B.frag_end = ro_end.p; B.frag_blk_off = ro_off.p;
    B.blk_start = ro_bs.p; B.blk_len = ro_bl.p; B.blk_read = ro_rd.p;
    return 0;
}

int irf_ctx::wait(uint64_t id) {
    HIP_TRY(hipSetDevice(device));
    for (auto& s : stage)
        if (s.in_use && s.batch_id == id) HIP_TRY(hipEventSynchronize(s.copied));
    return 0;    // older batches were waited for when their slot was reused
}

// one IR table: t = 0 non-directional (both strands), t = 1 directional under `dirn`
// (dual: both tables in one pass over the planes — table 0 into d_out_*[0], the directional one into d_out_*[1])
dev::IntronTables irf_ctx::intron_tables(int t, int dirn, bool dual) {
    dev::IntronTables T;
    memset(&T, 0, sizeof T);
    T.n_intron = n_intron; T.n_sp = n_sp; T.plane_stride = plane_stride; T.diff = d_diff.p; T.carry = d_carry.p;
    T.recs = d_in_recs.p; T.blks = d_blk_recs.p; T.pmap = pmap.empty() ? nullptr : d_pmap.p;
    T.sp_cnt = d_sp_cnt.p; T.left_cnt = d_left_cnt.p; T.right_cnt = d_right_cnt.p;
    T.n_ustart = n_us; T.n_uend = n_ue;
    T.ukeys = d_ukeys.p; T.ucount = d_ucount.p; T.n_unique = n_unique;
    T.known_cnt = d_known_cnt.p; T.n_kj = n_kj;
    T.ablate = (uint32_t)env_u64("IRF_ABLATE_STATS", 0);    // 0 unless built with -DIRF_DIAG
    T.trim_centre_pct = P.trim_centre_pct; T.pct_interpolate = P.pct_interpolate;
    T.flank_offset = P.flank_offset; T.flank_len = P.flank_len; T.flank_use_measured = P.flank_use_measured;
    T.directionality = dirn;
    T.warn_lowcover_min = P.warn_lowcover_min; T.warn_lowsplice_min = P.warn_lowsplice_min;
    T.warn_minor_ratio = P.warn_minor_ratio; T.warn_nonuniform_rel = P.warn_nonuniform_rel;
    T.warn_nonuniform_abs = P.warn_nonuniform_abs; T.warn_nonuniform_and = P.warn_nonuniform_and;
    T.out_f64 = d_out_f64[t].p; T.out_u32 = d_out_u32[t].p; T.err_flags = d_err.p;
    T.dual = dual ? 1u : 0u;
    if (dual) { T.out_f64 = d_out_f64[0].p; T.out_u32 = d_out_u32[0].p; T.out_f64_b = d_out_f64[1].p; T.out_u32_b = d_out_u32[1].p; }
    return T;
}

void irf_ctx::launch_stats(int t, int dirn, bool dual) {
    dev::launch_intron_stats(intron_tables(t, dirn, dual), d_itab.p + 2, stats_classes, stream);
}

// The following samples' planes (PlaneSet spare[]), cleared on the side stream.  Called by irf_finalize BEHIND the
// statistics kernels and the rows: the clear then runs into the next sample's count kernel and costs it 0.1 ms (2.46 ->
// 2.56 ms at 50 M pairs); queued in front of the statistics kernels it cost those 0.3 ms (1.9 -> 2.2 ms, measured with
// one and with two workgroups per CU), so this is the cheaper place.
int irf_ctx::queue_spare_clears() {
    if (plane_stride) {
        // (a set that is cleared for the FIRST time is a full memset, 5 GB: the first two samples of a context pay that once;
        //  afterwards only the lines the sample marked are written, 0.8 GB)
        const size_t n_seg2 = 2 * plane_stride / dev::kDepthSegment;
        bool waited = false;
        for (int k = 0; k < 2; k++) {
            PlaneSet& sp = spare[k];
            if (sp.clean) continue;
            const bool had = sp.allocated(plane_stride);
            // a first spare set when it leaves room for what the next sample may still grow (junction buffers, the reorder
            // copies: a set's size again, at least 1 GiB); the second one with it when the device has plenty of room (three
            // more sets would still fit), else never
            if (!had) {
                size_t free_b = 0, total_b = 0;
                const size_t set_b = (size_t)2 * plane_stride * sizeof(int32_t);
                const size_t need_b = k == 0 ? set_b + std::max<size_t>(set_b, (size_t)1 << 30) : 3 * set_b;
                if ((k == 1 && !spare[0].allocated(plane_stride)) || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < need_b) { (void)hipGetLastError(); continue; }
            }
            if (sp.diff.reserve(2 * plane_stride) != hipSuccess || sp.seg_sum.reserve(n_seg2) != hipSuccess ||
                sp.dirty.reserve(dev::dirty_map_bytes(plane_stride)) != hipSuccess) {
                (void)hipGetLastError();           // no room: without a spare set irf_reset clears the planes in line
                sp.release();
                continue;
            }
            if (!waited) {
                HIP_TRY(hipEventRecord(ev_stats_done, stream));
                HIP_TRY(hipStreamWaitEvent(clear_stream, ev_stats_done, 0));
                waited = true;
            }
            if (had) {
                // this set held a sample's planes: only the lines that sample marked hold anything (a fifth of them at 50 M pairs)
                dev::launch_sparse_clear(sp.diff.p, sp.dirty.p, plane_stride, clear_stream);
                HIP_TRY(hipGetLastError());
            } else {
                HIP_TRY(hipMemsetAsync(sp.diff.p, 0, 2 * plane_stride * sizeof(int32_t), clear_stream));
                HIP_TRY(hipMemsetAsync(sp.dirty.p, 0, dev::dirty_map_bytes(plane_stride), clear_stream));
            }
            HIP_TRY(hipMemsetAsync(sp.seg_sum.p, 0, n_seg2 * sizeof(uint32_t), clear_stream));
            HIP_TRY(hipEventRecord(sp.cleared, clear_stream));
            sp.clean = true;
            sp.seq = ++clear_seq;
        }
    }
    return 0;
}

int irf_ctx::finalize() {
    if (!have_map) return fail(IRF_E_STATE, "irf_finalize before irf_chr_map_update");
    if (finalized) return 0;
    HIP_TRY(hipSetDevice(device));
    EvSet& E = evset[cur_set];
    hipEvent_t* const ev = E.ev;
    HIP_TRY(hipStreamSynchronize(copy_stream));
    HIP_TRY(hipEventRecord(ev[0], stream));
    // Everything that needs no answer from the device is queued before the first synchronisation: the depth carries
    // (exclusive scan of the per-segment sums the count kernel kept next to the planes) and the known-junction
    // lookups with the strandedness vote (known junctions were counted in place; novel ones never vote).
    if (!kj_depth_applied) {        // once per sample: a second irf_finalize after a failed one must not add them again
        dev::launch_known_junction_depth(count_tables(), stream);
        kj_depth_applied = true;
    }
    if (env_u64("IRF_RESCAN_PLANES", 0)) dev::launch_segment_sums(d_diff.p, 2 * plane_stride, d_seg_sum.p, stream);     // diagnostic: recompute the sums from the planes
    dev::launch_exclusive_scan_u32(d_seg_sum.p, d_carry.p, 2 * plane_stride / dev::kDepthSegment, d_ws.p, stream);
    HIP_TRY(hipEventRecord(ev[1], stream));
    // (the vote came with the known-junction depth events: irf_reset zeroed the two sums, the kernel runs once per sample)
    // The scalars the host needs (records in the junction buffer, the vote, the error flags) are copied to page-locked
    // memory behind the scan, and the statistics kernels are queued BEHIND that copy before the host waits for it: they
    // take the strandedness decision themselves from the vote in device memory (kernels.h: IntronTables::vote), so the
    // device has 1.8 ms of work queued while the host reads the scalars and sets up the junction sort.
    HIP_TRY(hipMemcpyAsync(h_scal, d_scalars.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(h_scal + 8, d_err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipEventRecord(ev_scalars, stream));
    // ---- per-intron statistics ----
    // The statistics kernels (1.9 ms of plane reads at 50 M pairs) need nothing from the novel junctions except in the last
    // columns of a row, and the sort-reduce of the novel junctions is ~40 small kernels and two host round trips that
    // leave the device idle: the statistics start now, the sort runs next to them on its own stream, and launch_ir_rows
    // writes the junction-dependent columns of the rows when both have finished.
    TRY(order_after_exports());
    HIP_TRY(hipEventRecord(ev[4], stream));
    {
        // both shapes are queued; the one that does not fit the vote returns at once (a few microseconds)
        dev::IntronTables one = intron_tables(0, 0, false), two = intron_tables(0, 1, true);
        one.vote = two.vote = d_scalars.p + 3;
        one.dir_ratio = two.dir_ratio = P.dir_ratio;
        dev::launch_intron_stats(one, d_itab.p + 0, stats_classes, stream);
        dev::launch_intron_stats(two, d_itab.p + 1, stats_classes, stream);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev[5], stream));
    HIP_TRY(hipEventSynchronize(ev_scalars));
    const unsigned long long* scal = h_scal;
    const uint32_t errf = (uint32_t)h_scal[8];
    if (errf) {
        (void)hipStreamSynchronize(stream);      // (the statistics of the refused sample: nothing reads them)
        char m[160];
        snprintf(m, sizeof m, "count kernel reported input errors (flags 0x%x: 1 chr id, 2 block beyond chromosome, 4 junction range, 8 junction capacity, 32 frag_blk_off of a device batch)", errf);
        return fail(errf & dev::ERR_JUNC_CAPACITY ? IRF_E_STATE : ((errf & dev::ERR_CSR) ? IRF_E_ARG : IRF_E_RANGE), m);
    }
    n_events = scal[0];
    votes[0] = scal[3]; votes[1] = scal[4];
    if (n_events > 0xffffffffull) { (void)hipStreamSynchronize(stream); return fail(IRF_E_RANGE, "more than 2^32-1 junction records"); }
    n_unique = 0;
    d_sorted = d_junc_a.p;
    uint32_t n_radix_passes = 0;
    // JunctionCount::Directional() (OQ-5): known junctions only, so the vote is in.  (The same expression as in the
    // statistics kernels, on the same two sums.)
    directionality = 0;
    if (votes[0] > votes[1] * (unsigned long long)P.dir_ratio) directionality = 1;
    else if (votes[1] > votes[0] * (unsigned long long)P.dir_ratio) directionality = -1;
    // the junction stream starts behind the count kernels and the scalars' copy (until now the host's wait ordered them)
    HIP_TRY(hipStreamWaitEvent(junc_stream, ev_scalars, 0));
    struct StreamSwap {            // (the code below says `stream`: it runs on the junction stream until this goes out of scope)
        hipStream_t& ref; hipStream_t saved;
        StreamSwap(hipStream_t& r, hipStream_t to) : ref(r), saved(r) { ref = to; }
        ~StreamSwap() { ref = saved; }
    };
    {
    StreamSwap on_junction_stream(stream, junc_stream);
    // ---- SpliceLeft / SpliceRight contributions of the known junctions ----
    if (n_us) HIP_TRY(hipMemsetAsync(d_left_cnt.p, 0, (size_t)n_us * 2 * 4, stream));
    if (n_ue) HIP_TRY(hipMemsetAsync(d_right_cnt.p, 0, (size_t)n_ue * 2 * 4, stream));
    dev::JuncLookupTables J;
    memset(&J, 0, sizeof J);
    J.ustart_pos = d_us_pos.p; J.uend_pos = d_ue_pos.p; J.n_ustart = n_us; J.n_uend = n_ue;
    J.left_cnt = d_left_cnt.p; J.right_cnt = d_right_cnt.p;
    J.known_keys = d_known_keys.p; J.known_kjidx = d_known_kjidx.p; J.known_dir = d_known_dir.p;
    J.known_us_idx = d_known_us_idx.p; J.known_ue_idx = d_known_ue_idx.p;
    J.n_known = n_known_dev; J.n_kj = n_kj; J.known_cnt = d_known_cnt.p;
    J.dir_votes = nullptr;
    dev::launch_known_lookups(J, stream);
    HIP_TRY(hipGetLastError());
    // ---- junction events -> sorted unique table ----
    if (n_events) {
        // (sized to the records there are, not to the buffer's capacity: a few hundred thousand after the per-chunk merge)
        if (d_junc_b.reserve(n_events) != hipSuccess || d_juncw_b.reserve(n_events) != hipSuccess || d_wex.reserve(n_events) != hipSuccess)
            return fail(IRF_E_NOMEM, "hipMalloc of the sort buffer failed");
        uint64_t ws_need = std::max(dev::radix_workspace_elems(n_events), dev::scan_workspace_elems(n_events));
        // the sort has its own workspace: d_ws still holds nothing the statistics need, but keep the two apart
        if (ws_need > d_ws_sort.cap) { if (d_ws_sort.reserve(ws_need) != hipSuccess) return fail(IRF_E_NOMEM, "hipMalloc of the sort workspace failed"); }
        unsigned long long init[2] = {0ull, ~0ull};
        HIP_TRY(hipMemcpyAsync(d_scalars.p + 1, init, sizeof init, hipMemcpyHostToDevice, stream));
        dev::launch_key_bits(d_junc_a.p, n_events, d_scalars.p + 1, stream);
        unsigned long long bits[2];
        HIP_TRY(hipMemcpyAsync(bits, d_scalars.p + 1, sizeof bits, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        const unsigned long long varying = bits[0] ^ bits[1];
        uint64_t* src = d_junc_a.p;
        uint64_t* dst = d_junc_b.p;
        uint32_t* wsrc = d_juncw_a.p;
        uint32_t* wdst = d_juncw_b.p;
        for (int shift = 0; shift < 64; shift += 8) {
            if (((varying >> shift) & 0xffull) == 0) continue;     // digit identical in every key
            n_radix_passes++;
            dev::launch_radix_pass(src, dst, wsrc, wdst, n_events, shift, d_ws_sort.p, stream);
            std::swap(src, dst);
            std::swap(wsrc, wdst);
        }
        d_sorted = src;
        if (d_flags.reserve(n_events) != hipSuccess || d_uidx.reserve(n_events) != hipSuccess)
            return fail(IRF_E_NOMEM, "hipMalloc of the junction compaction buffers failed");
        dev::launch_head_flags(src, n_events, d_flags.p, stream);
        dev::launch_exclusive_scan_u32(d_flags.p, d_uidx.p, n_events, d_ws_sort.p, stream);
        dev::launch_exclusive_scan_u32(wsrc, d_wex.p, n_events, d_ws_sort.p, stream);
        uint32_t last[2];
        HIP_TRY(hipMemcpyAsync(&last[0], d_uidx.p + (n_events - 1), 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(&last[1], d_flags.p + (n_events - 1), 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        n_unique = (uint64_t)last[0] + last[1];
        if (d_ukeys.reserve(n_unique) != hipSuccess || d_ustart.reserve(n_unique) != hipSuccess || d_ucount.reserve(n_unique) != hipSuccess)
            return fail(IRF_E_NOMEM, "hipMalloc of the junction table failed");
        dev::launch_unique_runs(src, d_flags.p, d_uidx.p, n_events, d_ukeys.p, d_ustart.p, stream);
        dev::launch_run_sums(d_ustart.p, n_unique, n_events, d_wex.p, wsrc, d_ucount.p, stream);
        // ---- SpliceLeft / SpliceRight contributions of the novel junctions ----
        J.ukeys = d_ukeys.p; J.ucount = d_ucount.p; J.n_unique = n_unique;
        dev::launch_junction_lookups(J, stream);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(E.junc_done, stream));
    }
    // ---- the junction-dependent columns of the rows ----
    HIP_TRY(hipStreamWaitEvent(stream, E.junc_done, 0));
    HIP_TRY(hipEventRecord(ev[2], stream));
    if (directionality != 0) dev::launch_ir_rows(intron_tables(0, directionality, true), stream);
    else dev::launch_ir_rows(intron_tables(0, 0, false), stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev[3], stream));
    TRY(queue_spare_clears());
    // Everything is queued.  The host does not wait for it here: whatever reads a result (irf_get_results, an export that
    // is not IRF_COL_NO_SYNC, irf_get_timing, irf_get_depth), irf_reset and irf_sync wait first (settle()), and exports with
    // IRF_COL_NO_SYNC queue behind it.  A caller that counts sample after sample thus prepares the next sample's
    // submission while the statistics kernels of this one still run.
    E.timing.n_count_launches = (uint32_t)E.count_events_used;
    E.timing.n_radix_passes = n_radix_passes;
    E.timing.n_sort_records = n_events;
    E.finalized = true;
    E.timed = false;
    finalized = true;
    final_pending = true;
    gathered = false;
    results_ready = false;
    return 0;
}

// Waits for what irf_finalize queued.
int irf_ctx::settle() {
    if (!final_pending) return 0;
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamSynchronize(stream));
    final_pending = false;
    return 0;
}

// The kernel times of a finalized sample, read off its events (which are complete: settle() ran).
void irf_ctx::collect_timing(EvSet& E) {
    if (E.timed || !E.finalized) return;
    irf_timing& timing = E.timing;
    hipEvent_t* const ev = E.ev;
    float ms = 0;
    timing.count_ms = 0;
    for (size_t i = 0; i < E.count_events_used; i++) {
        if (hipEventElapsedTime(&ms, E.count_events[i].first, E.count_events[i].second) == hipSuccess) timing.count_ms += ms;
    }
    if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) timing.scan_ms = ms;
    // (junction_ms: known-junction lookups and the vote, up to the start of the statistics kernels, + the sort-reduce of the
    //  novel junctions, which runs NEXT TO the statistics kernels: the two overlap, their sum is not a wall time)
    if (hipEventElapsedTime(&ms, ev[1], ev[4]) == hipSuccess) timing.junction_ms = ms;
    if (hipEventElapsedTime(&ms, ev[4], E.junc_done) == hipSuccess) timing.junction_ms += ms;
    if (hipEventElapsedTime(&ms, ev[4], ev[5]) == hipSuccess) timing.stats_ms = ms;
    if (hipEventElapsedTime(&ms, ev[2], ev[3]) == hipSuccess) timing.table_ms = ms;       // launch_ir_rows
    if (hipEventElapsedTime(&ms, ev[0], ev[3]) == hipSuccess) timing.total_finalize_ms = ms;
    (void)hipGetLastError();
    E.timed = true;
}

// Multi-GPU: the strandedness vote must be taken over ALL shards.  The host sums irf_get_votes over the ranks,
// decides, and imposes the decision here; the directional table is (re)computed if the local decision differed.
int irf_ctx::finalize_directional(int32_t dirn) {
    if (!finalized) return fail(IRF_E_STATE, "irf_finalize_directional before irf_finalize");
    dirn = dirn > 0 ? 1 : (dirn < 0 ? -1 : 0);
    TRY(settle());
    if (dirn == directionality) return 0;
    HIP_TRY(hipSetDevice(device));
    directionality = dirn;
    ir_prepared[1] = false;
    if (dirn != 0) {
        TRY(order_after_exports());
        launch_stats(1, dirn);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
    }
    results_ready = false;
    return 0;
}

// Device -> host, plus the host-side joins that only the text writers need
// (known ∪ observed junctions with strand annotation).
int irf_ctx::fetch_results() {
    if (!finalized) return fail(IRF_E_STATE, "results requested before irf_finalize");
    if (results_ready) return 0;
    HIP_TRY(hipSetDevice(device));
    TRY(settle());
    std::vector<uint32_t> tmp;
    auto down32 = [&](const uint32_t* p, size_t n) -> hipError_t {
        tmp.resize(n);
        if (!n) return hipSuccess;
        return hipMemcpy(tmp.data(), p, n * 4, hipMemcpyDeviceToHost);
    };
    HIP_TRY(down32(d_chr_cnt.p, (size_t)n_chr * 2));
    h_chr_cnt.assign(tmp.begin(), tmp.end());
    HIP_TRY(down32(d_roi_cnt.p, (size_t)n_roi * 2));
    h_roi_cnt.assign(tmp.begin(), tmp.end());
    HIP_TRY(down32(d_sp_cnt.p, (size_t)n_sp * 2));
    h_sp_cnt = tmp;
    h_sp_chr = sp_slot;
    h_sp_pos = sp_pos_sorted;
    // junctions: known rows carry their in-place counters; novel unique keys are decoded, their two strands paired,
    // and both lists (each sorted by (slot,left,right)) are merged
    {
        std::vector<uint32_t> kc((size_t)n_kj * 2, 0u);
        if (n_kj) HIP_TRY(hipMemcpy(kc.data(), d_known_cnt.p, kc.size() * 4, hipMemcpyDeviceToHost));
        std::vector<ImpJunc> nov;
        TRY(own_novel_junctions(nov));
        if (!imp_junc.empty()) {
            // novel junctions of other shards (irf_gather / irf_import_junctions): one sorted list, equal coordinates merged
            nov.insert(nov.end(), imp_junc.begin(), imp_junc.end());
            std::stable_sort(nov.begin(), nov.end(), [](const ImpJunc& x, const ImpJunc& y) {
                if (x.chr != y.chr) return x.chr < y.chr;
                if (x.left != y.left) return x.left < y.left;
                return x.right < y.right;
            });
            size_t w = 0;
            for (size_t i = 0; i < nov.size(); i++) {
                if (w && nov[w - 1].chr == nov[i].chr && nov[w - 1].left == nov[i].left && nov[w - 1].right == nov[i].right) {
                    nov[w - 1].c0 += nov[i].c0; nov[w - 1].c1 += nov[i].c1;
                } else nov[w++] = nov[i];
            }
            nov.resize(w);
        }
        std::vector<uint32_t> oc, ol, orr, c0, c1;
        for (const ImpJunc& j : nov) { oc.push_back(j.chr); ol.push_back(j.left); orr.push_back(j.right); c0.push_back(j.c0); c1.push_back(j.c1); }
        h_j_chr.clear(); h_j_left.clear(); h_j_right.clear(); h_j_refdir.clear();
        std::vector<uint32_t> m0, m1;
        size_t a = 0, b = 0, na = oc.size(), nb = kj_left.size();
        auto less_ab = [&](size_t i, size_t j) {      // novel[i] < known[j]
            if (oc[i] != kj_slot[j]) return oc[i] < kj_slot[j];
            if (ol[i] != kj_left[j]) return ol[i] < kj_left[j];
            return orr[i] < kj_right[j];
        };
        auto eq_ab = [&](size_t i, size_t j) { return oc[i] == kj_slot[j] && ol[i] == kj_left[j] && orr[i] == kj_right[j]; };
        n_junction_events_total = 0;
        while (a < na || b < nb) {
            if (a < na && b < nb && eq_ab(a, b)) {     // cannot happen for keys the device table holds; kept for safety
                h_j_chr.push_back(oc[a]); h_j_left.push_back(ol[a]); h_j_right.push_back(orr[a]);
                m0.push_back(c0[a] + kc[b]); m1.push_back(c1[a] + kc[(size_t)n_kj + b]); h_j_refdir.push_back(kj_dir[b]);
                a++; b++;
            } else if (b >= nb || (a < na && less_ab(a, b))) {
                h_j_chr.push_back(oc[a]); h_j_left.push_back(ol[a]); h_j_right.push_back(orr[a]);
                m0.push_back(c0[a]); m1.push_back(c1[a]); h_j_refdir.push_back(0);
                a++;
            } else {
                h_j_chr.push_back(kj_slot[b]); h_j_left.push_back(kj_left[b]); h_j_right.push_back(kj_right[b]);
                m0.push_back(kc[b]); m1.push_back(kc[(size_t)n_kj + b]); h_j_refdir.push_back(kj_dir[b]);
                b++;
            }
            n_junction_events_total += (uint64_t)m0.back() + m1.back();
        }
        h_j_cnt = m0;
        h_j_cnt.insert(h_j_cnt.end(), m1.begin(), m1.end());
    }
    for (int t = 0; t < 2; t++) {
        if (t == 1 && directionality == 0) { h_f64[1].clear(); h_u32[1].clear(); h_warn[1].clear(); break; }
        h_f64[t].resize((size_t)n_intron * 8);
        h_u32[t].resize((size_t)n_intron * 6);
        if (n_intron) {
            HIP_TRY(hipMemcpy(h_f64[t].data(), d_out_f64[t].p, h_f64[t].size() * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(h_u32[t].data(), d_out_u32[t].p, h_u32[t].size() * 4, hipMemcpyDeviceToHost));
        }
        h_warn[t].resize(n_intron);
        for (uint32_t i = 0; i < n_intron; i++) h_warn[t][i] = (uint8_t)h_u32[t][(size_t)5 * n_intron + i];
    }
    results_ready = true;
    return 0;
}

// ---- multi-GPU collection ---------------------------------------------------------------------------------------
// Copies one result column into caller-owned device memory.  IR tables of a SHARDED context leave with the rows this
// context does not own zeroed (rows on no BAM reference at all stay with IRF_COL_KEEP_ORPHANS), so that adding the
// shards' columns (reduce(SUM) over RCCL, irf_import_device, irf_gather) gives the complete table.
int irf_ctx::export_col(uint32_t what_in, void* dst, uint64_t* n_out) {
    if (!n_out) return IRF_E_ARG;
    if (!finalized) return fail(IRF_E_STATE, "irf_export_device before irf_finalize");
    const bool keep_orphans = (what_in & IRF_COL_KEEP_ORPHANS) != 0, no_sync = (what_in & IRF_COL_NO_SYNC) != 0;
    const uint32_t what = what_in & ~(uint32_t)(IRF_COL_KEEP_ORPHANS | IRF_COL_NO_SYNC);
    const void* src = nullptr;
    uint64_t n = 0;
    size_t elem = 4;
    int table = -1;
    switch (what) {
        case IRF_COL_IR_NONDIR_F64: src = d_out_f64[0].p; n = (uint64_t)n_intron * 8; elem = 8; table = 0; break;
        case IRF_COL_IR_NONDIR_U32: src = d_out_u32[0].p; n = (uint64_t)n_intron * 6; table = 0; break;
        case IRF_COL_IR_DIR_F64: src = d_out_f64[1].p; n = directionality ? (uint64_t)n_intron * 8 : 0; elem = 8; table = 1; break;
        case IRF_COL_IR_DIR_U32: src = d_out_u32[1].p; n = directionality ? (uint64_t)n_intron * 6 : 0; table = 1; break;
        case IRF_COL_JUNC_KEYS: src = d_ukeys.p; n = n_unique; elem = 8; break;
        case IRF_COL_JUNC_COUNTS: src = d_ucount.p; n = n_unique; break;
        case IRF_COL_SP_CNT: src = d_sp_cnt.p; n = (uint64_t)n_sp * 2; break;
        case IRF_COL_ROI_CNT: src = d_roi_cnt.p; n = (uint64_t)n_roi * 2; break;
        case IRF_COL_CHR_CNT: src = d_chr_cnt.p; n = (uint64_t)n_chr * 2; break;
        case IRF_COL_KNOWN_CNT: src = d_known_cnt.p; n = (uint64_t)n_kj * 2; break;
        default: return fail(IRF_E_ARG, "irf_export_device: unknown column");
    }
    if (any_split && (what == IRF_COL_JUNC_KEYS || what == IRF_COL_JUNC_COUNTS))
        return fail(IRF_E_STATE, "irf_export_device: a context that owns a reference in part hands its novel junctions out as coordinates (irf_export_junctions)");
    *n_out = n;
    if (!dst || !n) return 0;
    HIP_TRY(hipSetDevice(device));
    if (table >= 0 && sharded) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, dst) != hipSuccess || at.type != hipMemoryTypeDevice) {
            (void)hipGetLastError();
            return fail(IRF_E_ARG, "irf_export_device: the IR tables of a sharded context are exported into DEVICE memory (their foreign rows are zeroed there)");
        }
    }
    if (table >= 0 && !sharded && no_sync) {
        // an IR table, copy only queued: on the export stream, behind everything queued on the main stream so far
        HIP_TRY(hipEventRecord(ev_tables_ready, stream));
        held_exports.push_back(HeldExport{dst, src, (size_t)(n * elem)});       // (issued by flush_exports)
        exports_pending = true;
        return 0;
    }
    HIP_TRY(hipMemcpyAsync(dst, src, n * elem, hipMemcpyDefault, stream));       // dst: device memory, or page-locked host memory
    if (table >= 0 && sharded) {
        const uint32_t keep = (1u << dev::kRowMine) | (keep_orphans ? (1u << dev::kRowOrphan) : 0u);
        if (elem == 8) dev::launch_mask_rows_f64((double*)dst, 8, n_intron, d_row_code.p, keep, stream);
        else dev::launch_mask_rows_u32((uint32_t*)dst, 6, n_intron, d_row_code.p, keep, stream);
        HIP_TRY(hipGetLastError());
    }
    if (!no_sync) { HIP_TRY(hipStreamSynchronize(stream)); TRY(settle()); }
    return 0;
}

// Adds a column another shard exported (and that an RCCL transfer or a peer copy brought to this device) into this
// context's table.  The first import into an IR table zeroes the rows other shards own (this context computed
// all-zero-depth rows for them).  Junction key columns cannot be added: their packing depends on the shard's layout
// (irf_export_junctions / irf_import_junctions carry them as coordinates).
int irf_ctx::import_col(uint32_t what, const void* src, uint64_t n) {
    if (!finalized) return fail(IRF_E_STATE, "irf_import_device before irf_finalize");
    if (!src && n) return fail(IRF_E_ARG, "irf_import_device: null source");
    HIP_TRY(hipSetDevice(device));
    uint64_t want = 0;
    int table = -1;
    bool f64 = false;
    void* dstp = nullptr;
    switch (what) {
        case IRF_COL_IR_NONDIR_F64: dstp = d_out_f64[0].p; want = (uint64_t)n_intron * 8; f64 = true; table = 0; break;
        case IRF_COL_IR_NONDIR_U32: dstp = d_out_u32[0].p; want = (uint64_t)n_intron * 6; table = 0; break;
        case IRF_COL_IR_DIR_F64: dstp = d_out_f64[1].p; want = directionality ? (uint64_t)n_intron * 8 : 0; f64 = true; table = 1; break;
        case IRF_COL_IR_DIR_U32: dstp = d_out_u32[1].p; want = directionality ? (uint64_t)n_intron * 6 : 0; table = 1; break;
        case IRF_COL_SP_CNT: dstp = d_sp_cnt.p; want = (uint64_t)n_sp * 2; break;
        case IRF_COL_ROI_CNT: dstp = d_roi_cnt.p; want = (uint64_t)n_roi * 2; break;
        case IRF_COL_CHR_CNT: dstp = d_chr_cnt.p; want = (uint64_t)n_chr * 2; break;
        case IRF_COL_KNOWN_CNT: dstp = d_known_cnt.p; want = (uint64_t)n_kj * 2; break;
        default: return fail(IRF_E_ARG, "irf_import_device: this column cannot be imported (junction keys: use irf_import_junctions)");
    }
    if (n != want) return fail(IRF_E_ARG, "irf_import_device: element count does not match this context's table");
    if (!n) return 0;
    TRY(settle());
    TRY(order_after_exports());
    if (table >= 0 && !ir_prepared[table]) {
        const uint32_t keep = (1u << dev::kRowMine) | (1u << dev::kRowOrphan);
        dev::launch_mask_rows_f64(d_out_f64[table].p, 8, n_intron, d_row_code.p, keep, stream);
        dev::launch_mask_rows_u32(d_out_u32[table].p, 6, n_intron, d_row_code.p, keep, stream);
        ir_prepared[table] = true;
    }
    if (f64) dev::launch_add_into_f64((double*)dstp, (const double*)src, n, stream);
    else dev::launch_add_into_u32((uint32_t*)dstp, (const uint32_t*)src, n, stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(stream));
    results_ready = false;
    return 0;
}

// This context's novel junctions as coordinates (the packed keys depend on the shard's plane layout), both strands of
// a junction in one row, sorted by (BAM refID, left, right).
int irf_ctx::own_novel_junctions(std::vector<ImpJunc>& out) {
    out.clear();
    if (!n_unique) return 0;
    HIP_TRY(hipSetDevice(device));
    TRY(settle());
    std::vector<uint64_t> uk(n_unique);
    std::vector<uint32_t> uc(n_unique);
    HIP_TRY(hipMemcpy(uk.data(), d_ukeys.p, n_unique * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(uc.data(), d_ucount.p, n_unique * 4, hipMemcpyDeviceToHost));
    uint32_t chr = 0;
    for (uint64_t u = 0; u < n_unique; u++) {
        const uint64_t gl = uk[u] >> dev::kJuncLeftShift;
        const uint32_t span = (uint32_t)((uk[u] >> 1) & ((1ull << dev::kJuncSpanBits) - 1));
        const uint32_t d = (uint32_t)(uk[u] & 1);
        while (chr + 1 < n_chr && (!chr_owned[chr] || gl >= chr_base[chr] + ((((uint64_t)chr_len[chr] + 1 + 63) / 64) * 64))) chr++;
        const uint32_t left = (uint32_t)(gl - chr_base[chr]);
        if (!in_own(chr, left)) continue;       // a reference owned in part: the junction belongs to the context that owns its left end
        if (!out.empty() && out.back().chr == chr && out.back().left == left && out.back().right == left + span) {
            (d ? out.back().c1 : out.back().c0) += uc[u];
        } else {
            out.push_back(ImpJunc{chr, left, left + span, d ? 0u : uc[u], d ? uc[u] : 0u});
        }
    }
    return 0;
}

// ============================================================================
extern "C" {

void irf_params_default(irf_params* p) {
    memset(p, 0, sizeof *p);
    p->struct_size = (uint32_t)sizeof *p;
    p->skip_flag_mask = 0x4;
    p->pair_long_inclusive = 0;
    p->pair_short_inclusive = 1;
    p->junc_min_flank = 5;
    p->sp_overhang_left = 5;
    p->sp_overhang_right = 4;
    p->sp_inclusive = 0;
    p->trim_centre_pct = 40;
    p->pct_interpolate = 1;
    p->flank_offset = 5;
    p->flank_len = 50;
    p->flank_use_measured = 1;
    p->dir_ratio = 10;
    p->out_precision = 6;
    p->warn_lowcover_min = 10.0;
    p->warn_lowsplice_min = 4.0;
    p->warn_minor_ratio = 1.33;
    p->warn_nonuniform_rel = 1.5;
    p->warn_nonuniform_abs = 4.0;
    p->junc_event_capacity = 1ull << 26;
    p->keep_depth = 0;
}

static void set_err(char* buf, size_t n, const std::string& s) {
    if (buf && n) { snprintf(buf, n, "%s", s.c_str()); }
}

int irf_ref_load_mem(const char* const bufs[4], const size_t lens[4], irf_ref** out, char* errbuf, size_t errbuf_len) {
    if (!bufs || !lens || !out) return IRF_E_ARG;
    irf_ref* r = new irf_ref();
    std::string err;
    bool ok = irf::parse_cover_bed(bufs[0], lens[0], r->t, err) && irf::parse_sj_ref(bufs[1], lens[1], r->t, err) &&
              irf::parse_read_continues_ref(bufs[2], lens[2], r->t, err) && irf::parse_roi_bed(bufs[3], lens[3], r->t, err);
    if (!ok) { set_err(errbuf, errbuf_len, err); delete r; return IRF_E_PARSE; }
    irf::finish_tables(r->t);
    *out = r;
    return 0;
}

int irf_ref_load_files(const char* cover, const char* sj, const char* rc, const char* roi, irf_ref** out, char* errbuf,
                       size_t errbuf_len) {
    if (!cover || !sj || !rc || !roi || !out) return IRF_E_ARG;
    std::string s[4], err;
    const char* paths[4] = {cover, sj, rc, roi};
    for (int i = 0; i < 4; i++)
        if (!irf::read_file(paths[i], s[i], err)) { set_err(errbuf, errbuf_len, err); return IRF_E_IO; }
    const char* bufs[4] = {s[0].data(), s[1].data(), s[2].data(), s[3].data()};
    size_t lens[4] = {s[0].size(), s[1].size(), s[2].size(), s[3].size()};
    return irf_ref_load_mem(bufs, lens, out, errbuf, errbuf_len);
}

void irf_ref_free(irf_ref* r) { delete r; }
uint32_t irf_ref_n_intron(const irf_ref* r) { return r ? (uint32_t)r->t.n_intron() : 0; }
uint32_t irf_ref_n_junction(const irf_ref* r) { return r ? (uint32_t)r->t.kj_left.size() : 0; }
uint32_t irf_ref_n_point(const irf_ref* r) { return r ? (uint32_t)r->t.sp_pos.size() : 0; }
uint32_t irf_ref_n_roi(const irf_ref* r) { return r ? (uint32_t)r->t.roi_start.size() : 0; }

int irf_create(const irf_params* params, const irf_ref* ref, int device, irf_ctx** out) {
    if (!ref || !out) return IRF_E_ARG;
    irf_params P;
    if (params) {
        if (params->struct_size != sizeof(irf_params)) return IRF_E_ARG;
        P = *params;
    } else {
        irf_params_default(&P);
    }
    if (P.flank_len > 512 || P.trim_centre_pct > 100) return IRF_E_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return IRF_E_DEVICE;
    irf_ctx* c = new irf_ctx();
    c->P = P;
    c->ref = ref->t;
    c->device = device;
    int rc = c->init();
    if (rc != 0) { c->free_events(); delete c; return rc; }
    *out = c;
    return 0;
}

void irf_destroy(irf_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->clear_stream) (void)hipStreamSynchronize(c->clear_stream);
    if (c->junc_stream) (void)hipStreamSynchronize(c->junc_stream);      // (table exports in flight)
    c->free_events();
    delete c;
}

const char* irf_last_error(const irf_ctx* c) { return c ? c->err.c_str() : "null context"; }

int irf_chr_map_update(irf_ctx* c, uint32_t n_chr, const char* const* names, const uint32_t* lens, const uint8_t* owned) {
    if (!c) return IRF_E_ARG;
    if (!names || !lens) return c->fail(IRF_E_ARG, "irf_chr_map_update needs names and lengths");
    for (uint32_t i = 0; i < n_chr; i++) {
        if (!names[i]) return c->fail(IRF_E_ARG, "null chromosome name");
        if (lens[i] > 0x7fffffffu) return c->fail(IRF_E_RANGE, "reference length beyond 2^31-1");
    }
    return c->build_map(n_chr, names, lens, owned, nullptr, nullptr);
}

int irf_chr_map_update_ranges(irf_ctx* c, uint32_t n_chr, const char* const* names, const uint32_t* lens, const uint8_t* owned,
                              const uint32_t* own_start, const uint32_t* own_end) {
    if (!c) return IRF_E_ARG;
    if (!names || !lens) return c->fail(IRF_E_ARG, "irf_chr_map_update_ranges needs names and lengths");
    if ((own_start == nullptr) != (own_end == nullptr)) return c->fail(IRF_E_ARG, "irf_chr_map_update_ranges: own_start and own_end come together");
    for (uint32_t i = 0; i < n_chr; i++) {
        if (!names[i]) return c->fail(IRF_E_ARG, "null chromosome name");
        if (lens[i] > 0x7fffffffu) return c->fail(IRF_E_RANGE, "reference length beyond 2^31-1");
    }
    return c->build_map(n_chr, names, lens, owned, own_start, own_end);
}

int irf_submit_batch(irf_ctx* c, const irf_batch* b, uint64_t* id) { return c ? c->submit(b, id) : IRF_E_ARG; }
int irf_wait(irf_ctx* c, uint64_t id) { return c ? c->wait(id) : IRF_E_ARG; }
int irf_sync(irf_ctx* c) {
    if (!c) return IRF_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(IRF_E_DEVICE, "hipSetDevice");
    if (int rc = c->flush_exports()) return rc;
    if (hipStreamSynchronize(c->stream) != hipSuccess ||
        (c->export_stream && hipStreamSynchronize(c->export_stream) != hipSuccess)) return c->fail(IRF_E_DEVICE, "irf_sync: stream synchronize failed");
    c->exports_pending = false;
    return c->settle();
}
int irf_set_input_order(irf_ctx* c, uint32_t order) {
    if (!c) return IRF_E_ARG;
    if (order != IRF_ORDER_SORTED && order != IRF_ORDER_ANY) return c->fail(IRF_E_ARG, "irf_set_input_order: unknown order");
    c->input_order = order;
    return 0;
}
int irf_finalize(irf_ctx* c) { return c ? c->finalize() : IRF_E_ARG; }
int irf_finalize_directional(irf_ctx* c, int32_t d) { return c ? c->finalize_directional(d) : IRF_E_ARG; }
int irf_get_votes(irf_ctx* c, uint64_t out[2]) {
    if (!c || !out) return IRF_E_ARG;
    if (!c->finalized) return c->fail(IRF_E_STATE, "irf_get_votes before irf_finalize");
    out[0] = c->votes[0]; out[1] = c->votes[1];
    return 0;
}

int irf_reset(irf_ctx* c) {
    if (!c) return IRF_E_ARG;
    if (!c->have_map) return c->fail(IRF_E_STATE, "irf_reset before irf_chr_map_update");
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(IRF_E_DEVICE, "hipSetDevice");
    if (hipStreamSynchronize(c->copy_stream) != hipSuccess) return c->fail(IRF_E_DEVICE, "hipStreamSynchronize");
    return c->zero_counters();
}

int irf_get_results(irf_ctx* c, irf_results* out) {
    if (!c || !out) return IRF_E_ARG;
    int rc = c->fetch_results();
    if (rc != 0) return rc;
    memset(out, 0, sizeof *out);
    out->directionality = c->directionality;
    out->n_chr = c->n_chr; out->chr_cnt = c->h_chr_cnt.data();
    out->n_roi = c->n_roi; out->roi_cnt = c->h_roi_cnt.data();
    out->n_sp = c->n_sp; out->sp_chr = c->h_sp_chr.data(); out->sp_pos = c->h_sp_pos.data(); out->sp_cnt = c->h_sp_cnt.data();
    out->n_junc = c->h_j_left.size();
    out->junc_chr = c->h_j_chr.data(); out->junc_left = c->h_j_left.data(); out->junc_right = c->h_j_right.data();
    out->junc_cnt = c->h_j_cnt.data(); out->junc_refdir = c->h_j_refdir.data();
    out->n_intron = c->n_intron;
    out->has_dir = c->directionality != 0;
    const size_t n = c->n_intron;
    for (int t = 0; t < 2; t++) {
        irf_ir_table& T = t == 0 ? out->nondir : out->dir;
        if (t == 1 && !out->has_dir) break;
        const double* f = c->h_f64[t].data();
        const uint32_t* u = c->h_u32[t].data();
        T.coverage = f; T.intron_depth = f + n; T.p25 = f + 2 * n; T.p50 = f + 3 * n; T.p75 = f + 4 * n;
        T.first50 = f + 5 * n; T.last50 = f + 6 * n; T.irratio = f + 7 * n;
        T.sp_left = u; T.sp_right = u + n; T.splice_left = u + 2 * n; T.splice_right = u + 3 * n; T.splice_exact = u + 4 * n;
        T.warning = c->h_warn[t].data();
    }
    out->n_fragments = c->n_fragments; out->n_blocks = c->n_blocks; out->n_junction_events = c->n_junction_events_total;
    return 0;
}

const char* irf_chr_name(const irf_ctx* c, uint32_t slot) {
    return (c && slot < c->slot_names.size()) ? c->slot_names[slot].c_str() : nullptr;
}
uint32_t irf_n_chr_slots(const irf_ctx* c) { return c ? (uint32_t)c->slot_names.size() : 0; }

int irf_get_depth(irf_ctx* c, uint32_t chr, uint32_t start, uint32_t end, uint32_t dir, uint32_t* out) {
    if (!c || !out) return IRF_E_ARG;
    if (!c->finalized) return c->fail(IRF_E_STATE, "irf_get_depth before irf_finalize");
    if (!c->pmap.empty()) return c->fail(IRF_E_STATE, "irf_get_depth needs params.keep_depth = 1 (the default keeps only the measured intron regions in HBM)");
    if (chr >= c->n_chr || !c->chr_owned[chr] || start > end || end > c->chr_len[chr] || dir > 2) return c->fail(IRF_E_ARG, "irf_get_depth: bad range");
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(IRF_E_DEVICE, "hipSetDevice");
    if (int rc = c->settle()) return rc;
    size_t n = end - start;
    if (!n) return 0;
    // depth = carry of the enclosing 256-position segment + running sum of the difference events (host side)
    std::vector<uint32_t> a(n, 0u);
    for (int p = 0; p < 2; p++) {
        if (dir != 2 && (int)dir != p) continue;
        const uint64_t g0 = c->chr_base[chr] + start, g1 = c->chr_base[chr] + end;
        const uint64_t seg = g0 / irf::dev::kDepthSegment, e0 = seg * irf::dev::kDepthSegment;
        std::vector<int32_t> ev(g1 - e0);
        uint32_t carry = 0;
        if (hipMemcpy(ev.data(), c->d_diff.p + (uint64_t)p * c->plane_stride + e0, ev.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(&carry, c->d_carry.p + (uint64_t)p * (c->plane_stride / irf::dev::kDepthSegment) + seg, 4, hipMemcpyDeviceToHost) != hipSuccess)
            return c->fail(IRF_E_DEVICE, "hipMemcpy depth");
        uint32_t run = carry;
        for (uint64_t x = e0; x < g1; x++) {
            run += (uint32_t)ev[x - e0];
            if (x >= g0) a[x - g0] += run;
        }
    }
    memcpy(out, a.data(), n * 4);
    return 0;
}

int irf_export_device(irf_ctx* c, uint32_t what, void* dst, uint64_t* n_out) { return c ? c->export_col(what, dst, n_out) : IRF_E_ARG; }
int irf_import_device(irf_ctx* c, uint32_t what, const void* src, uint64_t n) { return c ? c->import_col(what, src, n) : IRF_E_ARG; }

int irf_export_junctions(irf_ctx* c, uint64_t* n, uint32_t* chr, uint32_t* left, uint32_t* right, uint32_t* cnt_neg, uint32_t* cnt_pos) {
    if (!c || !n) return IRF_E_ARG;
    if (!c->finalized) return c->fail(IRF_E_STATE, "irf_export_junctions before irf_finalize");
    std::vector<irf_ctx::ImpJunc> rows;
    int rc = c->own_novel_junctions(rows);
    if (rc != 0) return rc;
    const uint64_t cap = *n;
    *n = rows.size();
    if (!chr && !left && !right && !cnt_neg && !cnt_pos) return 0;          // size query
    if (!chr || !left || !right || !cnt_neg || !cnt_pos) return c->fail(IRF_E_ARG, "irf_export_junctions: null array");
    if (cap < rows.size()) return c->fail(IRF_E_RANGE, "irf_export_junctions: arrays too small (*n is the size needed)");
    for (size_t i = 0; i < rows.size(); i++) {
        chr[i] = rows[i].chr; left[i] = rows[i].left; right[i] = rows[i].right; cnt_neg[i] = rows[i].c0; cnt_pos[i] = rows[i].c1;
    }
    return 0;
}

int irf_import_junctions(irf_ctx* c, uint64_t n, const uint32_t* chr, const uint32_t* left, const uint32_t* right, const uint32_t* cnt_neg,
                         const uint32_t* cnt_pos) {
    if (!c) return IRF_E_ARG;
    if (!c->finalized) return c->fail(IRF_E_STATE, "irf_import_junctions before irf_finalize");
    if (n && (!chr || !left || !right || !cnt_neg || !cnt_pos)) return c->fail(IRF_E_ARG, "irf_import_junctions: null array");
    for (uint64_t i = 0; i < n; i++) {
        if (chr[i] >= c->n_chr || right[i] < left[i]) return c->fail(IRF_E_ARG, "irf_import_junctions: row outside the chromosome map");
        c->imp_junc.push_back(irf_ctx::ImpJunc{chr[i], left[i], right[i], cnt_neg[i], cnt_pos[i]});
    }
    c->results_ready = false;
    return 0;
}

// In-process collection: one host process drives several contexts (one per GPU, or several on one GPU), each finalized
// over its own chromosomes.  The strandedness vote is taken over all of them, then every table of the others is added
// into ctxs[root] (device-to-device copies into a scratch buffer on the root's device, no host round trip except the
// small novel-junction tables), which afterwards answers irf_get_results / irf_write_outputs for the whole sample.
int irf_gather(irf_ctx* const* ctxs, uint32_t n, uint32_t root) {
    if (!ctxs || n == 0 || root >= n) return IRF_E_ARG;
    for (uint32_t i = 0; i < n; i++) if (!ctxs[i]) return IRF_E_ARG;
    irf_ctx* R = ctxs[root];
    // the peers' columns are ADDED to the root's: a second gather of the same sample would add them twice
    if (R->gathered) return R->fail(IRF_E_STATE, "irf_gather: this sample was gathered already (irf_reset and count again, or read the results)");
    for (uint32_t i = 0; i < n; i++) {
        irf_ctx* c = ctxs[i];
        if (!c->finalized) return R->fail(IRF_E_STATE, "irf_gather: every context must be finalized");
        if (c->n_chr != R->n_chr || c->n_intron != R->n_intron || c->n_sp != R->n_sp || c->n_roi != R->n_roi || c->n_kj != R->n_kj)
            return R->fail(IRF_E_ARG, "irf_gather: contexts with different reference tables or chromosome maps");
        for (uint32_t k = 0; k < c->n_chr; k++)
            for (uint32_t j = 0; j < i; j++)
                if (c->chr_owned[k] && ctxs[j]->chr_owned[k] && c->own_start[k] < ctxs[j]->own_end[k] && ctxs[j]->own_start[k] < c->own_end[k])
                    return R->fail(IRF_E_ARG, "irf_gather: a BAM reference (or a part of it) is owned by two contexts");
    }
    // JunctionCount::Directional() over all shards
    unsigned long long same = 0, opp = 0;
    for (uint32_t i = 0; i < n; i++) { same += ctxs[i]->votes[0]; opp += ctxs[i]->votes[1]; }
    int32_t dirn = 0;
    if (same > opp * (unsigned long long)R->P.dir_ratio) dirn = 1;
    else if (opp > same * (unsigned long long)R->P.dir_ratio) dirn = -1;
    for (uint32_t i = 0; i < n; i++) TRY(ctxs[i]->finalize_directional(dirn));
    static const uint32_t cols[] = {IRF_COL_IR_NONDIR_F64, IRF_COL_IR_NONDIR_U32, IRF_COL_IR_DIR_F64, IRF_COL_IR_DIR_U32,
                                    IRF_COL_SP_CNT, IRF_COL_ROI_CNT, IRF_COL_CHR_CNT, IRF_COL_KNOWN_CNT};
    for (uint32_t i = 0; i < n; i++) {
        if (i == root) continue;
        irf_ctx* c = ctxs[i];
        for (uint32_t what : cols) {
            uint64_t cnt = 0;
            TRY(c->export_col(what, nullptr, &cnt));
            if (!cnt) continue;
            const size_t bytes = cnt * ((what == IRF_COL_IR_NONDIR_F64 || what == IRF_COL_IR_DIR_F64) ? 8 : 4);
            void* on_peer = nullptr;
            void* on_root = nullptr;
            if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&on_peer, bytes) != hipSuccess) return R->fail(IRF_E_NOMEM, "irf_gather: hipMalloc");
            int rc = c->export_col(what, on_peer, &cnt);
            if (rc == 0 && (hipSetDevice(R->device) != hipSuccess || hipMalloc(&on_root, bytes) != hipSuccess)) rc = R->fail(IRF_E_NOMEM, "irf_gather: hipMalloc");
            if (rc == 0 && hipMemcpyPeer(on_root, R->device, on_peer, c->device, bytes) != hipSuccess) rc = R->fail(IRF_E_DEVICE, "irf_gather: hipMemcpyPeer");
            if (rc == 0) rc = R->import_col(what, on_root, cnt);
            if (rc == 0 && (hipSetDevice(R->device) != hipSuccess || hipStreamSynchronize(R->stream) != hipSuccess)) rc = R->fail(IRF_E_DEVICE, "irf_gather: synchronize");
            if (on_root) { (void)hipSetDevice(R->device); (void)hipFree(on_root); }
            if (on_peer) { (void)hipSetDevice(c->device); (void)hipFree(on_peer); }
            if (rc != 0) { if (c != R && R->err.empty()) R->err = c->err; return rc; }
        }
        std::vector<irf_ctx::ImpJunc> rows;
        TRY(c->own_novel_junctions(rows));
        R->imp_junc.insert(R->imp_junc.end(), rows.begin(), rows.end());
        // (n_fragments, the fragments SUBMITTED, stays per context: every shard may have been fed the whole sample)
    }
    R->gathered = true;
    R->results_ready = false;
    return 0;
}

int irf_host_alloc(size_t bytes, void** out) {
    if (!out) return IRF_E_ARG;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? 0 : IRF_E_NOMEM;
}
void irf_host_free(void* p) { if (p) (void)hipHostFree(p); }

int irf_get_layout(irf_ctx* c, irf_layout* out) {
    if (!c || !out) return IRF_E_ARG;
    if (!c->have_map) return c->fail(IRF_E_STATE, "irf_get_layout before irf_chr_map_update");
    out->plane_positions = c->plane_stride;
    out->full_positions = c->full_size;
    out->measured_positions = c->measured_valid;
    out->junction_capacity = c->d_junc_a.cap;
    return 0;
}

int irf_get_timing(irf_ctx* c, irf_timing* out) {
    if (!c || !out) return IRF_E_ARG;
    if (int rc = c->settle()) return rc;
    irf_ctx::EvSet& E = c->evset[c->finalized ? c->cur_set : c->cur_set ^ 1];      // (after irf_reset: the sample before it)
    c->collect_timing(E);
    *out = E.timing;
    return 0;
}

}   // extern "C"

// accessors for the writers (writers.cpp), which live in the same library
namespace irf {
const RefTables& ctx_ref(const irf_ctx* c) { return c->ref; }
const irf_params& ctx_params(const irf_ctx* c) { return c->P; }
const std::vector<uint32_t>& ctx_refchr_slot(const irf_ctx* c) { return c->refchr_slot; }
const std::vector<std::string>& ctx_slot_names(const irf_ctx* c) { return c->slot_names; }
int ctx_fail(irf_ctx* c, int code, const std::string& msg) { return c->fail(code, msg); }
}   // namespace irf
