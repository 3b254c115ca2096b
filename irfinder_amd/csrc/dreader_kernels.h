// dreader_kernels.h — launch interface of the device reader (SURVEY §8 f N1): BGZF inflate (bgzf.hip) and BAM record
// location + fragment assembly (bamparse.hip).  Stands where ⟨upstream⟩ bin/IRFinder pipes `gzip -cd` into
// BAM2blocks::processAll: here the host hands over compressed BGZF members and everything up to the structure-of-arrays
// fragment batch happens in HBM.  Every launcher is asynchronous on the given stream.
#pragma once
#include <hip/hip_runtime.h>
#include "wg_barrier.h"

#include <cstdint>

namespace irf {
namespace dev {

// ---- BGZF ------------------------------------------------------------------------------------------------------
// One complete member (gzip member with a 'BC' subfield), located by the host (bam_reader.cpp: bgzf_locate).
struct BgzfMember {
    uint64_t in_off;     // first byte of the raw DEFLATE data inside the compressed buffer
    uint64_t out_off;    // where its inflated bytes go (prefix sum of isize)
    uint32_t clen;       // bytes of DEFLATE data
    uint32_t isize;      // inflated size from the trailer (<= 65536)
    uint32_t crc;        // CRC-32 from the trailer
    uint32_t pad;
};
enum BgzfStatus : uint32_t { BGZF_OK = 0, BGZF_BAD_DEFLATE = 1, BGZF_BAD_CRC = 2 };
constexpr uint32_t kInflateSlack = 1024; // readable bytes the compressed buffer must have behind its last member (a corrupt
                                         // dynamic-block header is read to its end, <= 600 bytes, before the input limit is checked)

// status[m] = BGZF_OK / BGZF_BAD_DEFLATE.  `in` must be readable kInflateSlack bytes past the last member.
void launch_bgzf_inflate(const uint8_t* in, const BgzfMember* members, uint32_t n_members, uint8_t* out, uint32_t* status,
                         hipStream_t s);
// members the current device inflates at once (one lane each; CUs x resident workgroups x 64): a launch with more runs
// in rounds of that many, and a round lasts as long as one lane needs for its member however few members it holds
uint32_t bgzf_inflate_slots();
// CRC-32 of every inflated member against its trailer: status[m] = BGZF_BAD_CRC where it differs (and status was OK)
void launch_bgzf_crc(const uint8_t* out, const BgzfMember* members, uint32_t n_members, uint32_t* status, hipStream_t s);
// first member with a non-zero status: first_bad[0] = min index (pre-set to 0xffffffff)
void launch_first_bad(const uint32_t* status, uint32_t n_members, uint32_t* first_bad, hipStream_t s);

// ---- BAM records ------------------------------------------------------------------------------------------------
constexpr uint32_t kChainSegment = 4096;            // bytes of the inflated stream per chain walker
constexpr uint64_t kNoRecord = ~0ull;
constexpr uint32_t kMaxBamBlockSize = 1u << 28;     // same bound as the host reader
constexpr uint32_t kPairTile = 1024;                // records per workgroup of the pairing / assembly kernels

// Record starts.  A BAM record is found only by walking block_size fields from a known start, which is a sequential
// chain; here every kChainSegment-byte segment guesses the first record start at or after its beginning (header
// plausibility + two chained records), walks to its end, and the guesses are then VERIFIED against the walks of the
// predecessors: guess[i+1] must equal exit[i], guess[0] is the known start.  Where it differs the walker of segment i
// carries on through the following segments until it meets a guess it confirms.  When a launch of chain_verify changes
// nothing, guess[] is the true chain by induction — the heuristic decides speed, never the result.
struct ChainBufs {
    const uint8_t* buf;      // inflated stream
    uint64_t start;          // first record start (a true one)
    uint64_t total;          // bytes in buf
    uint32_t n_seg;
    int32_t n_ref;
    uint64_t* guess;         // [n_seg]   first record start >= segment begin (kNoRecord: none / chain broken)
    uint64_t* exit_;         // [n_seg]   first record start >= segment end reached from guess (or the partial tail)
    uint32_t* count;         // [n_seg+1] complete records that start in the segment (last entry stays 0 for the scan)
    uint32_t* changed;       // [1]
};
void launch_chain_guess(const ChainBufs& C, hipStream_t s);
void launch_chain_verify(const ChainBufs& C, hipStream_t s);

struct RecordIndex {
    uint64_t* rec_off;       // [n_rec] offset of the record's block_size field
    uint32_t* rec_nf;        // [n_rec] n_cigar_op | flag << 16
    unsigned long long* err; // [1] min over bad records of (offset << 2 | code), pre-set to ~0: 1 bad block_size, 2 fields overrun
};
// seg_base = exclusive scan of ChainBufs::count
void launch_chain_emit(const ChainBufs& C, const uint32_t* seg_base, const RecordIndex& R, hipStream_t s);

enum ReaderStat { RS_RECORDS = 0, RS_NUCLEOTIDES, RS_SINGLES, RS_PAIRS, RS_SHORT, RS_INTERSECT, RS_LONG, RS_SKIPPED, RS_ERROR, RS_COUNT };

struct AssembleArgs {
    const uint8_t* buf;
    uint32_t n_rec;
    int32_t n_ref;
    RecordIndex R;
    const uint32_t* tile_carry;   // [tiles] (index of the last unpaired record before the tile) + 1, 0: none
    uint32_t skip_flag_mask, pair_long_inclusive, pair_short_inclusive, gap_pair_is_intersect;
    uint32_t* fr_cnt;             // [n_rec+1] fragments the record completes (0/1); after the scan: their offset
    uint32_t* bl_cnt;             // [n_rec+1] aligned blocks of that fragment; after the scan: their offset
    unsigned long long* stats;    // [RS_COUNT]
    uint32_t* dangling;           // [1] 1: the last record opens a pair whose mate is not in this buffer (not consumed)
    const uint32_t* mate;         // name-joined pairing (params.mate_join): per record a JoinCode or the index of its first mate;
                                  // nullptr: upstream's pairing (the next record is the mate)
    // outputs of the write pass (irf_batch columns)
    uint32_t* frag_chr; uint8_t* frag_flags; uint32_t* frag_start; uint32_t* frag_end; uint32_t* frag_blk_off;
    uint32_t* blk_start; uint32_t* blk_len; uint8_t* blk_read;
};
// ---- name-joined pairing (params.mate_join = 1: coordinate-sorted input) -------------------------
// host/bam_reader.cpp: on_paired_by_name, on the device.  A paired record waits under its read name until the record
// of the OTHER mate (flag 0x40 / 0x80) arrives; a second record of the SAME mate displaces the waiting one (an error
// read).  Per feed: a 64-bit hash of every paired record's name, a stable radix sort of (hash, record index), then one
// thread per group of equal hashes walks its records in arrival order (names compared byte for byte: two names with
// one hash are reported, not mis-paired).  Records still waiting at the end of a feed are copied, whole, in front of
// the next feed's inflated bytes and take part again as its first n_old records.
enum JoinCode : uint32_t {
    JOIN_SINGLE = 0xfffffff0u,    // not a paired record: a fragment of its own
    JOIN_FIRST = 0xfffffff1u,     // first mate of a pair completed by a later record (no output of its own)
    JOIN_WAITING = 0xfffffff2u,   // its mate has not arrived: carried to the next feed
    JOIN_DEAD = 0xfffffff3u       // displaced by a second record of the same mate (counted as an error read)
};
constexpr unsigned long long kJoinNoKey = ~0ull;
struct JoinArgs {
    const uint8_t* buf;
    uint32_t n_rec, n_old;        // records [0, n_old) were carried over: counted when they first arrived
    RecordIndex R;
    unsigned long long* keys;     // [n_rec] name hash (kJoinNoKey: not paired)
    uint32_t* vals;               // [n_rec] record index
    uint32_t* mate;               // [n_rec] JoinCode / first-mate index
    unsigned long long* stats;    // [RS_COUNT]
    uint32_t* collision;          // [1] set when two different names share a hash
    uint32_t* n_waiting;          // [1] records still waiting for their mate (join_sizes adds them up)
    uint32_t* sizes;              // [n_rec + 1] bytes of the record if it is carried, else 0 (then scanned)
};
void launch_join_keys(const JoinArgs& J, hipStream_t s);
// keys / vals sorted by key (stable): fills mate[]
void launch_join_groups(const JoinArgs& J, hipStream_t s);
void launch_join_sizes(const JoinArgs& J, hipStream_t s);
// offsets = exclusive scan of sizes: copies the waiting records, in arrival order, to dst
void launch_join_carry(const JoinArgs& J, const uint32_t* offsets, uint8_t* dst, hipStream_t s);

uint32_t pair_tiles(uint32_t n_rec);
// (index of the last record without the paired flag in each tile) + 1, then the exclusive running maximum over tiles
void launch_pair_state(const uint32_t* rec_nf, uint32_t n_rec, uint32_t* tile_last, uint32_t* tile_carry, hipStream_t s);
// ⟨upstream⟩ BAM2blocks::processAll / processPair / processSingle per record: counts, then (after the scans) writes
void launch_assemble_count(const AssembleArgs& A, hipStream_t s);
void launch_assemble_write(const AssembleArgs& A, uint32_t n_frag, uint32_t n_blk, hipStream_t s);

}   // namespace dev
}   // namespace irf
