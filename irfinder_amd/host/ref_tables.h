// ref_tables.h — flat, sorted reference tables parsed from the four IRFinder
// reference files.  Replaces the loadRef methods of ⟨upstream⟩
// ReadBlockProcessor.cpp (JunctionCount, SpansPoint, FragmentsInROI) and
// ReadBlockProcessor_CoverageBlocks.cpp (CoverageBlocks) — upstream builds
// per-chromosome std::map / vector objects; here everything is structure-of-
// arrays so it can be uploaded to the device as is.  Formats: SURVEY App. B.2.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace irf {

struct RefTables {
    // chromosome names in order of first appearance over the four files
    std::vector<std::string> chr_names;
    std::unordered_map<std::string, uint32_t> chr_index;

    // ref-cover.bed, file order
    std::vector<uint32_t> in_chr, in_start, in_end;
    std::vector<char> in_strand;
    std::vector<std::string> in_name, in_score;
    std::vector<uint32_t> in_blk_off;             // n_intron + 1
    std::vector<uint32_t> blk_start, blk_end;     // absolute coordinates, file order within a record
    std::vector<uint64_t> in_measured;            // sum of block sizes

    // ref-sj.ref, sorted unique by (chr,left,right); dir bits 1 '-', 2 '+'
    std::vector<uint32_t> kj_chr, kj_left, kj_right;
    std::vector<uint8_t> kj_dir;

    // ref-read-continues.ref, sorted unique by (chr,pos)
    std::vector<uint32_t> sp_chr, sp_pos;

    // ref-ROI.bed, file order
    std::vector<uint32_t> roi_chr, roi_start, roi_end;
    std::vector<std::string> roi_name;

    uint32_t chr_id(const std::string& name);     // inserts if new
    size_t n_intron() const { return in_start.size(); }
};

// Each parser appends to `t`; returns false and fills err on malformed input.
bool parse_cover_bed(const char* buf, size_t len, RefTables& t, std::string& err);
bool parse_sj_ref(const char* buf, size_t len, RefTables& t, std::string& err);
bool parse_read_continues_ref(const char* buf, size_t len, RefTables& t, std::string& err);
bool parse_roi_bed(const char* buf, size_t len, RefTables& t, std::string& err);
// sorts / uniques the junction and point tables after all files are parsed
void finish_tables(RefTables& t);

bool read_file(const char* path, std::string& out, std::string& err);

}   // namespace irf
