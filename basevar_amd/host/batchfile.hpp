// batchfile.hpp -- the data format on the INPUT side of the path: what a row holds, how a
// batchfile is written, how its header is read.
//
// The reference's `BaseVarBatchFile_v1.0` (written at src/basetype_caller.cpp:821-825, 1080-1086; read at :586-611): bgzip
// text, header
//     ##fileformat=BaseVarBatchFile_v1.0
//     ##SampleIDs=a,b,c
//     #CHROM POS REF Depth(CoveredSample) MappingQuality Readbases ReadbasesQuality ReadPositionRank Strand
// then one row per position, 9 tab-separated columns, the five per-sample columns being space-separated with one token per
// sample, e.g.
//     chr11 \t 5246595 \t N \t 1 \t 37 0 0 \t C N N \t A ! ! \t 2 0 0 \t + . .
// One site = one row from EVERY batchfile (each holds a slice of the samples), concatenated in batchfile order.
//
// Header-only, plain C++17, no htslib.  ROWS ARE READ by batchfile_fast.hpp (a byte-level reader straight into the slab's planes);
// this file holds the row record, the writer and the text helpers of the output side.  A literal restatement of the reference's
// reader lives in literal_reader.hpp outside the product tree (test infrastructure) as the checker for both (tests/cpp/host_formats_check.cpp).
//
// PARITY STATUS: number formatting is pinned against the reference's own compiled join() in
// tests/test_host_formats.py; whole-file parity with a run of the reference binary is unpinned.
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace bvamd {

// One site's per-sample columns: the interface SlabBuilder::add_site packs from (the fields of the reference's
// struct BatchInfo, src/basetype.h:25-43)
struct BatchInfo {
    size_t n = 0;
    std::string ref_id;
    std::string ref_base;
    uint32_t ref_pos = 0;
    uint32_t depth = 0;
    std::vector<std::string> align_bases;
    std::vector<char> align_base_quals;
    std::vector<int> mapqs;
    std::vector<char> map_strands;
    std::vector<int> base_pos_ranks;
};

// ---- text helpers.  One item's text: strings and characters as they are, integers in decimal, doubles with six significant
// digits ("%g" -- what the reference's ostringstream formatting produces for CM_AF / CM_CAF / <group>_AF, pinned by the tests).
inline void append_item(std::string &s, const std::string &v) { s += v; }
inline void append_item(std::string &s, char v) { s.push_back(v); }
inline void append_item(std::string &s, int v) { s += std::to_string(v); }
inline void append_item(std::string &s, unsigned v) { s += std::to_string(v); }
inline void append_item(std::string &s, double v) {
    char buf[32];
    std::snprintf(buf, sizeof buf, "%g", v);
    s += buf;
}
// the items of [first, last) with `delim` between them
template <class It>
inline std::string join(It first, It last, const char *delim) {
    std::string s;
    for (It it = first; it != last; ++it) {
        if (it != first) s += delim;
        append_item(s, *it);
    }
    return s;
}
template <class T>
inline std::string join(const std::vector<T> &v, const char *delim = "\t") { return join(v.begin(), v.end(), delim); }
// the pieces of `in` between occurrences of the character `delim` (an empty input is one empty piece)
inline std::vector<std::string> pieces(const std::string &in, char delim) {
    std::vector<std::string> out;
    size_t start = 0;
    for (;;) {
        const size_t i = in.find(delim, start);
        if (i == std::string::npos) { out.push_back(in.substr(start)); return out; }
        out.push_back(in.substr(start, i - start));
        start = i + 1;
    }
}

inline std::string batchfile_header(const std::vector<std::string> &sample_ids) {  // caller.cpp:821-825
    return "##fileformat=BaseVarBatchFile_v1.0\n##SampleIDs=" + join(sample_ids, ",") +
           "\n#CHROM\tPOS\tREF\tDepth(CoveredSample)\tMappingQuality\tReadbases\tReadbasesQuality\tReadPositionRank\tStrand\n";
}

// sample ids of one batchfile from its header lines (src/basetype_caller.cpp:637-665): the text behind the first '=' of a
// "##SampleIDs=" line, cut at commas
inline bool parse_sample_ids(const std::string &header_line, std::vector<std::string> &ids) {
    static const char key[] = "##SampleIDs=";
    if (header_line.compare(0, sizeof key - 1, key) != 0) return false;
    std::string list = header_line.substr(sizeof key - 1);
    const size_t eq = list.find('=');  // (the reference cuts the line at every '=' and takes the second piece)
    if (eq != std::string::npos) list.resize(eq);
    for (auto &id : pieces(list, ',')) ids.push_back(std::move(id));
    return true;
}

// One row of one batchfile: samples [first, first + count) of the site (the reference's row layout, caller.cpp:1080-1086)
inline std::string format_batchfile_row(const BatchInfo &bi, size_t first, size_t count, uint32_t covered) {
    std::string s = bi.ref_id;
    s += '\t'; s += std::to_string(bi.ref_pos);
    s += '\t'; s += bi.ref_base;
    s += '\t'; s += std::to_string(covered);
    s += '\t'; s += join(bi.mapqs.begin() + first, bi.mapqs.begin() + first + count, " ");
    s += '\t'; s += join(bi.align_bases.begin() + first, bi.align_bases.begin() + first + count, " ");
    s += '\t'; s += join(bi.align_base_quals.begin() + first, bi.align_base_quals.begin() + first + count, " ");
    s += '\t'; s += join(bi.base_pos_ranks.begin() + first, bi.base_pos_ranks.begin() + first + count, " ");
    s += '\t'; s += join(bi.map_strands.begin() + first, bi.map_strands.begin() + first + count, " ");
    s += '\n';
    return s;
}

}  // namespace bvamd
