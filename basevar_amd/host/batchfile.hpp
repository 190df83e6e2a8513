// batchfile.hpp -- the data format on the INPUT side of the path.
//
// The reference's `BaseVarBatchFile_v1.0` (written at src/basetype_caller.cpp:821-825, 1080-1086;
// read at :586-611 and parsed at :688-736): bgzip text, header
//     ##fileformat=BaseVarBatchFile_v1.0
//     ##SampleIDs=a,b,c
//     #CHROM POS REF Depth(CoveredSample) MappingQuality Readbases ReadbasesQuality ReadPositionRank Strand
// then one row per position, 9 tab-separated columns, the five per-sample columns being
// space-separated with one token per sample, e.g.
//     chr11 \t 5246595 \t N \t 1 \t 37 0 0 \t C N N \t A ! ! \t 2 0 0 \t + . .
// One site = one row from EVERY batchfile (each holds a slice of the samples), concatenated in
// batchfile order.
//
// Header-only, plain C++17, no htslib: rows arrive as std::string (the caller decompresses; the
// `bv_call` tool uses zlib, which reads bgzip members transparently).
//
// PARITY STATUS: the tokenisers below restate ngslib::split (src/utils.cpp:81-99, src/utils.h:75-122)
// and are pinned against the reference's own compiled functions in tests/test_host_formats.py.
// The row layout is transcribed from the lines cited above; the full reference binary cannot be
// built under this round's rules (htslib needs generated config.h/version.h), so whole-file parity
// is not pinned by a reference run.
#pragma once

#include <cstdint>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace bvamd {

// struct BatchInfo, member for member (src/basetype.h:25-43)
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

// ngslib::split for std::string items (src/utils.cpp:81-99): every delimiter yields an item,
// empty items included; an empty input yields one empty item.
inline void split(const std::string &in, std::vector<std::string> &out, const char *delim, bool is_append = false) {
    if (!is_append) out.clear();
    const size_t dl = std::strlen(delim);
    size_t i = 0, start = 0;
    while (i != std::string::npos) {
        i = in.find(delim, start);
        const size_t len = (i == std::string::npos) ? in.length() - start : i - start;
        out.push_back(in.substr(start, len));
        start = i + dl;
    }
}

// ngslib::split<T> for arithmetic items (src/utils.h:87-122): each token goes through
// `istringstream >> T` (so a char column yields the token's first non-blank character and an
// int column stops at the first non-digit); an EMPTY token yields 0.
template <typename T>
inline void split(const std::string &in, std::vector<T> &out, const char *delim, bool is_append = false) {
    if (!is_append) out.clear();
    std::istringstream ss;
    const size_t dl = std::strlen(delim);
    size_t i = 0, start = 0;
    T d;
    while (i != std::string::npos) {
        ss.clear();
        i = in.find(delim, start);
        const size_t len = (i == std::string::npos) ? in.length() - start : i - start;
        const std::string tok = in.substr(start, len);
        if (!tok.empty()) {
            ss.str(tok);
            ss >> d;
            out.push_back(d);
        } else {
            out.push_back(0);
        }
        start = i + dl;
    }
}

// ngslib::tostring / join (src/utils.h:38-43, 75-85): ostringstream default formatting, i.e.
// 6 significant digits for double, the character itself for char.
template <typename T>
inline std::string tostring(const T &v) {
    std::ostringstream ss;
    ss << v;
    return ss.str();
}
template <typename T>
inline std::string join(const std::vector<T> &v, const std::string &delim = "\t") {
    if (v.empty()) return "";
    std::string s = tostring(v[0]);
    for (size_t i = 1; i < v.size(); ++i) s += delim + tostring(v[i]);
    return s;
}

inline std::string batchfile_header(const std::vector<std::string> &sample_ids) {  // caller.cpp:821-825
    return "##fileformat=BaseVarBatchFile_v1.0\n##SampleIDs=" + join(sample_ids, ",") +
           "\n#CHROM\tPOS\tREF\tDepth(CoveredSample)\tMappingQuality\tReadbases\tReadbasesQuality\tReadPositionRank\tStrand\n";
}

// sample ids of one batchfile from its header lines (src/basetype_caller.cpp:637-665)
inline bool parse_sample_ids(const std::string &header_line, std::vector<std::string> &ids) {
    const std::string key = "##SampleIDs=";
    if (header_line.compare(0, key.size(), key) != 0) return false;
    std::vector<std::string> h;
    split(header_line, h, "=");
    if (h.size() < 2) return false;
    std::vector<std::string> part;
    split(h[1], part, ",");
    ids.insert(ids.end(), part.begin(), part.end());
    return true;
}

// One row of one batchfile (the reference's __write_record_to_batchfile, caller.cpp:1080-1086)
inline std::string format_batchfile_row(const BatchInfo &bi, size_t first, size_t count, uint32_t covered) {
    std::vector<int> mq(bi.mapqs.begin() + first, bi.mapqs.begin() + first + count);
    std::vector<std::string> bases(bi.align_bases.begin() + first, bi.align_bases.begin() + first + count);
    std::vector<char> quals(bi.align_base_quals.begin() + first, bi.align_base_quals.begin() + first + count);
    std::vector<int> ranks(bi.base_pos_ranks.begin() + first, bi.base_pos_ranks.begin() + first + count);
    std::vector<char> strands(bi.map_strands.begin() + first, bi.map_strands.begin() + first + count);
    return bi.ref_id + "\t" + std::to_string(bi.ref_pos) + "\t" + bi.ref_base + "\t" + std::to_string(covered) + "\t" +
           join(mq, " ") + "\t" + join(bases, " ") + "\t" + join(quals, " ") + "\t" + join(ranks, " ") + "\t" +
           join(strands, " ") + "\n";
}

// The text half of _basevar_caller (src/basetype_caller.cpp:688-736): one row from each batchfile
// for the same position -> BatchInfo over all n_sample samples.  Returns false for the rows the
// reference skips (total Depth == 0, :718).  Same errors, same messages.
inline bool parse_site_rows(const std::vector<std::string> &rows, size_t n_sample, BatchInfo &bi) {
    bi = BatchInfo();
    bi.align_bases.reserve(n_sample);
    bi.align_base_quals.reserve(n_sample);
    bi.mapqs.reserve(n_sample);
    bi.map_strands.reserve(n_sample);
    bi.base_pos_ranks.reserve(n_sample);
    bi.n = n_sample;
    std::vector<std::string> col;
    for (size_t i = 0; i < rows.size(); ++i) {
        split(rows[i], col, "\t");
        if (col.size() != 9) throw std::runtime_error("[ERROR] batchfile has invalid data:\n" + rows[i]);
        if (i == 0) {
            bi.ref_id = col[0];
            bi.ref_pos = (uint32_t)std::stoi(col[1]);
            bi.ref_base = col[2];
        } else if (bi.ref_id != col[0] || bi.ref_pos != (uint32_t)std::stoi(col[1]) || bi.ref_base != col[2]) {
            throw std::runtime_error("[ERROR] Batchfiles must have the same genome coordinate in each line.");
        }
        bi.depth += (uint32_t)std::stoi(col[3]);
        split(col[4], bi.mapqs, " ", true);
        split(col[5], bi.align_bases, " ", true);
        split(col[6], bi.align_base_quals, " ", true);
        split(col[7], bi.base_pos_ranks, " ", true);
        split(col[8], bi.map_strands, " ", true);
    }
    if (bi.depth == 0) return false;
    if (bi.mapqs.size() != n_sample || bi.align_bases.size() != n_sample || bi.align_base_quals.size() != n_sample ||
        bi.map_strands.size() != n_sample || bi.base_pos_ranks.size() != n_sample)
        throw std::runtime_error("[ERROR] Something is wrong in batchfiles.");
    return true;
}

}  // namespace bvamd
