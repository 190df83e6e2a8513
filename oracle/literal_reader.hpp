// literal_reader.hpp -- TEST INFRASTRUCTURE, part of the oracle (tests/ only; nothing under basevar_amd/ includes it).
//
// A literal restatement of the reference's text layer on the input side of the path, kept as the CHECKER for the product's own
// byte-level reader (basevar_amd/host/batchfile_fast.hpp) and formatters (basevar_amd/host/batchfile.hpp, vcf_emit.hpp):
//   * ngslib::split for std::string items          (reference src/utils.cpp:81-99)
//   * ngslib::split<T> for arithmetic items        (reference src/utils.h:87-122): `istringstream >> T` per token, 0 for an empty one
//   * ngslib::tostring / join                      (reference src/utils.h:38-43, 75-85): ostringstream default formatting
//   * the text half of _basevar_caller             (reference src/basetype_caller.cpp:688-736): one row from each batchfile of a
//                                                    position -> BatchInfo, same skips, same exceptions, same messages
// The tokenisers are themselves pinned against the reference's compiled functions (oracle/_ref, tests/cpp/host_formats_check.cpp
// part 1).  Whole-file parity with a run of the reference binary is unpinned (the binary cannot be built here: DESIGN.md).
#pragma once

#include <cstdint>
#include <cstring>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../basevar_amd/host/batchfile.hpp"

namespace bvlit {

using bvamd::BatchInfo;

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

}  // namespace bvlit
