// batchfile_fast.hpp -- batchfile rows -> slab row, byte level.
//
// The reference reads a position's rows with ngslib::split per column, one std::istringstream extraction per token
// (src/utils.h:87-122) and a BatchInfo of strings and vectors per site (src/basetype_caller.cpp:688-736) -- 49 k rows/s at
// 100 samples per row on one core, and what would bound any real run of bv_call by orders of magnitude.  The function below
// walks the same bytes once and writes the slab row directly: same accepted inputs, same results, same errors (message and
// precedence), pinned against a literal restatement of the reference's reader (literal_reader.hpp, test infrastructure outside the product tree: test
// infrastructure) by tests/cpp/host_formats_check.cpp on valid rows, ragged rows and malformed tokens.
//
// Token semantics kept from the reference's readers:
//   int columns (MappingQuality, ReadPositionRank)   `istringstream >> int` on the token: optional sign, digits, stops at the
//                                                    first other character; no digits -> 0; an EMPTY token -> 0
//   char columns (ReadbasesQuality, Strand)          the token's first character; an EMPTY token -> '\0'
//   Readbases                                        the token itself (first character decides; '+'/'-' tokens are kept as text
//                                                    for the CVG indel column)
#pragma once

#include <climits>
#include <cstring>

#include "basetype_gpu.hpp"
#include "batchfile.hpp"
#include "vcf_emit.hpp"

namespace bvamd {

// `istringstream >> int` on [p, e): the characters are a token of a space-split column, so there is no leading blank
inline int parse_int_token(const char *p, const char *e) {
    if (p == e) return 0;
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = (*p == '-'); ++p; }
    if (p == e || *p < '0' || *p > '9') return 0;  // extraction fails: the value is set to 0 (C++11)
    long long v = 0;
    for (; p != e && *p >= '0' && *p <= '9'; ++p) {
        v = v * 10 + (*p - '0');
        if (v > (long long)INT_MAX + 1) { v = (long long)INT_MAX + 1; }  // keep walking the digits, saturated
    }
    if (neg) return v > (long long)INT_MAX ? INT_MIN : (int)-v;
    return v > (long long)INT_MAX ? INT_MAX : (int)v;
}

// One row from every batchfile for the same position -> one slab row + its SiteText.  Returns false for the rows the reference
// skips (total Depth == 0, caller.cpp:718): nothing is added then.  Throws what the reference's reader and BaseType constructor throw (as restated in literal_reader.hpp, test infrastructure outside the product tree + SlabBuilder::add_site).
inline bool parse_site_rows_fast(const std::vector<std::string> &rows, size_t n_sample, SlabBuilder &sb, SiteText &st) {
    st = SiteText();
    SlabBuilder::Row r = sb.begin_row();
    struct Drop {  // the row stays only if commit() is reached
        SlabBuilder &sb; bool keep = false;
        ~Drop() { if (!keep) sb.drop_row(); }
    } guard{sb};
    size_t n_mq = 0, n_base = 0, n_qual = 0, n_rank = 0, n_strand = 0;
    uint32_t depth = 0;
    // What add_site would refuse, in the reference's order: the strand check of EVERY sample comes before the base tokens' checks
    // (strand_bias runs from _out_cvg_line, ahead of the BaseType constructor: basetype_caller.cpp:738-743), each in sample order.
    size_t err_sample = (size_t)-1;
    std::string err_msg;
    auto token_error = [&](size_t sample, int order, const std::string &m) {
        // order: 0 = the base token's checks (the constructor), 1 = the strand check (strand_bias: first)
        const size_t key = sample + (order == 1 ? 0 : ((size_t)1 << 62));
        if (key < err_sample) { err_sample = key; err_msg = m; }
    };
    // samples whose base token was refused: their cell stays 'N', but strand_bias looks at their strand all the same (the token
    // does not START with N / + / -); ascending, consumed by the strand column's walk of the same row
    std::vector<size_t> refused;
    size_t refused_at = 0;
    for (size_t i = 0; i < rows.size(); ++i) {
        const std::string &row = rows[i];
        const char *col[10];
        int nc = 0;
        col[nc++] = row.data();
        const char *end = row.data() + row.size();
        for (const char *p = row.data(); (p = (const char *)std::memchr(p, '\t', (size_t)(end - p))) != nullptr;) {
            ++p;
            if (nc <= 9) col[nc] = p;
            ++nc;
        }
        if (nc != 9) throw std::runtime_error("[ERROR] batchfile has invalid data:\n" + row);
        col[9] = end + 1;
        auto field = [&](int k) { return std::string(col[k], (size_t)(col[k + 1] - 1 - col[k])); };
        if (i == 0) {
            st.ref_id = field(0);
            st.ref_pos = (uint32_t)std::stoi(field(1));
            st.ref_base = field(2);
        } else if (st.ref_id != field(0) || st.ref_pos != (uint32_t)std::stoi(field(1)) || st.ref_base != field(2)) {
            throw std::runtime_error("[ERROR] Batchfiles must have the same genome coordinate in each line.");
        }
        depth += (uint32_t)std::stoi(field(3));
        // the five per-sample columns: one token per delimiter, empty tokens included (ngslib::split)
        // (tokens are one to three characters: a byte loop finds their ends several times faster than a memchr call per token)
        auto walk = [&](int k, auto &&fn) {
            const char *p = col[k], *e = col[k + 1] - 1;
            for (;;) {
                const char *t = p;
                while (t != e && *t != ' ') ++t;
                fn(p, t);
                if (t == e) break;
                p = t + 1;
            }
        };
        walk(4, [&](const char *p, const char *e) { if (n_mq < n_sample) r.mapq[n_mq] = (uint8_t)parse_int_token(p, e); ++n_mq; });
        walk(5, [&](const char *p, const char *e) {
            if (n_base < n_sample) {
                const char fb = p == e ? '\0' : *p;  // (an EMPTY token: its [0] is the terminator, and size() != 1 below -- the reference's error)
                uint8_t cell = BV_CELL_N;
                if (fb == 'N') cell = BV_CELL_N;
                else if (fb == '+' || fb == '-') {
                    cell = fb == '+' ? BV_CELL_INS : BV_CELL_DEL;
                    st.indel_tokens.emplace_back(p, (size_t)(e - p));
                } else if (e - p != 1) {  // src/basetype.cpp:54-56
                    token_error(n_base, 0, "[ERROR] Why dose the size of aligned base is not 1? Check: " + std::string(p, (size_t)(e - p)));
                    refused.push_back(n_base);
                } else {
                    const int c = base_code(fb);
                    if (c == BV_BASE_OTHER) {
                        token_error(n_base, 0, std::string("[ERROR] base character '") + fb +
                                                   "' is outside ACGTN+-: not representable in the slab (the reference would "
                                                   "count it in the depth)");
                        refused.push_back(n_base);
                    } else cell = (uint8_t)c;  // strand added below
                }
                r.cell[n_base] = cell;
            }
            ++n_base;
        });
        walk(6, [&](const char *p, const char *e) { if (n_qual < n_sample) r.phred[n_qual] = (uint8_t)((p == e ? '\0' : *p) - 33); ++n_qual; });
        walk(7, [&](const char *p, const char *e) { if (n_rank < n_sample) r.rank[n_rank] = (uint16_t)parse_int_token(p, e); ++n_rank; });
        walk(8, [&](const char *p, const char *e) {
            const bool was_refused = refused_at < refused.size() && refused[refused_at] == n_strand;
            if (was_refused) ++refused_at;
            if (n_strand < n_sample && n_strand < n_base && (was_refused || !(r.cell[n_strand] & BV_CELL_NOCALL))) {
                const char s = p == e ? '\0' : *p;
                if (s == '-') { if (!was_refused) r.cell[n_strand] |= BV_CELL_REV; }
                else if (s != '+') token_error(n_strand, 1, std::string("[ERROR] Get strange strand symbol: ") + s);  // src/basetype.cpp:272
            }
            ++n_strand;
        });
    }
    if (depth == 0) return false;
    if (n_mq != n_sample || n_base != n_sample || n_qual != n_sample || n_strand != n_sample || n_rank != n_sample)
        throw std::runtime_error("[ERROR] Something is wrong in batchfiles.");
    if (err_sample != (size_t)-1) throw std::runtime_error(err_msg);
    sb.commit_row((uint8_t)base_code((char)std::toupper((unsigned char)(st.ref_base.empty() ? 'N' : st.ref_base[0]))));
    guard.keep = true;
    return true;
}

}  // namespace bvamd
