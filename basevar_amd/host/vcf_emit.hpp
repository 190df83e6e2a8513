// vcf_emit.hpp -- the data format on the OUTPUT side of the path.
//
// Text of one VCF record (_out_vcf_line, src/basetype_caller.cpp:1103-1209) and one CVG row
// (_out_cvg_line, :1211-1260; __base_depth_and_indel, :1263-1289) from the engine's per-site
// record plus the site's slab row (the per-sample GT:AB:SO:BP strings need the call, strand and
// phred of every sample -- exactly the two byte planes the engine was given) and its few indel
// tokens (the CVG row's indel tally is a text operation on them), and the two file headers
// (src/basetype_utils.cpp:32-88).  A host keeps, per pending site, a SiteText and nothing else.
//
// Number formatting is the reference's: std::to_string(double) ("%f", 6 decimals) for QUAL, QD,
// FS, SOR, BP ("%f"); six significant digits ("%g", the reference's ostringstream formatting) for CM_AF, CM_CAF and
// the group AFs; the three rank sums truncated to int (caller.cpp:1151-1157).
//
// PARITY STATUS: the numeric inputs are the pinned bv_site_result fields; the two number formats are
// pinned against the reference's own compiled ngslib functions (tests/test_host_formats.py); the field order
// and literals are transcribed from the cited lines -- whole-line parity is not pinned by a run of
// the reference binary (not buildable under this round's rules).
#pragma once

#include <cctype>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "../../include/basevar_amd.h"
#include <cstddef>
#include <cstdio>

#include "batchfile.hpp"

namespace bvamd {

static const char EMIT_BASES[4] = {'A', 'C', 'G', 'T'};  // src/basetype.h:19
static const double EMIT_MLN10TO10 = -0.23025850929940458;  // src/basetype.h:20
static const int EMIT_QUAL_THRESHOLD = 20;                  // src/basetype.h:22

// What the text of one site needs besides the engine's record and the site's slab row: where it is, and the few
// indel tokens ("+AT", "-CAG": one per sample that carries one) for the CVG row's indel tally.
struct SiteText {
    std::string ref_id, ref_base;
    uint32_t ref_pos = 0;
    std::vector<std::string> indel_tokens;
};

// The CVG row's indel column (what the reference's __base_depth_and_indel tallies, caller.cpp:1263-1289): "TOKEN|count" of the
// '+' / '-' tokens in lexicographic order, comma-separated; "." if the site has none.
inline std::string indel_string(const std::vector<std::string> &tokens) {
    std::map<std::string, int> tally;
    for (const auto &t : tokens)
        if (!t.empty() && t[0] != 'N' && t[0] != 'A' && t[0] != 'C' && t[0] != 'G' && t[0] != 'T') ++tally[t];
    if (tally.empty()) return ".";
    std::string s;
    for (const auto &kv : tally) {
        if (!s.empty()) s += ',';
        s += kv.first; s += '|'; s += std::to_string(kv.second);
    }
    return s;
}

// "%f" of a double (what std::to_string gives: QUAL, QD, FS, SOR, BP), appended
inline void append_f6(std::string &s, double v) {
    char buf[352];
    std::snprintf(buf, sizeof buf, "%f", v);
    s += buf;
}

// One CVG row (the columns of _out_cvg_line, caller.cpp:1246-1257).  Empty string when the reference writes nothing.
inline std::string format_cvg_line(const SiteText &st, const bv_site_result &r) {
    if (r.total_depth == 0) return "";
    std::string s = st.ref_id;
    s += '\t'; s += std::to_string(st.ref_pos);
    s += '\t'; s += st.ref_base;
    s += '\t'; s += std::to_string((int)r.total_depth);
    for (int b = 0; b < 4; ++b) { s += '\t'; s += std::to_string((int)r.depth[b]); }
    s += '\t'; s += indel_string(st.indel_tokens);
    s += '\t'; append_f6(s, r.cvg_fs);
    s += '\t'; append_f6(s, r.cvg_sor);
    s += '\t';
    for (int k = 0; k < 4; ++k) { if (k) s += ','; s += std::to_string((int)r.cvg_sb[k]); }
    s += '\n';
    return s;
}

// The INFO column as a table: key, and what of the record it prints.  Order and spelling are the file format's
// (caller.cpp:1167-1180); the three rank sums are truncated to int as the reference's caller does (:1151-1157).
enum InfoKind { INFO_TOTAL_DEPTH, INFO_ALT_DEPTHS, INFO_ALT_AF, INFO_ALT_CAF, INFO_INT_OF_DOUBLE, INFO_F6, INFO_INT_PAIR };
struct InfoField {
    const char *key;
    InfoKind kind;
    size_t offset;  // of the double / of the first of the two counts in bv_site_result (INFO_INT_OF_DOUBLE, INFO_F6, INFO_INT_PAIR)
};
static const InfoField EMIT_INFO[] = {
    {"CM_DP", INFO_TOTAL_DEPTH, 0},
    {"CM_AC", INFO_ALT_DEPTHS, 0},
    {"CM_AF", INFO_ALT_AF, 0},
    {"CM_CAF", INFO_ALT_CAF, 0},
    {"MQRankSum", INFO_INT_OF_DOUBLE, offsetof(bv_site_result, mq_ranksum)},
    {"ReadPosRankSum", INFO_INT_OF_DOUBLE, offsetof(bv_site_result, rpr_ranksum)},
    {"BaseQRankSum", INFO_INT_OF_DOUBLE, offsetof(bv_site_result, bq_ranksum)},
    {"QD", INFO_F6, offsetof(bv_site_result, qd)},
    {"SOR", INFO_F6, offsetof(bv_site_result, var_sor)},
    {"FS", INFO_F6, offsetof(bv_site_result, var_fs)},
    {"SB_REF", INFO_INT_PAIR, offsetof(bv_site_result, var_sb)},
    {"SB_ALT", INFO_INT_PAIR, offsetof(bv_site_result, var_sb) + 2 * sizeof(uint32_t)},
};
inline void append_info(std::string &s, const bv_site_result &r) {
    const char *raw = reinterpret_cast<const char *>(&r);
    bool first = true;
    for (const InfoField &f : EMIT_INFO) {
        if (!first) s += ';';
        first = false;
        s += f.key; s += '=';
        switch (f.kind) {
            case INFO_TOTAL_DEPTH: s += std::to_string((int)r.total_depth); break;
            case INFO_ALT_DEPTHS:
                for (int i = 0; i < r.n_alt; ++i) { if (i) s += ','; s += std::to_string((int)r.depth[r.alt[i] & 3]); }
                break;
            case INFO_ALT_AF:
                for (int i = 0; i < r.n_alt; ++i) { if (i) s += ','; append_item(s, r.af[i]); }
                break;
            case INFO_ALT_CAF:
                for (int i = 0; i < r.n_alt; ++i) { if (i) s += ','; append_item(s, r.caf[i]); }
                break;
            case INFO_INT_OF_DOUBLE: { double v; std::memcpy(&v, raw + f.offset, sizeof v); s += std::to_string((int)v); break; }
            case INFO_F6: { double v; std::memcpy(&v, raw + f.offset, sizeof v); append_f6(s, v); break; }
            case INFO_INT_PAIR: {
                uint32_t c[2]; std::memcpy(c, raw + f.offset, sizeof c);
                s += std::to_string((int)c[0]); s += ','; s += std::to_string((int)c[1]);
                break;
            }
        }
    }
}

// One VCF record (the columns of _out_vcf_line, caller.cpp:1103-1209) from the site's slab row: `cell` / `phred` are the n
// per-sample bytes of the base_strand and qual planes (include/basevar_amd.h).  `groups`/`group_names`: the site's
// bv_group_result records and the group names in the reference's iteration order (std::map: sorted by name); may be empty.
inline std::string format_vcf_line(const SiteText &st, const uint8_t *cell, const uint8_t *phred, size_t n, const bv_site_result &r,
                                   const bv_group_result *groups, const std::vector<std::string> &group_names) {
    if (r.n_alt == 0) return "";  // caller.cpp:745
    // FORMAT GT by base: "0/." for the reference base, "./k" for the k-th ALT, "./." for any other call (caller.cpp:1116, 1136-1143)
    const char upper_ref = (char)std::toupper((unsigned char)st.ref_base[0]);
    std::string gt_of[4];
    for (int b = 0; b < 4; ++b) gt_of[b] = EMIT_BASES[b] == upper_ref ? "0/." : "./.";
    for (int i = 0; i < r.n_alt; ++i)
        if (EMIT_BASES[r.alt[i] & 3] != upper_ref) gt_of[r.alt[i] & 3] = "./" + std::to_string(i + 1);
    // BP = std::to_string(1 - eps(q)) depends on the phred byte only: 256 strings, formed once (basetype.cpp:47-48)
    static const std::vector<std::string> bp_text = [] {
        std::vector<std::string> t(256);
        for (int qv = 0; qv < 256; ++qv) append_f6(t[(size_t)qv], 1.0 - std::exp(qv * EMIT_MLN10TO10));
        return t;
    }();
    std::string s = st.ref_id;
    s.reserve(128 + n * 18);
    s += '\t'; s += std::to_string(st.ref_pos);
    s += "\t.\t"; s += st.ref_base;
    s += '\t';
    for (int i = 0; i < r.n_alt; ++i) { if (i) s += ','; s += EMIT_BASES[r.alt[i] & 3]; }
    s += '\t'; append_f6(s, r.qual);
    s += (r.qual > EMIT_QUAL_THRESHOLD) ? "\t.\t" : "\tLowQual\t";
    append_info(s, r);
    if (groups && !group_names.empty()) {  // <group>_AF of the groups that have an ALT (caller.cpp:1182-1196)
        for (size_t g = 0; g < group_names.size(); ++g) {
            if (groups[g].n_alt == 0) continue;
            s += ';'; s += group_names[g]; s += "_AF=";
            for (int k = 0; k < groups[g].n_alt; ++k) { if (k) s += ','; append_item(s, groups[g].af[k]); }
        }
    }
    s += "\tGT:AB:SO:BP";
    // per-sample GT:AB:SO:BP (caller.cpp:1125-1145), appended directly: a VCF line is n_samples fields, and at 10^4 samples the
    // emitter, not the engine, sets the pace of a run (profiles/r3_host_pipeline.txt)
    for (size_t i = 0; i < n; ++i) {
        s += '\t';
        if (cell[i] & BV_CELL_NOCALL) { s += "./."; continue; }
        const int bc = cell[i] & 3;
        s += gt_of[bc]; s += ':'; s += EMIT_BASES[bc]; s += ':';
        s += (cell[i] & BV_CELL_REV) ? '-' : '+';
        s += ':'; s += bp_text[phred[i]];
    }
    s += '\n';
    return s;
}

// The same from a BatchInfo (the reference's per-site input): its tokens are packed into a slab row first.
inline SiteText site_text_of(const BatchInfo &bi) {
    SiteText st;
    st.ref_id = bi.ref_id; st.ref_base = bi.ref_base; st.ref_pos = bi.ref_pos;
    for (const auto &tok : bi.align_bases)
        if (!tok.empty() && (tok[0] == '+' || tok[0] == '-')) st.indel_tokens.push_back(tok);
    return st;
}
inline std::string format_cvg_line(const BatchInfo &bi, const bv_site_result &r) { return format_cvg_line(site_text_of(bi), r); }
inline std::string format_vcf_line(const BatchInfo &bi, const bv_site_result &r, const bv_group_result *groups,
                                   const std::vector<std::string> &group_names) {
    std::vector<uint8_t> cell(bi.n), phred(bi.n);
    for (size_t i = 0; i < bi.n; ++i) {
        const char fb = bi.align_bases[i].empty() ? 'N' : bi.align_bases[i][0];
        uint8_t c = BV_CELL_N;
        for (int k = 0; k < 4; ++k)
            if (fb == EMIT_BASES[k]) c = (uint8_t)(k | (bi.map_strands[i] == '-' ? BV_CELL_REV : 0));
        cell[i] = c;
        phred[i] = (uint8_t)(bi.align_base_quals[i] - 33);
    }
    return format_vcf_line(site_text_of(bi), cell.data(), phred.data(), bi.n, r, groups, group_names);
}

// cvg_header_define, src/basetype_utils.cpp:72-88
inline std::string cvg_header() {
    return "##fileformat=CVGv1.0\n##Group information is the depth of A:C:G:T:Indel\n"
           "#CHROM\tPOS\tREF\tDepth\tA\tC\tG\tT\tIndels\tFS\tSOR\tStrand_Coverage(REF_FWD,REF_REV,ALT_FWD,ALT_REV)";
}

// vcf_header_define, src/basetype_utils.cpp:32-70.  The reference reads contig names/lengths from
// the FASTA index; here they are passed in (the FASTA is ingest, out of scope).
struct Contig {
    std::string name;
    uint32_t length;
};
inline std::string vcf_header(const std::string &ref_file_path, const std::string &ref_abs_path,
                              const std::vector<Contig> &contigs, const std::vector<std::string> &addition_info,
                              const std::vector<std::string> &samples) {
    std::vector<std::string> header = {
        "##fileformat=VCFv4.2",
        "##FILTER=<ID=LowQual,Description=\"Low quality (QUAL < 60)\">",
        "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">",
        "##FORMAT=<ID=AB,Number=1,Type=String,Description=\"Allele Base\">",
        "##FORMAT=<ID=SO,Number=1,Type=String,Description=\"Strand orientation of the mapping base. Marked as + or -\">",
        "##FORMAT=<ID=BP,Number=1,Type=String,Description=\"Base Probability which calculate by base quality\">",
        "##INFO=<ID=CM_AF,Number=A,Type=Float,Description=\"An ordered, comma delimited list of allele frequencies base on LRT algorithm\">",
        "##INFO=<ID=CM_CAF,Number=A,Type=Float,Description=\"An ordered, comma delimited list of allele frequencies just base on read count\">",
        "##INFO=<ID=CM_AC,Number=A,Type=Integer,Description=\"An ordered, comma delimited allele depth in CMDB\">",
        "##INFO=<ID=CM_DP,Number=A,Type=Integer,Description=\"Total Depth in CMDB\">",
        "##INFO=<ID=SB_REF,Number=A,Type=Integer,Description=\"Read number support REF: Forward,Reverse\">",
        "##INFO=<ID=SB_ALT,Number=A,Type=Integer,Description=\"Read number support ALT: Forward,Reverse\">",
        "##INFO=<ID=FS,Number=1,Type=Float,Description=\"Phred-scaled p-value using Fisher's exact test to detect strand bias\">",
        "##INFO=<ID=BaseQRankSum,Number=1,Type=Float,Description=\"Phred-score from Wilcoxon rank sum test of Alt Vs. Ref base qualities\">",
        "##INFO=<ID=SOR,Number=1,Type=Float,Description=\"Symmetric Odds Ratio of 2x2 contingency table to detect strand bias\">",
        "##INFO=<ID=MQRankSum,Number=1,Type=Float,Description=\"Phred-score From Wilcoxon rank sum test of Alt vs. Ref read mapping qualities\">",
        "##INFO=<ID=ReadPosRankSum,Number=1,Type=Float,Description=\"Phred-score from Wilcoxon rank sum test of Alt vs. Ref read position bias\">",
        "##INFO=<ID=QD,Number=1,Type=Float,Description=\"Variant Confidence Quality by Depth\">"};
    header.insert(header.end(), addition_info.begin(), addition_info.end());
    for (const auto &c : contigs)
        header.push_back("##contig=<ID=" + c.name + ",length=" + std::to_string(c.length) + ",assembly=" + ref_file_path + ">");
    header.push_back("##reference=file://" + ref_abs_path);
    header.push_back("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + join(samples, "\t"));
    return join(header, "\n");
}

}  // namespace bvamd
