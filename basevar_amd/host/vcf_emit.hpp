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
// FS, SOR, BP; ostringstream default (6 significant digits) through join() for CM_AF, CM_CAF and
// the group AFs; the three rank sums truncated to int (caller.cpp:1151-1157).
//
// PARITY STATUS: the numeric inputs are the pinned bv_site_result fields; join()/tostring() are
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

// __base_depth_and_indel, caller.cpp:1263-1289: "TOKEN|count" of the non-ACGT, non-N tokens,
// ordered by std::map (lexicographic), "." if none.
inline std::string indel_string(const std::vector<std::string> &tokens) {
    std::map<std::string, int> indel_depth;
    for (const auto &bs : tokens) {
        if (bs.empty() || bs[0] == 'N') continue;
        if (bs[0] == 'A' || bs[0] == 'C' || bs[0] == 'G' || bs[0] == 'T') continue;
        indel_depth[bs]++;
    }
    std::vector<std::string> indels;
    for (const auto &kv : indel_depth) indels.push_back(kv.first + "|" + std::to_string(kv.second));
    return indels.empty() ? "." : join(indels, ",");
}

// _out_cvg_line, caller.cpp:1246-1257.  Empty string when the reference writes nothing.
inline std::string format_cvg_line(const SiteText &st, const bv_site_result &r) {
    if (r.total_depth == 0) return "";
    std::vector<int> dd = {(int)r.depth[0], (int)r.depth[1], (int)r.depth[2], (int)r.depth[3]};
    return st.ref_id + "\t" + std::to_string(st.ref_pos) + "\t" + st.ref_base + "\t" + std::to_string((int)r.total_depth) +
           "\t" + join(dd, "\t") + "\t" + indel_string(st.indel_tokens) + "\t" + std::to_string(r.cvg_fs) + "\t" +
           std::to_string(r.cvg_sor) + "\t" + std::to_string((int)r.cvg_sb[0]) + "," + std::to_string((int)r.cvg_sb[1]) +
           "," + std::to_string((int)r.cvg_sb[2]) + "," + std::to_string((int)r.cvg_sb[3]) + "\n";
}

// _out_vcf_line, caller.cpp:1103-1209, from the site's slab row: `cell` / `phred` are the n per-sample bytes of the
// base_strand and qual planes (include/basevar_amd.h).  `groups`/`group_names`: the site's bv_group_result records
// and the group names in the reference's iteration order (std::map: sorted by name); may be empty.
inline std::string format_vcf_line(const SiteText &st, const uint8_t *cell, const uint8_t *phred, size_t n, const bv_site_result &r,
                                   const bv_group_result *groups, const std::vector<std::string> &group_names) {
    if (r.n_alt == 0) return "";  // caller.cpp:745
    std::map<char, std::string> alt_gt;
    std::vector<int> cm_ac;
    std::vector<double> cm_af, cm_caf;
    std::vector<char> alt_bases;
    for (int i = 0; i < r.n_alt; ++i) {
        const char b = EMIT_BASES[r.alt[i] & 3];
        alt_bases.push_back(b);
        alt_gt[b] = "./" + std::to_string(i + 1);
        cm_ac.push_back((int)r.depth[r.alt[i] & 3]);  // (int)get_base_depth, :1120
        cm_af.push_back(r.af[i]);
        cm_caf.push_back(r.caf[i]);
    }
    // per-sample GT:AB:SO:BP, caller.cpp:1125-1145 -- the same text the reference builds with a vector of strings and
    // ngslib::join (one ostringstream per sample), appended directly: a VCF line is n_samples fields, and at 10^4 samples the
    // emitter, not the engine, sets the pace of a run (profiles/r3_host_pipeline.txt).  std::to_string(1 - eps(q)) depends on
    // the phred byte only: 256 strings, formed once.
    static const std::vector<std::string> bp_text = [] {
        std::vector<std::string> t(256);
        for (int qv = 0; qv < 256; ++qv) t[(size_t)qv] = std::to_string(1.0 - std::exp(qv * EMIT_MLN10TO10));  // basetype.cpp:47-48
        return t;
    }();
    std::string gt_of[4];
    for (int b = 0; b < 4; ++b) {
        const auto it = alt_gt.find(EMIT_BASES[b]);
        gt_of[b] = it == alt_gt.end() ? "./." : it->second;
    }
    const char upper_ref = (char)std::toupper((unsigned char)st.ref_base[0]);
    std::string samples;
    samples.reserve(n * 5);
    for (size_t i = 0; i < n; ++i) {
        if (i) samples += '\t';
        if (!(cell[i] & BV_CELL_NOCALL)) {
            const int bc = cell[i] & 3;
            const char fb = EMIT_BASES[bc];
            if (fb == upper_ref) samples += "0/."; else samples += gt_of[bc];
            samples += ':'; samples += fb; samples += ':';
            samples += (cell[i] & BV_CELL_REV) ? '-' : '+';
            samples += ':';
            samples += bp_text[phred[i]];
        } else {
            samples += "./.";
        }
    }
    const int mq_rank_sum = (int)r.mq_ranksum, read_pos_rank_sum = (int)r.rpr_ranksum, base_q_rank_sum = (int)r.bq_ranksum;
    std::vector<std::string> info = {
        "CM_DP=" + std::to_string((int)r.total_depth),
        "CM_AC=" + join(cm_ac, ","),
        "CM_AF=" + join(cm_af, ","),
        "CM_CAF=" + join(cm_caf, ","),
        "MQRankSum=" + std::to_string(mq_rank_sum),
        "ReadPosRankSum=" + std::to_string(read_pos_rank_sum),
        "BaseQRankSum=" + std::to_string(base_q_rank_sum),
        "QD=" + std::to_string(r.qd),
        "SOR=" + std::to_string(r.var_sor),
        "FS=" + std::to_string(r.var_fs),
        "SB_REF=" + std::to_string((int)r.var_sb[0]) + "," + std::to_string((int)r.var_sb[1]),
        "SB_ALT=" + std::to_string((int)r.var_sb[2]) + "," + std::to_string((int)r.var_sb[3]),
    };
    if (groups && !group_names.empty()) {  // caller.cpp:1182-1196
        for (size_t g = 0; g < group_names.size(); ++g) {
            std::vector<double> af;
            for (int k = 0; k < groups[g].n_alt; ++k) af.push_back(groups[g].af[k]);
            if (!af.empty()) info.push_back(group_names[g] + "_AF=" + join(af, ","));
        }
    }
    const std::string qs = (r.qual > EMIT_QUAL_THRESHOLD) ? "." : "LowQual";
    return st.ref_id + "\t" + std::to_string(st.ref_pos) + "\t.\t" + st.ref_base + "\t" + join(alt_bases, ",") + "\t" +
           std::to_string(r.qual) + "\t" + qs + "\t" + join(info, ";") + "\tGT:AB:SO:BP\t" + samples + "\n";
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
    std::vector<char> bases(EMIT_BASES, EMIT_BASES + 4);
    const std::string h = "#CHROM\tPOS\tREF\tDepth\t" + join(bases, "\t") + "\t" +
                          "Indels\tFS\tSOR\tStrand_Coverage(REF_FWD,REF_REV,ALT_FWD,ALT_REV)";
    std::vector<std::string> header = {"##fileformat=CVGv1.0", "##Group information is the depth of A:C:G:T:Indel", h};
    return join(header, "\n");
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
