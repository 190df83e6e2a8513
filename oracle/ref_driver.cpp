// TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
//
// ref_driver.cpp: a thin C-ABI driver around the *real* reference hot path.
// It is compiled by oracle/Makefile together with the reference's own sources,
// taken where they lie under /root/reference (never copied into this repo):
//     src/basetype.cpp  src/utils.cpp  htslib/kfunc.c
// and produces oracle/_ref/libbvref.so.  Only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load that library.
//
// The driver contains no algorithm of its own.  It converts the slab planes of
// include/basevar_amd.h into the reference's `BatchInfo` (src/basetype.h:25-43)
// one site at a time and then performs exactly the call sequence of
//   _basevar_caller   src/basetype_caller.cpp:738-762
//   _out_cvg_line     src/basetype_caller.cpp:1236-1245
//   _out_vcf_line     src/basetype_caller.cpp:1113-1164
//   __gb              src/basetype_caller.cpp:767-797
// using only the reference's public API (src/basetype.h:102-181).

#include <cctype>
#include <climits>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "basetype.h"  // reference header (-I/root/reference/src)
#include "utils.h"     // reference header: ngslib::split / join / tostring

#include "../include/basevar_amd.h"

// Non-inline free functions that the reference defines in algorithm.h, which is
// compiled exactly once inside basetype.cpp's translation unit (src/basetype.cpp:14).
double chi2_test(double chi_sqrt_value, double degree_of_freedom);                  // algorithm.h:44
double norm_dist(double x);                                                         // algorithm.h:48
double fisher_exact_test(int, int, int, int, bool, bool, bool);                     // algorithm.h:62
double wilcoxon_ranksum_test(const std::vector<double> &, const std::vector<double> &);  // algorithm.h:76

namespace {

const char BASE2CHAR[8] = {'A', 'C', 'G', 'T', 'N', 'N', 'N', 'N'};   // bv_slab.ref_base codes
const char NOCALL2CHAR[4] = {'N', '+', '-', 'N'};                     // cell bits 0-1 when bit 3 set

inline int base_index(char b) {
    switch (b) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return 4;
    }
}

void fill_batchinfo(BatchInfo &bi, const uint8_t *bs, const uint8_t *q, const uint8_t *mq,
                    const uint16_t *rp, uint32_t n, uint8_t ref_code) {
    bi.n = n;
    bi.ref_id = "chrS";
    bi.ref_pos = 1;
    bi.ref_base = std::string(1, BASE2CHAR[ref_code & 7]);
    bi.depth = 0;
    bi.align_bases.resize(n);
    bi.align_base_quals.resize(n);
    bi.mapqs.resize(n);
    bi.map_strands.resize(n);
    bi.base_pos_ranks.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        bool nocall = (bs[i] & BV_CELL_NOCALL) != 0;
        char c = nocall ? NOCALL2CHAR[bs[i] & 3u] : BASE2CHAR[bs[i] & 3u];
        if (c == '+' || c == '-') {
            bi.align_bases[i] = std::string(1, c) + "A";  // indel token, e.g. "+A"
        } else {
            bi.align_bases[i] = std::string(1, c);
        }
        bi.align_base_quals[i] = (char)(q[i] + 33);
        bi.mapqs[i] = mq ? mq[i] : 0;
        bi.base_pos_ranks[i] = rp ? rp[i] : 0;
        if (c == 'N') {
            bi.map_strands[i] = '.';
        } else {
            bi.map_strands[i] = (bs[i] & BV_CELL_REV) ? '-' : '+';
            bi.depth++;
        }
    }
}

// `path_seconds` (optional) accumulates the time spent inside the reference's own code only:
// the conversion of the slab row into a BatchInfo above is this driver's overhead, not part of
// the path (the reference fills BatchInfo while parsing text, caller.cpp:688-715, also off-path).
void run_site(const uint8_t *bs, const uint8_t *q, const uint8_t *mq, const uint16_t *rp,
              uint8_t ref_code, const uint8_t *group_id, uint32_t n_groups, uint32_t n,
              double min_af, bv_site_result *r, bv_group_result *g, double *path_seconds) {
    std::memset(r, 0, sizeof(*r));
    r->chi2 = NAN;  // not observable through the reference's public API
    r->mq_ranksum = r->rpr_ranksum = r->bq_ranksum = NAN;
    if (g) std::memset(g, 0, sizeof(*g) * n_groups);

    BatchInfo bi;
    fill_batchinfo(bi, bs, q, mq, rp, n, ref_code);
    if (bi.depth == 0) return;  // caller.cpp:718
    struct Timer {
        double *acc;
        std::chrono::steady_clock::time_point t0;
        explicit Timer(double *a) : acc(a), t0(std::chrono::steady_clock::now()) {}
        ~Timer() {
            if (acc) *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
    } timer(path_seconds);

    // first characters, as both emitters build them (caller.cpp:1131-1134, 1229-1233)
    std::vector<char> align_bases(n);
    for (uint32_t i = 0; i < n; ++i) align_bases[i] = bi.align_bases[i][0];
    char upper_ref = toupper(bi.ref_base[0]);

    // ---- _out_cvg_line: caller.cpp:1236-1245.  The CVG row (and with it fs/sor) is only
    // written when the ACGT depth is > 0 (caller.cpp:1246); otherwise the record stays zero.
    {
        std::string alts;
        for (char b : std::string("ACGT"))
            if (b != upper_ref) alts.push_back(b);
        StrandBiasInfo s = strand_bias(upper_ref, alts, align_bases, bi.map_strands);
        if (s.ref_fwd + s.ref_rev + s.alt_fwd + s.alt_rev > 0) {
            r->cvg_sb[0] = s.ref_fwd; r->cvg_sb[1] = s.ref_rev;
            r->cvg_sb[2] = s.alt_fwd; r->cvg_sb[3] = s.alt_rev;
            r->cvg_fs = s.fs; r->cvg_sor = s.sor;
            // ABI bookkeeping (not reference behaviour): mark SOR values whose `int` products overflowed
            if ((int64_t)s.ref_fwd * s.alt_rev > INT32_MAX || (int64_t)s.ref_rev * s.alt_fwd > INT32_MAX)
                r->status |= BV_SITE_SOR_OVERFLOW;
        }
    }

    // ---- caller.cpp:742-743
    BaseType bt(&bi, min_af);
    bt.lrt();
    r->total_depth = bt.get_total_depth();
    const char ACGT[4] = {'A', 'C', 'G', 'T'};
    for (int j = 0; j < 4; ++j) r->depth[j] = (uint32_t)bt.get_base_depth(ACGT[j]);
    if (r->total_depth > 0) r->status |= BV_SITE_COVERED;

    const std::vector<char> &alt = bt.get_alt_bases();
    if (alt.empty()) return;  // caller.cpp:745
    r->status |= BV_SITE_VARIANT | BV_SITE_RANKSUM;
    r->n_alt = (uint8_t)alt.size();
    r->qual = bt.get_var_qual();

    // ---- _out_vcf_line: caller.cpp:1113-1164
    double ad_sum = 0;
    std::string alt_str;
    for (size_t i = 0; i < alt.size(); ++i) {
        char b = alt[i];
        alt_str.push_back(b);
        r->alt[i] = (uint8_t)base_index(b);
        ad_sum = ad_sum + bt.get_base_depth(b);
        r->af[i] = bt.get_lrt_af(b);
        r->caf[i] = bt.get_base_depth(b) / bt.get_total_depth();
    }
    r->mq_ranksum = ref_vs_alt_ranksumtest(upper_ref, alt_str, align_bases, bi.mapqs);
    r->rpr_ranksum = ref_vs_alt_ranksumtest(upper_ref, alt_str, align_bases, bi.base_pos_ranks);
    r->bq_ranksum = ref_vs_alt_ranksumtest(upper_ref, alt_str, align_bases, bi.align_base_quals);
    double qd = bt.get_var_qual() / ad_sum;
    if (qd == 0) qd = 0.0;
    r->qd = qd;
    StrandBiasInfo s = strand_bias(upper_ref, alt_str, align_bases, bi.map_strands);
    r->var_sb[0] = s.ref_fwd; r->var_sb[1] = s.ref_rev;
    r->var_sb[2] = s.alt_fwd; r->var_sb[3] = s.alt_rev;
    r->var_fs = s.fs; r->var_sor = s.sor;
    if ((int64_t)s.ref_fwd * s.alt_rev > INT32_MAX || (int64_t)s.ref_rev * s.alt_fwd > INT32_MAX)
        r->status |= BV_SITE_SOR_OVERFLOW;

    // ---- per-group calls: caller.cpp:746-759, __gb :767-797
    if (g && n_groups > 0 && group_id) {
        std::vector<char> basecombination;
        basecombination.push_back(upper_ref);
        basecombination.insert(basecombination.end(), alt.begin(), alt.end());
        for (uint32_t gi = 0; gi < n_groups; ++gi) {
            BatchInfo gb;
            gb.ref_id = bi.ref_id; gb.ref_pos = bi.ref_pos; gb.ref_base = bi.ref_base;
            gb.depth = bi.depth;
            for (uint32_t i = 0; i < n; ++i) {
                if (group_id[i] != gi) continue;
                gb.align_bases.push_back(bi.align_bases[i]);
                gb.align_base_quals.push_back(bi.align_base_quals[i]);
                gb.mapqs.push_back(bi.mapqs[i]);
                gb.map_strands.push_back(bi.map_strands[i]);
                gb.base_pos_ranks.push_back(bi.base_pos_ranks[i]);
            }
            gb.n = gb.align_bases.size();
            BaseType gbt(&gb, min_af);
            gbt.lrt(basecombination);
            const std::vector<char> &ga = gbt.get_alt_bases();
            g[gi].n_alt = (uint8_t)ga.size();
            g[gi].total_depth = gbt.get_total_depth();
            for (size_t k = 0; k < ga.size(); ++k) {
                g[gi].alt[k] = (uint8_t)base_index(ga[k]);
                g[gi].af[k] = gbt.get_lrt_af(ga[k]);
            }
        }
    }
}

}  // namespace

extern "C" {

// Runs the reference path over a host slab.  Returns 0, or -1 if the reference threw.
static int run_impl(const uint8_t *base_strand, const uint8_t *qual, const uint8_t *mapq,
              const uint16_t *rpr, const uint8_t *ref_base, const uint8_t *group_id,
              uint32_t n_groups, uint32_t n_sites, uint32_t n_samples, uint64_t pitch,
              double min_af, bv_site_result *out, bv_group_result *gout, int n_threads,
              char *errbuf, size_t errlen, double *path_seconds_per_thread) {
    if (n_threads < 1) n_threads = 1;
    std::vector<std::string> errs(n_threads);
    auto work = [&](int t) {
        // static contiguous partition of the site range per thread, mirroring the
        // reference's per-sub-region tasks (caller.cpp:489-510)
        uint64_t lo = (uint64_t)n_sites * t / n_threads, hi = (uint64_t)n_sites * (t + 1) / n_threads;
        try {
            for (uint64_t s = lo; s < hi; ++s) {
                run_site(base_strand + s * pitch, qual + s * pitch, mapq ? mapq + s * pitch : nullptr,
                         rpr ? rpr + s * pitch : nullptr, ref_base[s], group_id, n_groups, n_samples,
                         min_af, out + s, gout ? gout + s * n_groups : nullptr,
                         path_seconds_per_thread ? path_seconds_per_thread + t : nullptr);
            }
        } catch (const std::exception &ex) {
            errs[t] = ex.what();
        }
    };
    if (n_threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    for (auto &e : errs)
        if (!e.empty()) {
            if (errbuf && errlen) {
                std::strncpy(errbuf, e.c_str(), errlen - 1);
                errbuf[errlen - 1] = 0;
            }
            return -1;
        }
    return 0;
}

int bvref_run(const uint8_t *base_strand, const uint8_t *qual, const uint8_t *mapq,
              const uint16_t *rpr, const uint8_t *ref_base, const uint8_t *group_id,
              uint32_t n_groups, uint32_t n_sites, uint32_t n_samples, uint64_t pitch,
              double min_af, bv_site_result *out, bv_group_result *gout, int n_threads,
              char *errbuf, size_t errlen) {
    return run_impl(base_strand, qual, mapq, rpr, ref_base, group_id, n_groups, n_sites, n_samples, pitch, min_af,
                    out, gout, n_threads, errbuf, errlen, nullptr);
}

// Same, and reports per thread the seconds spent inside the reference's path proper (BaseType
// ctor + lrt + strand_bias + rank sums + group calls), excluding this driver's slab -> BatchInfo
// conversion.  `path_seconds` must hold n_threads zero-initialised doubles.
int bvref_run_timed(const uint8_t *base_strand, const uint8_t *qual, const uint8_t *mapq,
                    const uint16_t *rpr, const uint8_t *ref_base, const uint8_t *group_id,
                    uint32_t n_groups, uint32_t n_sites, uint32_t n_samples, uint64_t pitch,
                    double min_af, bv_site_result *out, bv_group_result *gout, int n_threads,
                    char *errbuf, size_t errlen, double *path_seconds) {
    return run_impl(base_strand, qual, mapq, rpr, ref_base, group_id, n_groups, n_sites, n_samples, pitch, min_af,
                    out, gout, n_threads, errbuf, errlen, path_seconds);
}

// Scalar entry points for the known-answer inputs of tests/io/test_algorithm.cpp:13-31.
double bvref_chi2_test(double x, double df) { return chi2_test(x, df); }
double bvref_norm_dist(double x) { return norm_dist(x); }
double bvref_fisher_exact_test(int a, int b, int c, int d) { return fisher_exact_test(a, b, c, d, false, false, true); }
double bvref_wilcoxon(const double *s1, int n1, const double *s2, int n2) {
    return wilcoxon_ranksum_test(std::vector<double>(s1, s1 + n1), std::vector<double>(s2, s2 + n2));
}
double bvref_ranksum_int(char ref, const char *alts, const char *bases, const int *values, int n) {
    return ref_vs_alt_ranksumtest(ref, std::string(alts), std::vector<char>(bases, bases + n),
                                  std::vector<int>(values, values + n));
}
int bvref_strand_bias(char ref, const char *alts, const char *bases, const char *strands, int n,
                      int *counts, double *fs, double *sor) {
    StrandBiasInfo s = strand_bias(ref, std::string(alts), std::vector<char>(bases, bases + n),
                                   std::vector<char>(strands, strands + n));
    counts[0] = s.ref_fwd; counts[1] = s.ref_rev; counts[2] = s.alt_fwd; counts[3] = s.alt_rev;
    *fs = s.fs; *sor = s.sor;
    return 0;
}

// ---- the reference's own text primitives (src/utils.h:38-43, 75-122; src/utils.cpp:81-99), used
// to pin the tokenisers / formatters of basevar_amd/host/batchfile.hpp.  Results are written
// '\x1f'-separated into `out` (truncated at outlen - 1); the return value is the item count.
static int pack_items(const std::vector<std::string> &v, char *out, size_t outlen) {
    std::string s;
    for (size_t i = 0; i < v.size(); ++i) {
        if (i) s.push_back('\x1f');
        s += v[i];
    }
    if (out && outlen) {
        std::strncpy(out, s.c_str(), outlen - 1);
        out[outlen - 1] = 0;
    }
    return (int)v.size();
}
int bvref_split_str(const char *in, const char *delim, char *out, size_t outlen) {
    std::vector<std::string> v;
    ngslib::split(std::string(in), v, delim);
    return pack_items(v, out, outlen);
}
int bvref_split_int(const char *in, const char *delim, char *out, size_t outlen) {
    std::vector<int> v;
    ngslib::split(std::string(in), v, delim);
    std::vector<std::string> sv;
    for (int x : v) sv.push_back(std::to_string(x));
    return pack_items(sv, out, outlen);
}
int bvref_split_char(const char *in, const char *delim, char *out, size_t outlen) {
    std::vector<char> v;
    ngslib::split(std::string(in), v, delim);
    std::vector<std::string> sv;
    for (char x : v) sv.push_back(std::to_string((int)x));
    return pack_items(sv, out, outlen);
}
int bvref_join_double(const double *v, int n, const char *delim, char *out, size_t outlen) {
    std::string s = ngslib::join(std::vector<double>(v, v + n), delim);
    std::strncpy(out, s.c_str(), outlen - 1);
    out[outlen - 1] = 0;
    return (int)s.size();
}
int bvref_join_int(const int *v, int n, const char *delim, char *out, size_t outlen) {
    std::string s = ngslib::join(std::vector<int>(v, v + n), delim);
    std::strncpy(out, s.c_str(), outlen - 1);
    out[outlen - 1] = 0;
    return (int)s.size();
}
int bvref_join_char(const char *v, int n, const char *delim, char *out, size_t outlen) {
    std::string s = ngslib::join(std::vector<char>(v, v + n), delim);
    std::strncpy(out, s.c_str(), outlen - 1);
    out[outlen - 1] = 0;
    return (int)s.size();
}

}  // extern "C"
