/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * refcpu.c: plain-C, CPU-only restatement of the reference's per-site basetype path.
 * It exists to check the HIP engine; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it (as oracle/liboracle.so).  The product
 * library never links, loads or calls anything in this directory.
 *
 * Parity status: PINNED.  This restatement is validated (tests/test_oracle_cpu.py)
 *   (1) against the real reference compiled from /root/reference (oracle/_ref, built by
 *       oracle/Makefile) on seeded slabs in the build container, and
 *   (2) against golden vectors generated from that same compiled reference and committed
 *       under tests/golden/ (generator: tests/golden/make_golden.py), and
 *   (3) against the known answers for the inputs of the reference's own
 *       tests/io/test_algorithm.cpp:13-31.
 * The reference has no golden outputs of its own for EM/LRT (test_algorithm.cpp:41).
 *
 * The code is structure-faithful on purpose: per-sample n x 4 likelihood rows, the same
 * loop order and the same floating-point evaluation order as the reference, including
 * its observable quirks (integer abs() in the EM convergence test, float min_af, int
 * products in SOR).  Each function cites the reference lines it follows; all paths are
 * relative to /root/reference.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/basevar_amd.h"

/* src/basetype.h:20-22 */
static const double MLN10TO10 = -0.23025850929940458;
static const int LRT_THRESHOLD = 24;

/* =====================================================================================
 * Special functions -- htslib/kfunc.c (vendored htslib, HTS_VERSION 102190)
 * ===================================================================================== */

/* kf_lgamma, htslib/kfunc.c:39-52 (AS245 algorithm 2).  Terms are accumulated from the
 * (z+7) term down to the 1/z term, then the constant, exactly in that order. */
static double o_kf_lgamma(double z) {
    static const double coef[8] = {/* divided by (z+k), k = index */
                                   676.5203681218835,      -1259.139216722289,   771.3234287757674,
                                   -176.6150291498386,     12.50734324009056,    -0.1385710331296526,
                                   0.9934937113930748e-05, 0.1659470187408462e-06};
    double x = 0;
    for (int k = 7; k >= 1; --k) x += coef[k] / (z + k);
    x += coef[0] / z;
    x += 0.9999999999995183;
    return log(x) - 5.58106146679532777 - z + (z - 0.5) * log(z + 6.5);
}

/* kf_erfc, htslib/kfunc.c:58-84 (AS66 algorithm 2). */
static double o_kf_erfc(double x) {
    static const double P[7] = {220.2068679123761, 221.2135961699311, 112.0792914978709, 33.912866078383,
                                6.37396220353165,  .7003830644436881, .03526249659989109};
    static const double Q[8] = {440.4137358247522, 793.8265125199484, 637.3336333788311, 296.5642487796737,
                                86.78073220294608, 16.06417757920695, 1.755667163182642, .08838834764831844};
    double z = fabs(x) * M_SQRT2;
    if (z > 37.) return x > 0. ? 0. : 2.;
    double expntl = exp(z * z * -.5);
    double p;
    if (z < 10. / M_SQRT2) {
        double num = P[6], den = Q[7];
        for (int k = 5; k >= 0; --k) num = num * z + P[k];
        for (int k = 6; k >= 0; --k) den = den * z + Q[k];
        p = expntl * num / den;
    } else {
        p = expntl / 2.506628274631001 / (z + 1. / (z + 2. / (z + 3. / (z + 4. / (z + .65)))));
    }
    return x > 0. ? 2. * p : 2. * (1. - p);
}

#define O_KF_GAMMA_EPS 1e-14
#define O_KF_TINY 1e-290

/* _kf_gammap, htslib/kfunc.c:103-112: series for the regularized lower incomplete gamma */
static double o_gammap_series(double s, double z) {
    double sum = 1., x = 1.;
    for (int k = 1; k < 100; ++k) {
        x *= z / (s + k);
        sum += x;
        if (x / sum < O_KF_GAMMA_EPS) break;
    }
    return exp(s * log(z) - z - o_kf_lgamma(s + 1.) + log(sum));
}

/* _kf_gammaq, htslib/kfunc.c:114-133: modified Lentz continued fraction */
static double o_gammaq_cf(double s, double z) {
    double f = 1. + z - s, C = f, D = 0.;
    for (int j = 1; j < 100; ++j) {
        double a = j * (s - j), b = (j << 1) + 1 + z - s, d;
        D = b + a * D;
        if (D < O_KF_TINY) D = O_KF_TINY;
        C = b + a / C;
        if (C < O_KF_TINY) C = O_KF_TINY;
        D = 1. / D;
        d = C * D;
        f *= d;
        if (fabs(d - 1.) < O_KF_GAMMA_EPS) break;
    }
    return exp(s * log(z) - z - o_kf_lgamma(s) - log(f));
}

/* kf_gammaq, htslib/kfunc.c:140-143 */
static double o_kf_gammaq(double s, double z) {
    return (z <= 1. || z < s) ? 1. - o_gammap_series(s, z) : o_gammaq_cf(s, z);
}

/* chi2_test, src/algorithm.h:44-46 */
double oracle_chi2_test(double chi, double df) { return o_kf_gammaq(df / 2.0, chi / 2.0); }

/* norm_dist, src/algorithm.h:48-50 */
double oracle_norm_dist(double x) { return o_kf_erfc((double)(x / sqrt(2.0))) / 2.0; }

/* lbinom + hypergeo, htslib/kfunc.c:197-212 (libm lgamma, not kf_lgamma) */
static double o_lbinom(int n, int k) {
    if (k == 0 || n == k) return 0;
    return lgamma(n + 1) - lgamma(k + 1) - lgamma(n - k + 1);
}
static double o_hypergeo(int n11, int n1_, int n_1, int n) {
    return exp(o_lbinom(n1_, n11) + o_lbinom(n - n1_, n_1 - n11) - o_lbinom(n, n_1));
}

/* hypergeo_acc, htslib/kfunc.c:214-243: table probability, updated multiplicatively from
 * the previous n11 except when n11 % 11 == 0 or n22 == 0, where it is recomputed. */
typedef struct {
    int n11, n1_, n_1, n;
    double p;
} o_hg_t;

static double o_hg_start(o_hg_t *h, int n11, int n1_, int n_1, int n) {
    /* the reference re-seeds when any of n1_, n_1, n is non-zero */
    h->n11 = n11; h->n1_ = n1_; h->n_1 = n_1; h->n = n;
    h->p = o_hypergeo(n11, n1_, n_1, n);
    return h->p;
}
static double o_hg_move(o_hg_t *h, int n11) {
    int n22 = n11 + h->n - h->n1_ - h->n_1;
    if ((n11 % 11) && n22) {
        if (n11 == h->n11 + 1) {
            h->p *= (double)(h->n1_ - h->n11) / n11 * (h->n_1 - h->n11) / n22;
            h->n11 = n11;
            return h->p;
        }
        if (n11 == h->n11 - 1) {
            h->p *= (double)h->n11 / (h->n1_ - n11) * (h->n11 + h->n - h->n1_ - h->n_1) / (h->n_1 - n11);
            h->n11 = n11;
            return h->p;
        }
    }
    h->n11 = n11;
    h->p = o_hypergeo(h->n11, h->n1_, h->n_1, h->n);
    return h->p;
}

/* kt_fisher_exact (two-sided value only), htslib/kfunc.c:245-313, through
 * fisher_exact_test, src/algorithm.h:62-74 */
double oracle_fisher_exact_test(int n11, int n12, int n21, int n22) {
    int n1_ = n11 + n12, n_1 = n11 + n21, n = n11 + n12 + n21 + n22;
    int max = (n_1 < n1_) ? n_1 : n1_;
    int min = n1_ + n_1 - n;
    if (min < 0) min = 0;
    if (min == max) return 1.;
    o_hg_t h;
    double q = o_hg_start(&h, n11, n1_, n_1, n);
    if (q == 0.0) return 0.0; /* kfunc.c:260-289: *two = 0 on either branch */

    int i, j;
    double p, left, right;
    p = o_hg_move(&h, min);
    for (left = 0., i = min + 1; p < 0.99999999 * q && i <= max; ++i) {
        left += p;
        p = o_hg_move(&h, i);
    }
    --i;
    if (p < 1.00000001 * q) left += p;
    else --i;
    p = o_hg_move(&h, max);
    for (right = 0., j = max - 1; p < 0.99999999 * q && j >= 0; --j) {
        right += p;
        p = o_hg_move(&h, j);
    }
    ++j;
    if (p < 1.00000001 * q) right += p;
    else ++j;
    double two = left + right;
    if (two > 1.) two = 1.;
    return two;
}

/* wilcoxon_ranksum_test, src/algorithm.h:76-136.  Descending sort, average ranks for
 * ties, no tie or continuity correction. */
typedef struct {
    double v;
    size_t idx;
} o_rk_t;
static int o_rk_desc(const void *a, const void *b) {
    double x = ((const o_rk_t *)a)->v, y = ((const o_rk_t *)b)->v;
    return (x > y) ? -1 : (x < y) ? 1 : 0;
}
double oracle_wilcoxon(const double *s1, size_t n1, const double *s2, size_t n2) {
    size_t n = n1 + n2;
    o_rk_t *c = (o_rk_t *)malloc(sizeof(o_rk_t) * (n ? n : 1));
    double *rankv = (double *)malloc(sizeof(double) * (n ? n : 1));
    for (size_t i = 0; i < n1; ++i) { c[i].v = s1[i]; c[i].idx = i; }
    for (size_t i = 0; i < n2; ++i) { c[n1 + i].v = s2[i]; c[n1 + i].idx = n1 + i; }
    qsort(c, n, sizeof(o_rk_t), o_rk_desc);
    for (size_t i = 0; i < n; ++i) rankv[i] = (double)(i + 1);

    double ranksum = 0.0, same_n = 1;
    size_t i;
    for (i = 0; i < n; ++i) {
        if (i > 0 && c[i].v != c[i - 1].v) {
            if (same_n > 1) {
                double avg = ranksum / same_n;
                for (size_t j = i - (size_t)same_n; j < i; ++j) rankv[j] = avg;
            }
            same_n = 1;
            ranksum = 0;
        } else if (i > 0) {
            same_n++;
        }
        ranksum += (double)(i + 1);
    }
    if (same_n > 1) {
        double avg = ranksum / same_n;
        for (size_t j = i - (size_t)same_n; j < i; ++j) rankv[j] = avg;
    }
    double r1 = 0.0;
    for (size_t k = 0; k < n; ++k)
        if (c[k].idx < n1) r1 += rankv[k];
    free(c);
    free(rankv);

    double e = (double)(n1 * (n1 + n2 + 1)) / 2.0;
    double z = (r1 - e) / sqrt((double)(n1 * n2 * (n1 + n2 + 1)) / 12.0);
    return 2 * oracle_norm_dist(fabs(z));
}

/* =====================================================================================
 * EM -- src/algorithm.h:148-255
 * ===================================================================================== */
typedef struct {
    size_t n;          /* covered samples (rows) */
    double (*lh)[4];   /* _ind_allele_likelihood, n x 4 */
    double (*post)[4]; /* ind_allele_post_prob     */
    double *marg;      /* marginal_likelihood      */
    double *llh;       /* log_marginal_likelihood  */
} o_em_ws;

/* e_step, algorithm.h:148-175 */
static void o_e_step(const double f[4], o_em_ws *w) {
    for (size_t i = 0; i < w->n; ++i) {
        double L[4];
        w->marg[i] = 0;
        for (int j = 0; j < 4; ++j) {
            L[j] = w->lh[i][j] * f[j];
            w->marg[i] += L[j];
        }
        for (int j = 0; j < 4; ++j) w->post[i][j] = L[j] / w->marg[i];
    }
}
/* m_step, algorithm.h:184-198 */
static void o_m_step(double f[4], const o_em_ws *w) {
    for (int j = 0; j < 4; ++j) {
        f[j] = 0;
        for (size_t i = 0; i < w->n; ++i) f[j] += w->post[i][j];
        f[j] /= (double)(w->n);
    }
}
/* EM, algorithm.h:210-255.  Returns the number of loop iterations executed. */
static int o_em(double f[4], o_em_ws *w) {
    const float epsilon = 0.001f; /* `const float epsilon=0.001`, algorithm.h:213 */
    int iter_num = 100, iters = 0;
    o_e_step(f, w);
    for (size_t i = 0; i < w->n; ++i) w->llh[i] = log(w->marg[i]);
    o_m_step(f, w);
    while (iter_num--) {
        o_e_step(f, w);
        o_m_step(f, w);
        double delta = 0, l;
        for (size_t i = 0; i < w->n; ++i) {
            l = log(w->marg[i]);
            /* algorithm.h:245 calls unqualified abs() on a double: with libstdc++ this is
             * int abs(int), i.e. the difference is truncated to int first (SURVEY trap #1;
             * confirmed against oracle/_ref by tests/test_oracle_cpu.py). */
            delta += abs((int)(l - w->llh[i]));
            w->llh[i] = l;
        }
        ++iters;
        if (delta < epsilon) break;
    }
    o_m_step(f, w);
    return iters;
}

/* =====================================================================================
 * BaseType -- src/basetype.cpp:22-199
 * ===================================================================================== */
typedef struct {
    double depth[5]; /* _depth for A,C,G,T (+ a zero slot for non-ACGT lookups) */
    int total_depth;
    double min_af;
    o_em_ws w;
    /* lrt outputs */
    int n_alt;
    int alt[4];
    double af[4];
    double var_qual;
    double chi2;
    int em_iters, n_em, zero_freq;
    double margin; /* diagnostic: smallest gap that decided a discrete choice in lrt() (see o_lrt) */
} o_bt;

/* BaseType::BaseType, basetype.cpp:22-72.  `idx` optionally selects a subset of samples
 * (__get_group_batchinfo, caller.cpp:779-797). */
static void o_bt_init(o_bt *bt, const uint8_t *bs, const uint8_t *q, uint32_t n, const uint32_t *idx,
                      uint32_t n_idx, double min_af) {
    memset(bt, 0, sizeof(*bt));
    bt->min_af = min_af;
    uint32_t cnt = idx ? n_idx : n;
    bt->w.lh = (double(*)[4])malloc(sizeof(double[4]) * (cnt ? cnt : 1));
    size_t rows = 0;
    for (uint32_t k = 0; k < cnt; ++k) {
        uint32_t i = idx ? idx[k] : k;
        double epsilon = exp((double)q[i] * MLN10TO10); /* (qual_char - 33) * MLN10TO10, :47 */
        unsigned fb = bs[i] & 3u;
        if (!(bs[i] & BV_CELL_NOCALL)) { /* not 'N', '+', '-' : basetype.cpp:51 */
            bt->depth[fb]++;
            bt->total_depth++;
            for (unsigned j = 0; j < 4; ++j) bt->w.lh[rows][j] = (fb == j) ? 1.0 - epsilon : epsilon / 3;
            ++rows;
        }
    }
    bt->w.n = rows;
    bt->w.post = (double(*)[4])malloc(sizeof(double[4]) * (rows ? rows : 1));
    bt->w.marg = (double *)malloc(sizeof(double) * (rows ? rows : 1));
    bt->w.llh = (double *)malloc(sizeof(double) * (rows ? rows : 1));
}
static void o_bt_free(o_bt *bt) {
    free(bt->w.lh); free(bt->w.post); free(bt->w.marg); free(bt->w.llh);
}

typedef struct {
    int n;             /* number of combinations */
    int bc[6][4];      /* bases of each combination */
    int bc_len;
    double bp[6][4];   /* final freqs */
    double lr[6];      /* sum log marginal likelihood */
} o_aa;

/* BaseType::_f, basetype.cpp:105-128, with Combinations<char> order
 * (src/external/combinations.h:55-69: lexicographic on positions). */
static void o_f(o_bt *bt, const int *bases, int m, int n, o_aa *aa) {
    int pos[4];
    aa->n = 0;
    aa->bc_len = n;
    for (int k = 0; k < n; ++k) pos[k] = k;
    for (;;) {
        int c = aa->n;
        double f[4] = {0, 0, 0, 0};
        /* _set_allele_initial_freq, basetype.cpp:93-103 (not renormalised) */
        double s = 0;
        for (int k = 0; k < n; ++k) {
            int b = bases[pos[k]];
            aa->bc[c][k] = b;
            if (bt->total_depth > 0) f[b] = bt->depth[b] / (double)(bt->total_depth);
        }
        for (int j = 0; j < 4; ++j) s += f[j];
        if (s == 0) bt->zero_freq = 1; /* reference throws, basetype.cpp:113-115 */
        bt->em_iters += o_em(f, &bt->w);
        bt->n_em++;
        double lr = 0;
        for (size_t i = 0; i < bt->w.n; ++i) lr += bt->w.llh[i]; /* sum(), algorithm.h:35-41 */
        for (int j = 0; j < 4; ++j) aa->bp[c][j] = f[j];
        aa->lr[c] = lr;
        aa->n++;
        /* next combination in lexicographic order */
        int k = n - 1;
        while (k >= 0 && pos[k] == m - n + k) --k;
        if (k < 0) break;
        ++pos[k];
        for (int t = k + 1; t < n; ++t) pos[t] = pos[t - 1] + 1;
    }
}

/* BaseType::lrt, basetype.cpp:130-199.  `specific` holds base codes (>= 4: never active). */
static void o_lrt(o_bt *bt, const int *specific, int n_specific, int ref_code) {
    bt->n_alt = 0;
    bt->chi2 = 0;
    bt->var_qual = 0;
    bt->margin = INFINITY;
    if (bt->total_depth == 0) return;
    int active[4], m = 0;
    for (int k = 0; k < n_specific; ++k) {
        int b = specific[k];
        double d = (b < 4) ? bt->depth[b] : 0.0;
        if (d / bt->total_depth >= bt->min_af && b < 4) active[m++] = b;
    }
    if (m == 0) return;

    o_aa var;
    o_f(bt, active, m, m, &var);
    double chi = 0;
    double freq[4];
    memcpy(freq, var.bp[0], sizeof(freq));
    double lr_alt = var.lr[0];

    for (int n = m - 1; n > 0; --n) { /* `n` starts from the ORIGINAL size, :151 */
        o_f(bt, active, m, n, &var);
        double chiv[6];
        int i_min = 0;
        for (int j = 0; j < var.n; ++j) {
            chiv[j] = 2 * (lr_alt - var.lr[j]);
            if (chiv[j] < chiv[i_min]) i_min = j; /* std::min_element: first minimum */
        }
        /* DIAGNOSTIC ONLY (not reference behaviour): how close this level's two discrete decisions
         * were -- the runner-up subset in the argmin, and the LRT threshold.  The reference's
         * tie-breaking depends on the order in which it adds the per-sample log-likelihoods; an
         * engine that works on (base, phred) histograms cannot see that order, so the parity tests
         * report (not fail) the sites whose margin is at rounding-noise level. */
        for (int j = 0; j < var.n; ++j)
            if (j != i_min && chiv[j] - chiv[i_min] < bt->margin) bt->margin = chiv[j] - chiv[i_min];
        if (fabs(chiv[i_min] - LRT_THRESHOLD) < bt->margin) bt->margin = fabs(chiv[i_min] - LRT_THRESHOLD);
        lr_alt = var.lr[i_min];
        chi = chiv[i_min];
        if (chi < LRT_THRESHOLD) {
            m = n;
            for (int k = 0; k < n; ++k) active[k] = var.bc[i_min][k];
            memcpy(freq, var.bp[i_min], sizeof(freq));
        } else {
            break;
        }
    }
    bt->chi2 = chi;
    for (int k = 0; k < m; ++k) {
        if (active[k] != ref_code) {
            bt->alt[bt->n_alt] = active[k];
            bt->af[bt->n_alt] = freq[active[k]];
            bt->n_alt++;
        }
    }
    if (bt->n_alt > 0) {
        double r = bt->depth[active[0]] / (double)(bt->total_depth);
        if (m == 1 && bt->total_depth > 10 && r > 0.5) {
            bt->var_qual = 5000.0;
        } else {
            double p = oracle_chi2_test(chi, 1);
            if (isnan(p)) p = 1.0;
            bt->var_qual = (p) ? -10 * log10(p) : 10000.0;
            if (bt->var_qual == -0.0) bt->var_qual = 0.0;
        }
    }
}

/* strand_bias, basetype.cpp:244-295; alt membership given as a 4-bit mask over ACGT */
static void o_strand_bias(int ref_code, unsigned alt_mask, const uint8_t *bs, uint32_t n, uint32_t cnt[4],
                          double *fs_out, double *sor_out) {
    int ref_fwd = 0, ref_rev = 0, alt_fwd = 0, alt_rev = 0;
    for (uint32_t i = 0; i < n; ++i) {
        unsigned b = bs[i] & 3u;
        if (bs[i] & BV_CELL_NOCALL) continue;
        int is_ref = ((int)b == ref_code), is_alt = (alt_mask >> b) & 1u;
        if (!(bs[i] & BV_CELL_REV)) {
            if (is_ref) ++ref_fwd; else if (is_alt) ++alt_fwd;
        } else {
            if (is_ref) ++ref_rev; else if (is_alt) ++alt_rev;
        }
    }
    double fs = -10 * log10(oracle_fisher_exact_test(ref_fwd, ref_rev, alt_fwd, alt_rev));
    if (isinf(fs)) fs = 10000;
    else if (fs == 0) fs = 0.0;
    /* basetype.cpp:286 multiplies `int`s.  Once a product reaches 2^31 that is signed overflow
     * (undefined behaviour); what the reference AS COMPILED (gcc -O3, x86-64: oracle/_ref) does is
     * pinned by tests/golden/deep_sor.npz: the guard `ref_rev * alt_fwd > 0` is folded under the
     * no-overflow assumption into "both factors non-zero", while the quotient uses the 32-bit
     * wrapped products (imul r32).  Written out explicitly here so that this file does not depend
     * on how a compiler treats the UB. */
    int num = (int)((uint32_t)ref_fwd * (uint32_t)alt_rev);
    int den = (int)((uint32_t)ref_rev * (uint32_t)alt_fwd);
    double sor = (ref_rev > 0 && alt_fwd > 0) ? (double)num / (double)den : 10000;
    cnt[0] = ref_fwd; cnt[1] = ref_rev; cnt[2] = alt_fwd; cnt[3] = alt_rev;
    *fs_out = fs;
    *sor_out = sor;
}

/* ref_vs_alt_ranksumtest, basetype.cpp:201-233 */
static double o_ranksum(int ref_code, unsigned alt_mask, const uint8_t *bs, uint32_t n, const uint8_t *v8,
                        const uint16_t *v16) {
    double *ref = (double *)malloc(sizeof(double) * (n ? n : 1));
    double *alt = (double *)malloc(sizeof(double) * (n ? n : 1));
    size_t nr = 0, na = 0;
    for (uint32_t i = 0; i < n; ++i) {
        unsigned b = bs[i] & 3u;
        if (bs[i] & BV_CELL_NOCALL) continue;
        double v = v8 ? (double)v8[i] : (double)v16[i];
        if ((int)b == ref_code) ref[nr++] = v;
        else if ((alt_mask >> b) & 1u) alt[na++] = v;
    }
    double ph;
    if (nr > 0 && na > 0) {
        double p = oracle_wilcoxon(ref, nr, alt, na);
        ph = -10 * log10(p);
        if (isinf(ph)) ph = 10000;
    } else {
        ph = 10000;
    }
    free(ref);
    free(alt);
    return ph;
}

/* One site: the call sequence of _basevar_caller (caller.cpp:738-762), _out_cvg_line
 * (:1236-1245) and _out_vcf_line (:1113-1164). */
static void o_run_site(const uint8_t *bs, const uint8_t *q, const uint8_t *mq, const uint16_t *rp, uint8_t ref_code,
                       const uint8_t *group_id, uint32_t n_groups, uint32_t n, double min_af, bv_site_result *r,
                       bv_group_result *g, double *margin_out) {
    if (margin_out) *margin_out = INFINITY;
    memset(r, 0, sizeof(*r));
    r->mq_ranksum = r->rpr_ranksum = r->bq_ranksum = NAN;
    if (g) memset(g, 0, sizeof(*g) * n_groups);
    uint32_t depth_all = 0;
    for (uint32_t i = 0; i < n; ++i) depth_all += (bs[i] != BV_CELL_N);
    if (depth_all == 0) return; /* caller.cpp:718 */

    int ref = (ref_code < 4) ? (int)ref_code : 4;
    unsigned nonref_mask = 0xFu & ~((ref < 4) ? (1u << ref) : 0u);
    o_strand_bias(ref, nonref_mask, bs, n, r->cvg_sb, &r->cvg_fs, &r->cvg_sor);
    if (r->cvg_sb[0] + r->cvg_sb[1] + r->cvg_sb[2] + r->cvg_sb[3] == 0) { /* no CVG row: caller.cpp:1246 */
        r->cvg_fs = 0;
        r->cvg_sor = 0;
    }

    o_bt bt;
    o_bt_init(&bt, bs, q, n, NULL, 0, min_af);
    static const int ACGT[4] = {0, 1, 2, 3};
    o_lrt(&bt, ACGT, 4, ref);
    if (margin_out) *margin_out = bt.margin;
    for (int j = 0; j < 4; ++j) r->depth[j] = (uint32_t)bt.depth[j];
    r->total_depth = (uint32_t)bt.total_depth;
    if (bt.total_depth > 0) r->status |= BV_SITE_COVERED;
    if (bt.zero_freq) r->status |= BV_SITE_ZERO_FREQ;
    r->em_iters = (uint16_t)bt.em_iters;
    r->n_em = (uint8_t)bt.n_em;
    r->chi2 = bt.chi2;
    for (uint32_t i = 0; i < n; ++i)
        if (!(bs[i] & BV_CELL_NOCALL) && q[i] > BV_MAX_PHRED) r->status |= BV_SITE_BAD_QUAL;

    if (bt.n_alt > 0) {
        r->status |= BV_SITE_VARIANT;
        r->n_alt = (uint8_t)bt.n_alt;
        r->qual = bt.var_qual;
        double ad_sum = 0;
        unsigned alt_mask = 0;
        for (int k = 0; k < bt.n_alt; ++k) {
            int b = bt.alt[k];
            r->alt[k] = (uint8_t)b;
            alt_mask |= 1u << b;
            ad_sum = ad_sum + bt.depth[b];
            r->af[k] = bt.af[k];
            r->caf[k] = bt.depth[b] / bt.total_depth;
        }
        if (mq && rp) {
            r->status |= BV_SITE_RANKSUM;
            r->mq_ranksum = o_ranksum(ref, alt_mask, bs, n, mq, NULL);
            r->rpr_ranksum = o_ranksum(ref, alt_mask, bs, n, NULL, rp);
        }
        r->bq_ranksum = o_ranksum(ref, alt_mask, bs, n, q, NULL); /* +33 offset does not change ranks */
        double qd = bt.var_qual / ad_sum;
        if (qd == 0) qd = 0.0;
        r->qd = qd;
        o_strand_bias(ref, alt_mask, bs, n, r->var_sb, &r->var_fs, &r->var_sor);
        if ((int64_t)r->var_sb[0] * r->var_sb[3] > INT32_MAX || (int64_t)r->var_sb[1] * r->var_sb[2] > INT32_MAX)
            r->status |= BV_SITE_SOR_OVERFLOW;

        if (g && n_groups && group_id) { /* caller.cpp:746-759 */
            int comb[5], nc = 0; /* [REF] + up to four alts (ref not ACGT) */
            comb[nc++] = ref;
            for (int k = 0; k < bt.n_alt; ++k) comb[nc++] = bt.alt[k];
            uint32_t *idx = (uint32_t *)malloc(sizeof(uint32_t) * n);
            for (uint32_t gi = 0; gi < n_groups; ++gi) {
                uint32_t ni = 0;
                for (uint32_t i = 0; i < n; ++i)
                    if (group_id[i] == gi) idx[ni++] = i;
                o_bt gb;
                o_bt_init(&gb, bs, q, n, idx, ni, min_af);
                o_lrt(&gb, comb, nc, ref);
                if (margin_out && gb.margin < *margin_out) *margin_out = gb.margin;
                g[gi].n_alt = (uint8_t)gb.n_alt;
                g[gi].total_depth = (uint32_t)gb.total_depth;
                for (int k = 0; k < gb.n_alt; ++k) {
                    g[gi].alt[k] = (uint8_t)gb.alt[k];
                    g[gi].af[k] = gb.af[k];
                }
                o_bt_free(&gb);
            }
            free(idx);
        }
    }
    if ((int64_t)r->cvg_sb[0] * r->cvg_sb[3] > INT32_MAX || (int64_t)r->cvg_sb[1] * r->cvg_sb[2] > INT32_MAX)
        r->status |= BV_SITE_SOR_OVERFLOW;
    o_bt_free(&bt);
}

typedef struct {
    const uint8_t *bs, *q, *mq, *ref, *gid;
    const uint16_t *rp;
    uint32_t n_groups, n;
    uint64_t pitch, lo, hi;
    double min_af;
    bv_site_result *out;
    bv_group_result *gout;
    double *margins;
} o_job;

static void *o_worker(void *arg) {
    o_job *j = (o_job *)arg;
    for (uint64_t s = j->lo; s < j->hi; ++s)
        o_run_site(j->bs + s * j->pitch, j->q + s * j->pitch, j->mq ? j->mq + s * j->pitch : NULL,
                   j->rp ? j->rp + s * j->pitch : NULL, j->ref[s], j->gid, j->n_groups, j->n, j->min_af, j->out + s,
                   j->gout ? j->gout + s * j->n_groups : NULL, j->margins ? j->margins + s : NULL);
    return NULL;
}

/* Same signature as bvref_run (oracle/ref_driver.cpp) minus the error buffer; `margins` (optional,
 * [n_sites]) receives the decision margin of every site (see o_lrt). */
int oracle_run_ex(const uint8_t *base_strand, const uint8_t *qual, const uint8_t *mapq, const uint16_t *rpr,
                  const uint8_t *ref_base, const uint8_t *group_id, uint32_t n_groups, uint32_t n_sites,
                  uint32_t n_samples, uint64_t pitch, double min_af, bv_site_result *out, bv_group_result *gout,
                  int n_threads, double *margins) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    o_job jobs[256];
    for (int t = 0; t < n_threads; ++t) {
        o_job *j = &jobs[t];
        j->bs = base_strand; j->q = qual; j->mq = mapq; j->rp = rpr; j->ref = ref_base; j->gid = group_id;
        j->n_groups = n_groups; j->n = n_samples; j->pitch = pitch; j->min_af = min_af;
        j->out = out; j->gout = gout; j->margins = margins;
        j->lo = (uint64_t)n_sites * t / n_threads;
        j->hi = (uint64_t)n_sites * (t + 1) / n_threads;
        if (n_threads == 1) o_worker(j);
        else pthread_create(&th[t], NULL, o_worker, j);
    }
    if (n_threads > 1)
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    return 0;
}

int oracle_run(const uint8_t *base_strand, const uint8_t *qual, const uint8_t *mapq, const uint16_t *rpr,
               const uint8_t *ref_base, const uint8_t *group_id, uint32_t n_groups, uint32_t n_sites,
               uint32_t n_samples, uint64_t pitch, double min_af, bv_site_result *out, bv_group_result *gout,
               int n_threads) {
    return oracle_run_ex(base_strand, qual, mapq, rpr, ref_base, group_id, n_groups, n_sites, n_samples, pitch, min_af,
                         out, gout, n_threads, NULL);
}

/* float-rounded min_af, src/basetype_caller.cpp:122 with src/basetype_utils.h:80 */
double oracle_min_af(uint32_t n_samples, float user_min_af) {
    float a = (float)100 / n_samples;
    float m = (a < user_min_af) ? a : user_min_af;
    return (double)m;
}
