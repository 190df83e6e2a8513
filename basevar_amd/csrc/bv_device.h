// bv_device.h -- gfx950 device building blocks of the per-site basetype solver.
//
// Everything here operates on a per-site (base x phred) histogram that the tally loop
// builds in LDS.  The key identity: the reference's per-sample
// likelihood row depends only on (first base in ACGT, phred q) -- src/basetype.cpp:47-64 --
// so its n_cov x 4 EM collapses exactly onto <= 4 x 94 weighted bins.
//
// Execution model: wave64.  Every solver routine is a *wave-level* function: the 64 lanes
// of one wavefront cooperate through __shfl_xor butterflies (all lanes end up with the
// bit-identical sum because IEEE addition is commutative), and independent routines
// (the up-to-4 EM runs of one LRT level, Fisher tests, rank sums) are spread over the
// waves of the workgroup.  No MFMA: this is categorical-table arithmetic in FP64.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/basevar_amd.h"

#ifndef BV_FISHER_ATTR
#define BV_FISHER_ATTR inline
#endif
#define BV_WAVE 64
#define BV_QBINS 128                       /* phred axis of the LDS histogram          */
#define BV_ROWS 8                          /* (reverse << 2) | base                    */
#define BV_HIST_WORDS (BV_ROWS * BV_QBINS) /* 1024 x u32 = 4 KiB                       */
#define BV_NQ_VALID 94                     /* phred 0..93                              */
#define BV_MAX_BINS (4 * BV_NQ_VALID)      /* 376                                      */
#define BV_SLOTS 6                         /* ceil(376 / 64) bins per lane             */

// cell encoding used by the planes (include/basevar_amd.h):
//   bits 0-1 base, bit 2 reverse strand, bit 3 "not a base call" (N / + / -)
#ifndef BV_CELL_NOCALL
#define BV_CELL_NOCALL 0x08u
#endif

// native 16-byte vector: one lane's share of a coalesced 1 KiB wave load
typedef uint32_t bv_u32x4 __attribute__((ext_vector_type(4)));

// (1 - eps_q) and eps_q / 3 for q = 0..127, filled on the host with glibc exp() so that
// eps is bit-identical to the reference's exp((qchar - 33) * MLN10TO10), basetype.cpp:47.
struct BvTables {
    double hit[BV_QBINS];
    double miss[BV_QBINS];
};

// ------------------------------------------------------------------ wave reductions
__device__ __forceinline__ double bv_wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, BV_WAVE);
    return v;
}
__device__ __forceinline__ uint32_t bv_wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, BV_WAVE);
    return v;
}
__device__ __forceinline__ unsigned long long bv_wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += (unsigned long long)__shfl_xor((long long)v, m, BV_WAVE);
    return v;
}
__device__ __forceinline__ int bv_wave_min_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, BV_WAVE));
    return v;
}
__device__ __forceinline__ int bv_wave_max_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, BV_WAVE));
    return v;
}
// inclusive prefix sum over lanes (Hillis-Steele with shfl_up)
__device__ __forceinline__ uint32_t bv_wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < BV_WAVE; d <<= 1) {
        uint32_t t = (uint32_t)__shfl_up((int)v, d, BV_WAVE);
        if (lane >= d) v += t;
    }
    return v;
}

// ------------------------------------------------------------------ special functions
// Device forms of htslib/kfunc.c as the reference calls them.  Same evaluation order as
// the reference so that only libm-vs-ocml ulp differences of exp/log remain.

// kf_lgamma, htslib/kfunc.c:39-52
__device__ inline double bv_kf_lgamma(double z) {
    double x = 0;
    x += 0.1659470187408462e-06 / (z + 7);
    x += 0.9934937113930748e-05 / (z + 6);
    x -= 0.1385710331296526 / (z + 5);
    x += 12.50734324009056 / (z + 4);
    x -= 176.6150291498386 / (z + 3);
    x += 771.3234287757674 / (z + 2);
    x -= 1259.139216722289 / (z + 1);
    x += 676.5203681218835 / z;
    x += 0.9999999999995183;
    return log(x) - 5.58106146679532777 - z + (z - 0.5) * log(z + 6.5);
}

// kf_erfc, htslib/kfunc.c:58-84
__device__ inline double bv_kf_erfc(double x) {
    const double SQRT2 = 1.41421356237309504880;
    double z = fabs(x) * SQRT2;
    if (z > 37.) return x > 0. ? 0. : 2.;
    double expntl = exp(z * z * -.5);
    double p;
    if (z < 10. / SQRT2) {
        double num = .03526249659989109;
        num = num * z + .7003830644436881;
        num = num * z + 6.37396220353165;
        num = num * z + 33.912866078383;
        num = num * z + 112.0792914978709;
        num = num * z + 221.2135961699311;
        num = num * z + 220.2068679123761;
        double den = .08838834764831844;
        den = den * z + 1.755667163182642;
        den = den * z + 16.06417757920695;
        den = den * z + 86.78073220294608;
        den = den * z + 296.5642487796737;
        den = den * z + 637.3336333788311;
        den = den * z + 793.8265125199484;
        den = den * z + 440.4137358247522;
        p = expntl * num / den;
    } else {
        p = expntl / 2.506628274631001 / (z + 1. / (z + 2. / (z + 3. / (z + 4. / (z + .65)))));
    }
    return x > 0. ? 2. * p : 2. * (1. - p);
}

// kf_gammaq, htslib/kfunc.c:103-143
__device__ inline double bv_kf_gammaq(double s, double z) {
    const double EPS = 1e-14, TINY = 1e-290;
    if (z <= 1. || z < s) {
        double sum = 1., x = 1.;
        for (int k = 1; k < 100; ++k) {
            x *= z / (s + k);
            sum += x;
            if (x / sum < EPS) break;
        }
        return 1. - exp(s * log(z) - z - bv_kf_lgamma(s + 1.) + log(sum));
    }
    double f = 1. + z - s, C = f, D = 0.;
    for (int j = 1; j < 100; ++j) {
        double a = j * (s - j), b = (j << 1) + 1 + z - s, d;
        D = b + a * D;
        if (D < TINY) D = TINY;
        C = b + a / C;
        if (C < TINY) C = TINY;
        D = 1. / D;
        d = C * D;
        f *= d;
        if (fabs(d - 1.) < EPS) break;
    }
    return exp(s * log(z) - z - bv_kf_lgamma(s) - log(f));
}

// QUAL from the last LRT statistic, src/basetype.cpp:186-195 (chi2_test: algorithm.h:44-46)
__device__ inline double bv_qual_from_chi2(double chi) {
    double p = bv_kf_gammaq(0.5, chi / 2.0);
    if (isnan(p)) p = 1.0;
    double q = (p != 0.0) ? -10 * log10(p) : 10000.0;
    if (q == 0.0) q = 0.0;  // -0.0 -> 0.0
    return q;
}

// ------------------------------------------------------------------ Fisher exact test
// kt_fisher_exact (two-sided), htslib/kfunc.c:197-313, wave-parallel.
//
// The reference walks the hypergeometric pmf from both ends towards the observed table,
// updating p(i) multiplicatively and re-seeding it from lgamma() whenever i % 11 == 0
// (hypergeo_acc, kfunc.c:220-243).  Here each lane owns one such block of 11 consecutive
// tables (seed + <= 10 multiplicative steps -- the same arithmetic), 64 blocks per sweep;
// the reference's stopping rule ("first p >= 0.99999999 q") becomes a wave-min over the
// lanes' first violating index, valid because the pmf is unimodal.
__device__ inline double bv_lbinom(int n, int k) {
    if (k == 0 || n == k) return 0;
    return lgamma((double)(n + 1)) - lgamma((double)(k + 1)) - lgamma((double)(n - k + 1));
}
__device__ inline double bv_hypergeo(int n11, int n1_, int n_1, int n) {
    return exp(bv_lbinom(n1_, n11) + bv_lbinom(n - n1_, n_1 - n11) - bv_lbinom(n, n_1));
}

__device__ BV_FISHER_ATTR double bv_fisher_two_sided_wave(int n11, int n12, int n21, int n22, int lane) {
#ifdef BV_PROBE_NOFISHER
    return 0.5;
#endif
    const int n1_ = n11 + n12, n_1 = n11 + n21, n = n11 + n12 + n21 + n22;
    const int imax = (n_1 < n1_) ? n_1 : n1_;
    int imin = n1_ + n_1 - n;
    if (imin < 0) imin = 0;
    if (imin == imax) return 1.;
    const double q = bv_hypergeo(n11, n1_, n_1, n);
    if (q == 0.0) return 0.0;  // kfunc.c:260-289
    const double lo = 0.99999999 * q, hi = 1.00000001 * q;
    const int n22off = n - n1_ - n_1;  // n22 of table i is i + n22off

    // ---- left tail: ascending from imin
    double left = 0.;
    {
        int blk0 = imin / 11;  // block b covers [11b, 11b+10]
        for (;;) {
            int b = blk0 + lane;
            int start = max(b * 11, imin), end = min(b * 11 + 10, imax);
            double acc = 0., pv = 0.;
            int viol = 0x7fffffff;
            if (start <= end) {
                double p = bv_hypergeo(start, n1_, n_1, n);
                for (int i = start;; ++i) {
                    if (p < lo) {
                        acc += p;
                    } else {
                        viol = i;
                        pv = p;
                        break;
                    }
                    if (i == end) break;
                    // incremental step i -> i+1 (kfunc.c:226-231)
                    p *= (double)(n1_ - i) / (i + 1) * (n_1 - i) / (i + 1 + n22off);
                }
            }
            int first = bv_wave_min_i32(viol);
            // unimodal pmf: blocks that start at or before the first violating table hold only
            // tail terms (the block containing it stopped accumulating there); later blocks none
            double contrib = (start <= end && start <= first) ? acc : 0.;
            left += bv_wave_sum(contrib);
            if (first != 0x7fffffff) {
                double pb = bv_wave_sum(viol == first ? pv : 0.);  // exactly one lane holds it
                if (pb < hi) left += pb;
                break;
            }
            blk0 += BV_WAVE;
            if (blk0 * 11 > imax) break;
        }
    }
    // ---- right tail: descending from imax; block b covers [11b+1, 11b+11], seeded at its top
    double right = 0.;
    {
        int blk0 = (imax - 1) / 11;  // block containing imax (imax >= 1 here since imin < imax)
        for (;;) {
            int b = blk0 - lane;
            int start = min(b * 11 + 11, imax), end = max(b * 11 + 1, imin);
            bool have = (b >= 0) && (start >= end);
            // table 0 belongs to block -1 (index 0 is a multiple of 11: its own seed)
            if (b == -1 && imin == 0) {
                start = 0;
                end = 0;
                have = true;
            }
            double acc = 0., pv = 0.;
            int viol = -1;
            if (have) {
                double p = bv_hypergeo(start, n1_, n_1, n);
                for (int j = start;; --j) {
                    if (p < lo) {
                        acc += p;
                    } else {
                        viol = j;
                        pv = p;
                        break;
                    }
                    if (j == end) break;
                    // decremental step j -> j-1 (kfunc.c:232-237)
                    p *= (double)j / (n1_ - (j - 1)) * (j + n22off) / (n_1 - (j - 1));
                }
            }
            int first = bv_wave_max_i32(viol);
            double contrib = (have && start >= first) ? acc : 0.;
            right += bv_wave_sum(contrib);
            if (first >= 0) {
                double pb = bv_wave_sum(viol == first ? pv : 0.);
                if (pb < hi) right += pb;
                break;
            }
            blk0 -= BV_WAVE;
            if (blk0 < -1) break;
        }
    }
    double two = left + right;
    if (two > 1.) two = 1.;
    return two;
}

// strand_bias tail, src/basetype.cpp:277-286
__device__ inline void bv_strand_bias_wave(uint32_t ref_fwd, uint32_t ref_rev, uint32_t alt_fwd, uint32_t alt_rev,
                                           int lane, double *fs_out, double *sor_out, uint32_t *flags) {
    double fs = -10 * log10(bv_fisher_two_sided_wave((int)ref_fwd, (int)ref_rev, (int)alt_fwd, (int)alt_rev, lane));
    if (isinf(fs)) fs = 10000;
    else if (fs == 0) fs = 0.0;
    // `int` products as in the reference (wrap like x86 imul; flagged because it is UB there)
    int den = (int)(ref_rev * alt_fwd), num = (int)(ref_fwd * alt_rev);
    if ((unsigned long long)ref_rev * alt_fwd > 0x7fffffffull || (unsigned long long)ref_fwd * alt_rev > 0x7fffffffull)
        *flags |= BV_SITE_SOR_OVERFLOW;
    double sor = (den > 0) ? (double)num / (double)den : 10000;
    *fs_out = fs;
    *sor_out = sor;
}

// ------------------------------------------------------------------ Wilcoxon rank sum
// ref_vs_alt_ranksumtest -> wilcoxon_ranksum_test, src/basetype.cpp:201-233 and
// src/algorithm.h:76-136, evaluated from value histograms.  With descending order and
// average ranks for ties the rank of value v is above_v + (t_v + 1) / 2, i.e.
// n - below_v - t_v + (t_v + 1)/2; twice the ref rank sum is an exact integer:
//     2 R_ref = sum_v ref_v * (2 n - 2 below_v - t_v + 1)
// (bit-identical to the sort-based reference).
//
// Streaming form: feed values in ascending order, 64 per call (lane = value), carrying
// `below`.  Returns this window's contribution to 2 R_ref and updates below.
__device__ inline unsigned long long bv_ranksum_window(uint32_t ref_v, uint32_t alt_v, unsigned long long n,
                                                       unsigned long long &below, int lane) {
    uint32_t t = ref_v + alt_v;
    uint32_t incl = bv_wave_incl_scan_u32(t, lane);
    unsigned long long below_v = below + (incl - t);
    unsigned long long term = (unsigned long long)ref_v * (2ull * n - 2ull * below_v - t + 1ull);
    unsigned long long s = bv_wave_sum_u64(term);
    below += (unsigned long long)__shfl((int)incl, BV_WAVE - 1, BV_WAVE);
    return s;
}
// z statistic -> phred, algorithm.h:130-132 + basetype.cpp:222-231
__device__ inline double bv_ranksum_phred(unsigned long long twoR, unsigned long long n1, unsigned long long n2) {
    if (n1 == 0 || n2 == 0) return 10000;
    double r1 = (double)twoR / 2.0;
    double e = (double)(n1 * (n1 + n2 + 1)) / 2.0;
    double z = (r1 - e) / sqrt((double)(n1 * n2 * (n1 + n2 + 1)) / 12.0);
    double p = 2 * (bv_kf_erfc((double)(fabs(z) / sqrt(2.0))) / 2.0);
    double ph = -10 * log10(p);
    if (isinf(ph)) ph = 10000;
    return ph;
}

// ------------------------------------------------------------------ EM on weighted bins
// The compacted non-empty histogram bins live in LDS (bin i: code = base<<7 | phred, count);
// bin i is handled by lane i % 64 in slot i / 64.  Only the per-bin log-marginal of the
// previous iteration is carried in registers, so the solver stays light on VGPRs and the
// streaming phase of the same kernel keeps its occupancy.
struct BvBins {
    const uint32_t *code;  // LDS
    const uint32_t *cnt;   // LDS
    const double *hit;     // LDS copy of BvTables::hit  (1 - eps)
    const double *miss;    // LDS copy of BvTables::miss (eps / 3)
    int nb;                // number of bins (wave-uniform)
};

// The reference's convergence term: algorithm.h:245 binds to int abs(int), so the double
// difference is truncated to int first (x86 cvttsd2si: NaN / out-of-range -> INT_MIN,
// and abs(INT_MIN) stays INT_MIN).
__device__ __forceinline__ double bv_int_abs_trunc(double d) {
    if (isnan(d) || fabs(d) >= 2147483648.0) return -2147483648.0;
    int t = (int)d;
    return (double)(t < 0 ? -t : t);
}

// EM, algorithm.h:210-255.  f: initial freqs in, final freqs out.  Returns loop iterations.
// Pass k = 0 is the reference's pre-loop e_step/m_step (algorithm.h:226-234); passes
// k = 1..100 are the `while (iter_num--)` body (algorithm.h:235-251).  Each pass is one
// fused sweep over the bins: e_step (algorithm.h:161-171), the m_step numerators
// (algorithm.h:190-193), log-marginals and the convergence sum (algorithm.h:243-247).
__device__ __forceinline__ int bv_em_wave(const BvBins &B, double f[4], double n_cov, double *lr_out, int lane) {
    const double epsilon = (double)0.001f;  // `const float epsilon=0.001`, algorithm.h:213
    const int nslots = (B.nb + BV_WAVE - 1) / BV_WAVE;
    double prev[BV_SLOTS];
#pragma unroll
    for (int s = 0; s < BV_SLOTS; ++s) prev[s] = 0.;
    int iters = 0;
    double lr = 0.;
    for (int k = 0; k <= 100; ++k) {
        double pf0 = 0., pf1 = 0., pf2 = 0., pf3 = 0., delta = 0.;
        lr = 0.;
#pragma unroll
        for (int s = 0; s < BV_SLOTS; ++s) {
            if (s < nslots) {
                const int i = s * BV_WAVE + lane;
                if (i < B.nb) {
                    const uint32_t code = B.code[i];
                    const double c = (double)B.cnt[i];
                    const uint32_t b = code >> 7;
                    const double hit = B.hit[code & 127u], miss = B.miss[code & 127u];
                    double L0 = (b == 0 ? hit : miss) * f[0];
                    double L1 = (b == 1 ? hit : miss) * f[1];
                    double L2 = (b == 2 ? hit : miss) * f[2];
                    double L3 = (b == 3 ? hit : miss) * f[3];
                    double marg = L0;  // 0 + L0, then += in j order (algorithm.h:162-165)
                    marg += L1;
                    marg += L2;
                    marg += L3;
                    // one IEEE division then four multiplies (reference: four divisions; <= 1 ulp apart)
                    double r = 1.0 / marg;
                    pf0 += c * (L0 * r);
                    pf1 += c * (L1 * r);
                    pf2 += c * (L2 * r);
                    pf3 += c * (L3 * r);
                    double llh = log(marg);
                    delta += c * bv_int_abs_trunc(llh - prev[s]);
                    lr += c * llh;
                    prev[s] = llh;
                }
            }
        }
        f[0] = bv_wave_sum(pf0) / n_cov;  // m_step, algorithm.h:194
        f[1] = bv_wave_sum(pf1) / n_cov;
        f[2] = bv_wave_sum(pf2) / n_cov;
        f[3] = bv_wave_sum(pf3) / n_cov;
        if (k == 0) continue;  // llh^0 is only the baseline of the first delta
        delta = bv_wave_sum(delta);
        ++iters;
        if (delta < epsilon) break;
    }
    // final m_step (algorithm.h:253) recomputes f from the unchanged posteriors: idempotent.
    *lr_out = bv_wave_sum(lr);  // sum(log_marginal_likelihood), basetype.cpp:120
    return iters;
}

// ------------------------------------------------------------------ LRT
// BaseType::lrt + _f, src/basetype.cpp:105-199.  The EM runs of one level (the n-subsets of
// the current active set, Combinations order: external/combinations.h:55-69) are independent.
//   NW >= 1: "block mode" -- the runs of a level are dealt to the NW waves of the workgroup,
//            results meet in LDS behind __syncthreads(); every thread then replays the cheap,
//            uniform argmin / threshold decision.
//   NW == 0: "wave mode"  -- the calling wave does every run itself (used for the pop-group
//            calls of pass 2, where each wave owns a different group); `sh` is wave-private.
struct BvLrtShared {
    double f[2][4][4];  // [level parity][combination][base]
    double lr[2][4];
    int iters[2][4];
};

struct BvLrtOut {
    int n_alt;
    int alt_packed;  // alt base k in bits [2k, 2k+1], reference order
    double af[4];    // only ever indexed with compile-time constants (registers, no scratch)
    int m;           // final active-set size
    int first;       // active_bases[0]
    double chi2;     // last chi_sqrt_value
    int em_iters, n_em;
    bool zero_freq;
};
__device__ __forceinline__ int bv_alt_at(const BvLrtOut &o, int k) { return (o.alt_packed >> (2 * k)) & 3; }

// 4-entry register files addressed by a run-time base code without touching scratch
__device__ __forceinline__ double bv_sel4(double v0, double v1, double v2, double v3, int i) {
    return i == 0 ? v0 : (i == 1 ? v1 : (i == 2 ? v2 : v3));
}
__device__ __forceinline__ uint32_t bv_sel4u(const uint32_t v[4], int i) {
    return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : v[3]));
}

template <int NW>
__device__ __forceinline__ void bv_lrt_sync() {
    if (NW > 0) {
        __syncthreads();
    } else {
        // same wave writes (lane 0) then reads (all lanes): LDS is in-order per wave; just
        // stop the compiler from moving the accesses across this point
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// `specific_packed`: the candidate bases in reference order, 3 bits each (value 4 = a base
// that is not ACGT, never active); `nspec` of them.
template <int NW>
__device__ inline void bv_lrt(const BvBins &B, const uint32_t depth[4], uint32_t total, int specific_packed,
                              int nspec, int ref_code, double min_af, BvLrtShared *sh, int wave, int lane,
                              BvLrtOut &o) {
    o.n_alt = 0; o.alt_packed = 0; o.af[0] = o.af[1] = o.af[2] = o.af[3] = 0.;
    o.m = 0; o.first = 0; o.chi2 = 0.; o.em_iters = 0; o.n_em = 0; o.zero_freq = false;
    // active_bases as a packed ordered list: position k in bits [2k, 2k+1]
    int act = 0, m = 0;
    for (int k = 0; k < nspec; ++k) {
        int b = (specific_packed >> (3 * k)) & 7;
        if (b < 4 && (double)bv_sel4u(depth, b) / (int)total >= min_af) {  // basetype.cpp:137
            act |= b << (2 * m);
            ++m;
        }
    }
    if (m == 0) return;
    const double n_cov = (double)total;
    const int m0 = m;
    const int first_c = (NW > 0) ? wave : 0, step_c = (NW > 0) ? NW : 1;
    // initial frequency of every base, depth/total (basetype.cpp:99)
    const double d0 = (double)depth[0] / n_cov, d1 = (double)depth[1] / n_cov, d2 = (double)depth[2] / n_cov,
                 d3 = (double)depth[3] / n_cov;

    double fr0 = 0., fr1 = 0., fr2 = 0., fr3 = 0.;  // active_bases_freq
    double lr_alt = 0., chi = 0.;
    int par = 0;
    // n == m0: F_m over the full active set (basetype.cpp:144); n < m0: the loop of :151-169,
    // whose bound is the ORIGINAL size while the subsets are drawn from the current set.
    for (int n = m0; n > 0; --n) {
        const bool top = (n == m0);
        const int ncomb = top ? 1 : m;  // the m subsets of size m-1; subset c drops position m-1-c
        for (int c = first_c; c < ncomb; c += step_c) {
            const int drop = top ? -1 : m - 1 - c;
            unsigned in_set = 0;  // bit b set <=> base b is in this subset
            for (int k = 0; k < m; ++k)
                if (k != drop) in_set |= 1u << ((act >> (2 * k)) & 3);
            double f[4];
            f[0] = (in_set & 1u) ? d0 : 0.;
            f[1] = (in_set & 2u) ? d1 : 0.;
            f[2] = (in_set & 4u) ? d2 : 0.;
            f[3] = (in_set & 8u) ? d3 : 0.;
            double s = 0.;
            s += f[0]; s += f[1]; s += f[2]; s += f[3];
            double lr;
            int it = bv_em_wave(B, f, n_cov, &lr, lane);
            if (lane == 0) {
                sh->f[par][c][0] = f[0]; sh->f[par][c][1] = f[1];
                sh->f[par][c][2] = f[2]; sh->f[par][c][3] = f[3];
                sh->lr[par][c] = lr;
                sh->iters[par][c] = (s == 0.) ? -it - 1 : it;  // negative marks basetype.cpp:113-115
            }
        }
        bv_lrt_sync<NW>();
        int i_min = 0;
        double chi_min = 0.;
        for (int c = 0; c < ncomb; ++c) {
            int it = sh->iters[par][c];
            if (it < 0) { o.zero_freq = true; it = -it - 1; }
            o.em_iters += it;
            o.n_em += 1;
            if (!top) {
                double v = 2 * (lr_alt - sh->lr[par][c]);
                if (c == 0 || v < chi_min) { chi_min = v; i_min = c; }  // first minimum, algorithm.h:24-27
            }
        }
        lr_alt = sh->lr[par][i_min];
        bool accept = top;
        if (!top) {
            chi = chi_min;
            accept = chi < 24;  // LRT_THRESHOLD, basetype.h:21
            if (accept) {
                const int drop = m - 1 - i_min;
                const int low = act & ((1 << (2 * drop)) - 1), high = act >> (2 * drop + 2);
                act = low | (high << (2 * drop));
                m = n;
            }
        }
        if (!accept) break;
        fr0 = sh->f[par][i_min][0]; fr1 = sh->f[par][i_min][1];
        fr2 = sh->f[par][i_min][2]; fr3 = sh->f[par][i_min][3];
        par ^= 1;
    }
    o.chi2 = chi;
    o.m = m;
    o.first = act & 3;
    double af0 = 0., af1 = 0., af2 = 0., af3 = 0.;
    int na = 0, packed = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < m) {
            const int b = (act >> (2 * k)) & 3;
            if (b != ref_code) {  // basetype.cpp:172-177
                const double v = bv_sel4(fr0, fr1, fr2, fr3, b);
                if (na == 0) af0 = v; else if (na == 1) af1 = v; else if (na == 2) af2 = v; else af3 = v;
                packed |= b << (2 * na);
                ++na;
            }
        }
    }
    o.n_alt = na;
    o.alt_packed = packed;
    o.af[0] = af0; o.af[1] = af1; o.af[2] = af2; o.af[3] = af3;
}
