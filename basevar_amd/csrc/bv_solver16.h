// bv_solver16.h -- the per-site solver on a GROUP of 16 lanes: four sites per wave, in lockstep.
//
// A short-row candidate (bv_pass1_short.hip) is a small problem: ~70 (base, phred) bins, Fisher families of a few
// hundred tables, 128 phred values.  On a whole wave most of its solver is scalar work that 64 lanes repeat (the LRT's
// control flow, QUAL's continued fraction, divisions, the tails of six-step reductions over mostly idle lanes) -- the
// solve kernel is VALU-issue-bound, so instruction count is its time.  Here a row of 16 lanes (one DPP row) owns a site:
// every instruction serves four sites, reductions are four DPP steps inside the row, and a site's bins live in the
// registers of its 16 lanes (bin i in lane i % 16, slot i / 16: at most 8 x 16 = 128 bins -- deeper or shallower
// candidates, or ones with a phred-0 call, keep the wave solver).
//
// Same reference functions, same arithmetic per term as the wave solver (bv_device.h / bv_solver.h); only the ORDER of
// the floating-point sums over bins / tables / phred values differs (tree inside a row instead of a scan over the wave),
// i.e. results agree with the wave solver to ~1e-15 relative, far inside the 1e-6 parity bar.
//
// Cross-lane rules: everything "uniform" here is uniform per GROUP and lives in VGPRs; control flow may diverge between
// the four groups of a wave (the compiler masks), so only row-local operations are used: DPP quad_perm / row_mirror /
// row_shr, and ballots cut to the group's 16 bits.
#pragma once

#include "bv_solver.h"


// ------------------------------------------------------------------ row-local primitives
#define BV_DPP_QUAD_XOR1 0xB1 /* quad_perm [1,0,3,2] */
#define BV_DPP_QUAD_XOR2 0x4E /* quad_perm [2,3,0,1] */
#define BV_DPP_ROW_MIRROR 0x140
#define BV_DPP_ROW_HALF_MIRROR 0x141

// All-reduce over the 16 lanes of a row.  Every step pairs lanes symmetrically (i <-> i^1, i^2, 7-i, 15-i) and IEEE
// addition is commutative, so all 16 lanes end with the bit-identical sum -- no broadcast needed.
// L: the lanes that own the item -- 16 (a whole DPP row), or 8 / 4 (aligned parts of one: the pop-group solver for small groups,
// bv_p2g_solve_small_kernel): the first log2(L) steps of the same pairing.
template <int L = 16>
__device__ __forceinline__ double bv_g16_sum(double v) {
    v += bv_dpp_f64<BV_DPP_QUAD_XOR1, 0xf>(v);
    v += bv_dpp_f64<BV_DPP_QUAD_XOR2, 0xf>(v);
    if (L >= 8) v += bv_dpp_f64<BV_DPP_ROW_HALF_MIRROR, 0xf>(v);
    if (L >= 16) v += bv_dpp_f64<BV_DPP_ROW_MIRROR, 0xf>(v);
    return v;
}
__device__ __forceinline__ uint32_t bv_g16_sum_u32(uint32_t v) {
    v += (uint32_t)bv_dpp_i32<BV_DPP_QUAD_XOR1, 0xf>(0, (int)v);
    v += (uint32_t)bv_dpp_i32<BV_DPP_QUAD_XOR2, 0xf>(0, (int)v);
    v += (uint32_t)bv_dpp_i32<BV_DPP_ROW_HALF_MIRROR, 0xf>(0, (int)v);
    v += (uint32_t)bv_dpp_i32<BV_DPP_ROW_MIRROR, 0xf>(0, (int)v);
    return v;
}
__device__ __forceinline__ unsigned long long bv_g16_sum_u64(unsigned long long v) {
    v += bv_dpp_u64<BV_DPP_QUAD_XOR1, 0xf>(v);
    v += bv_dpp_u64<BV_DPP_QUAD_XOR2, 0xf>(v);
    v += bv_dpp_u64<BV_DPP_ROW_HALF_MIRROR, 0xf>(v);
    v += bv_dpp_u64<BV_DPP_ROW_MIRROR, 0xf>(v);
    return v;
}
__device__ __forceinline__ int bv_g16_max_i32(int v) {
    v = max(v, bv_dpp_i32<BV_DPP_QUAD_XOR1, 0xf>(v, v));
    v = max(v, bv_dpp_i32<BV_DPP_QUAD_XOR2, 0xf>(v, v));
    v = max(v, bv_dpp_i32<BV_DPP_ROW_HALF_MIRROR, 0xf>(v, v));
    v = max(v, bv_dpp_i32<BV_DPP_ROW_MIRROR, 0xf>(v, v));
    return v;
}
__device__ __forceinline__ int bv_g16_min_i32(int v) {
    v = min(v, bv_dpp_i32<BV_DPP_QUAD_XOR1, 0xf>(v, v));
    v = min(v, bv_dpp_i32<BV_DPP_QUAD_XOR2, 0xf>(v, v));
    v = min(v, bv_dpp_i32<BV_DPP_ROW_HALF_MIRROR, 0xf>(v, v));
    v = min(v, bv_dpp_i32<BV_DPP_ROW_MIRROR, 0xf>(v, v));
    return v;
}
// inclusive prefix PRODUCTS inside the row
__device__ __forceinline__ double bv_g16_incl_scan_prod_f64(double v) {
    v *= bv_dpp_f64_one<BV_DPP_ROW_SHR(1), 0xf>(v);
    v *= bv_dpp_f64_one<BV_DPP_ROW_SHR(2), 0xf>(v);
    v *= bv_dpp_f64_one<BV_DPP_ROW_SHR(4), 0xf>(v);
    v *= bv_dpp_f64_one<BV_DPP_ROW_SHR(8), 0xf>(v);
    return v;
}
// inclusive prefix sum inside the row (lane 15 of the row ends with the total)
__device__ __forceinline__ uint32_t bv_g16_incl_scan_u32(uint32_t v) {
    v += (uint32_t)bv_dpp_i32<BV_DPP_ROW_SHR(1), 0xf>(0, (int)v);
    v += (uint32_t)bv_dpp_i32<BV_DPP_ROW_SHR(2), 0xf>(0, (int)v);
    v += (uint32_t)bv_dpp_i32<BV_DPP_ROW_SHR(4), 0xf>(0, (int)v);
    v += (uint32_t)bv_dpp_i32<BV_DPP_ROW_SHR(8), 0xf>(0, (int)v);
    return v;
}
// the 16 ballot bits of this lane's group
__device__ __forceinline__ uint32_t bv_g16_ballot(bool pred, int lane) {
    return (uint32_t)(__ballot(pred) >> (lane & 48)) & 0xFFFFu;
}
// value held by lane `k` (0..15, uniform in the group) of this lane's group
__device__ __forceinline__ double bv_g16_bcast_f64(double v, int k, int lane) {
    const int src = ((lane & 48) | (k & 15)) << 2;
    const int lo = __builtin_amdgcn_ds_bpermute(src, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(src, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// ------------------------------------------------------------------ Fisher exact test, 16 tables per round
// kt_fisher_exact (two-sided), htslib/kfunc.c:197-313; see bv_fisher_two_sided_wave for the formulation.  `T` as there.
__device__ inline double bv_fisher_two_sided_g16(int n11, int n12, int n21, int n22, int lane, const BvLnTab &T) {
    const int gl = lane & 15;
    const int n1_ = n11 + n12, n_1 = n11 + n21, n = n11 + n12 + n21 + n22;
    const int imax = (n_1 < n1_) ? n_1 : n1_;
    int imin = n1_ + n_1 - n;
    if (imin < 0) imin = 0;
    if (imin == imax) return 1.;
    {
        // a row margin of at most 12 reads: product form, at most 13 tables -- one per lane of the group
        const int n2_ = n21 + n22, n_2 = n - n_1;
        const bool alt_row = n2_ <= n1_;
        const int m = alt_row ? n2_ : n1_, jobs = alt_row ? n21 : n11;
        if (m <= 12) {
            const int jmin = max(0, m - n_2), jmax = min(m, n_1);
            const int j = jmin + gl;
            const bool have = j <= jmax;
            double pn = 1.0, pd = 1.0;
            for (int t = 0; t < m; ++t) {
                const double num = (t < j) ? (double)(n_1 - t) * (double)(m - t) : (double)(n_2 - (t - j));
                const double den = (t < j) ? (double)(n - t) * (double)(t + 1) : (double)(n - t);
                pn *= num;
                pd *= den;
            }
            const double p = have ? pn / pd : 0.;
            const double q = bv_g16_bcast_f64(p, jobs - jmin, lane);
            if (q == 0.0) return 0.0;
            const double lo = 0.99999999 * q, hi = 1.00000001 * q;
            const bool viol = have && !(p < lo);
            const uint32_t vm = bv_g16_ballot(viol, lane);  // never empty: the observed table is in it
            double two = bv_g16_sum(viol ? 0. : p);
            const double pL = bv_g16_bcast_f64(p, __builtin_ctz(vm), lane), pR = bv_g16_bcast_f64(p, 31 - __builtin_clz(vm), lane);
            if (pL < hi) two += pL;
            if (pR < hi) two += pR;
            return two > 1. ? 1. : two;
        }
    }
    BvHyper h;
    bv_hyper_init(h, T, n1_, n_1, n);
    const int R = imax - imin + 1;
    // Up to 128 tables (no tails to skip: the blocks below are those of [imin, imax]): the lane's seed table is known here,
    // and its four log-factorials travel with those of q instead of making a memory trip of their own after it.
    const bool narrow = R <= 8 * 16;
    const int i0n = imin + gl * ((R + 15) >> 4);
    double seed_n = (narrow && i0n <= imax) ? bv_hyper_logp(h, i0n) : 0.;
    asm volatile("" : "+v"(seed_n));  // (read here, not where it is used)
    const double logq = bv_hyper_logp(h, n11);
    const double q = exp(logq);
    if (q == 0.0) return 0.0;  // kfunc.c:260-289
    const double lo = 0.99999999 * q, hi = 1.00000001 * q;
    const int INF = 0x7fffffff;
    int wl = imin, wr = imax;
    if (!narrow) {
        // many tables: skip the far tails whose terms are below q * 2^-86 (16-point probes on either side)
        const double cut = logq - 60.0;
        {
            const int span = n11 - imin, step = span / 15 + 1;
            const int i = imin + gl * step;
            const bool below = (i <= n11) && (bv_hyper_logp(h, min(i, n11)) < cut);
            const int last = bv_g16_max_i32(below ? i : -1);
            if (last >= 0) wl = last;
        }
        {
            const int span = imax - n11, step = span / 15 + 1;
            const int i = imax - gl * step;
            const bool below = (i >= n11) && (bv_hyper_logp(h, max(i, n11)) < cut);
            const int first = bv_g16_min_i32(below ? i : INF);
            if (first != INF) wr = first;
        }
    }
    // The tables of [wl, wr] in 16 consecutive blocks, one per lane: the lane seeds its first table from log-factorials and
    // walks its block with the multiplicative step of the reference's own walk (hypergeo_acc, kfunc.c:226-231) -- at most
    // ~25 steps from an exact seed (the reference re-seeds every 11).  Per table: one division and a few multiplies on ONE
    // lane, against a 16-lane prefix product, two broadcasts and a ballot per 16 tables in the round form this replaces
    // (the Fisher tests were 45 % of the solver's instructions).
    const int Lb = (wr - wl + 16) >> 4;  // tables per lane
    const int i0 = wl + gl * Lb;
    const int i1 = min(i0 + Lb - 1, wr);
    const bool mine = i0 <= wr;
    double p = mine ? exp(narrow ? seed_n : bv_hyper_logp(h, i0)) : 0.;
    double tail = 0., pfirst = 0., plast = 0.;
    bool seen = false;
    for (int t = 0; t < Lb; ++t) {
        const int i = i0 + t;
        const bool in = mine && i <= i1;
        const bool viol = in && !(p < lo);
        tail += (in && !viol) ? p : 0.;
        if (viol && !seen) pfirst = p;
        if (viol) { plast = p; seen = true; }
        p *= bv_hyper_ratio(h, i);
    }
    const uint32_t vm = bv_g16_ballot(seen, lane);  // never empty: the observed table lies inside [wl, wr]
    double two = bv_g16_sum(tail);
    const double pL = bv_g16_bcast_f64(pfirst, __builtin_ctz(vm | 0x10000u), lane);
    const double pR = bv_g16_bcast_f64(plast, 31 - __builtin_clz(vm | 1u), lane);
    if (vm != 0u) {
        if (pL < hi) two += pL;
        if (pR < hi) two += pR;
    }
    return two > 1. ? 1. : two;
}

// strand_bias tail, src/basetype.cpp:277-286 (see bv_strand_bias_wave for the SOR overflow note)
__device__ inline void bv_strand_bias_g16(uint32_t ref_fwd, uint32_t ref_rev, uint32_t alt_fwd, uint32_t alt_rev, int lane,
                                          const BvLnTab &T, double *fs_out, double *sor_out, uint32_t *flags) {
    double fs = -10 * log10(bv_fisher_two_sided_g16((int)ref_fwd, (int)ref_rev, (int)alt_fwd, (int)alt_rev, lane, T));
    if (isinf(fs)) fs = 10000;
    else if (fs == 0) fs = 0.0;
    int den = (int)(ref_rev * alt_fwd), num = (int)(ref_fwd * alt_rev);
    if ((unsigned long long)ref_rev * alt_fwd > 0x7fffffffull || (unsigned long long)ref_fwd * alt_rev > 0x7fffffffull)
        *flags |= BV_SITE_SOR_OVERFLOW;
    *fs_out = fs;
    *sor_out = (ref_rev != 0u && alt_fwd != 0u) ? (double)num / (double)den : 10000;
}

// ------------------------------------------------------------------ EM on the group's bins (in registers)
// bin word = code << 16 | count (code = base << 7 | phred), 0 for an empty slot; bins with phred > 93 carry no likelihood
// row of their own in the EM (as in the wave solver, which leaves them out of its bin list) and are masked here.
struct BvG16Bins {
    uint32_t w[BV_G16_SLOTS];
    const double *hit, *miss;        // LDS tables
    const double *loghit, *logmiss;  // device memory
    double *pm;                      // LDS: this LANE's previous marginals, pm[s * 16] for slot s, inside its GROUP's scratch
                                     // (BV_G16_GRP_WORDS; kept out of the registers: the solver is occupancy-bound)
};
__device__ __forceinline__ bool bv_g16_bin(const BvG16Bins &B, int s, uint32_t &b, uint32_t &q, double &c) {
    const uint32_t w = B.w[s];
    q = (w >> 16) & 127u;
    b = w >> 23;
    c = (double)(w & 0xFFFFu);
    return (w & 0xFFFFu) != 0u && q < BV_NQ_VALID;
}

// EM, algorithm.h:210-255, on the bins (no phred-0 call among them, start frequencies not all zero -- the wave solver
// takes the other sites).  `in_set`: bit b = base b; bases outside keep frequency +0.0 (their terms are exact zeros).
// NS: slots in use (every group of the wave has at most 16 * NS bins): the loops over the slots stop there.
// TWO: with the two-base form below (the pop-group solve kernel; the kernels that solve whole sites beside streaming waves sit at
// their register limit -- with it their solvers spilled 18-23 registers into the dependent chains -- and keep the general form).
template <int NS = BV_G16_SLOTS, bool TWO = false, int L = 16>
__device__ __forceinline__ int bv_em_g16(const BvG16Bins &B, double f[4], unsigned in_set, double n_cov, double *lr_out) {
    const double epsilon = (double)0.001f;
    const double inv_n = 1.0 / n_cov;
    double *pm = B.pm;
#pragma unroll
    for (int s = 0; s < NS; ++s) pm[s * 16] = 1.0;
    int iters = 0;
    if (TWO && __popc(in_set) == 2) {
        // Two bases in the subset -- the EM of nearly every item (REF + one ALT): the two bases outside keep frequency +0.0, their
        // likelihood terms are exact +0.0 and adding them changes no bit of the marginal or of a posterior sum, so they are left
        // out: half the multiplies, two row reductions instead of four per iteration.  Bit-identical to the general form below.
        const int ba = __builtin_ctz(in_set), bb = 31 - __builtin_clz(in_set);
        double fa = bv_sel4(f[0], f[1], f[2], f[3], ba), fb = bv_sel4(f[0], f[1], f[2], f[3], bb);
        for (int k = 0; k <= 100; ++k) {
            double pfa = 0., pfb = 0., delta = 0.;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                uint32_t b, q;
                double c;
                if (bv_g16_bin(B, s, b, q, c)) {
                    const double hit = B.hit[q], miss = B.miss[q];
                    const double La = ((int)b == ba ? hit : miss) * fa, Lb = ((int)b == bb ? hit : miss) * fb;
                    const double marg = La + Lb;
                    const double r = c / marg;
                    pfa += La * r; pfb += Lb * r;
                    const double old = pm[s * 16];
                    if (k > 0 && !(marg < old * 2.7 && marg > old * 0.37)) delta += c * bv_int_abs_trunc(log(marg) - log(old));
                    pm[s * 16] = marg;
                }
            }
            fa = bv_g16_sum<L>(pfa) * inv_n; fb = bv_g16_sum<L>(pfb) * inv_n;
            if (k == 0) continue;
            ++iters;
            if (bv_g16_sum<L>(delta) < epsilon) break;
        }
        f[0] = (ba == 0) ? fa : ((bb == 0) ? fb : 0.); f[1] = (ba == 1) ? fa : ((bb == 1) ? fb : 0.);
        f[2] = (ba == 2) ? fa : ((bb == 2) ? fb : 0.); f[3] = (ba == 3) ? fa : ((bb == 3) ? fb : 0.);
    } else
    for (int k = 0; k <= 100; ++k) {
        double pf0 = 0., pf1 = 0., pf2 = 0., pf3 = 0., delta = 0.;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            uint32_t b, q;
            double c;
            if (bv_g16_bin(B, s, b, q, c)) {
                const double hit = B.hit[q], miss = B.miss[q];
                const double L0 = (b == 0 ? hit : miss) * f[0], L1 = (b == 1 ? hit : miss) * f[1];
                const double L2 = (b == 2 ? hit : miss) * f[2], L3 = (b == 3 ? hit : miss) * f[3];
                double marg = L0;
                marg += L1; marg += L2; marg += L3;
                const double r = c / marg;
                pf0 += L0 * r; pf1 += L1 * r; pf2 += L2 * r; pf3 += L3 * r;
                const double old = pm[s * 16];
                if (k > 0 && !(marg < old * 2.7 && marg > old * 0.37)) delta += c * bv_int_abs_trunc(log(marg) - log(old));
                pm[s * 16] = marg;
            }
        }
        pf0 = bv_g16_sum<L>(pf0); pf1 = bv_g16_sum<L>(pf1); pf2 = bv_g16_sum<L>(pf2); pf3 = bv_g16_sum<L>(pf3);
        f[0] = (in_set & 1u) ? pf0 * inv_n : 0.;
        f[1] = (in_set & 2u) ? pf1 * inv_n : 0.;
        f[2] = (in_set & 4u) ? pf2 * inv_n : 0.;
        f[3] = (in_set & 8u) ? pf3 * inv_n : 0.;
        if (k == 0) continue;
        ++iters;
        if (bv_g16_sum<L>(delta) < epsilon) break;  // zero unless some bin's log-marginal moved by >= 1
    }
    // (Tried, round 6, for small groups: ONE logarithm per lane, of the product of the lane's marginals -- a third of an EM run's
    // instructions on paper, +0-2 % measured, two more spilled registers: not kept.)
    double lr = 0.;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        uint32_t b, q;
        double c;
        if (bv_g16_bin(B, s, b, q, c)) lr += c * log(pm[s * 16]);
    }
    *lr_out = bv_g16_sum<L>(lr);
    return iters;
}

// ------------------------------------------------------------------ LRT
// BaseType::lrt + _f over ACGT (src/basetype.cpp:105-199), streaming form of bv_lrt: the subsets of a level are
// visited in Combinations order and the first minimum is kept as they come.  Requires q0_mask == 0.
// SPEC: the candidate bases are `nspec` entries of `specific_packed` (3 bits each, reference order, 4 = not ACGT) as in
// bv_lrt -- the pop-group calls of pass 2, lrt([REF] + alts); otherwise A, C, G, T.
#define BV_TIE_TOL 1e-7  /* the sums' rounding is ~1e-13 relative; a true gap this small has never been seen */
template <bool SPEC = false, int NS = BV_G16_SLOTS, bool TWO = false, int L = 16>
__device__ inline void bv_lrt_g16(const BvG16Bins &B, const uint32_t depth[4], uint32_t total, int ref_code, double min_af,
                                  BvLrtOut &o, int specific_packed = 0, int nspec = 0) {
    o.n_alt = 0; o.alt_packed = 0; o.af[0] = o.af[1] = o.af[2] = o.af[3] = 0.;
    o.m = 0; o.first = 0; o.chi2 = 0.; o.em_iters = 0; o.n_em = 0; o.zero_freq = false; o.tie_risk = false;
    int act = 0, m = 0;
    if (SPEC) {
        for (int k = 0; k < nspec; ++k) {
            const int b = (specific_packed >> (3 * k)) & 7;
            if (b < 4 && (double)bv_sel4u(depth, b) / (int)total >= min_af) {  // basetype.cpp:137
                act |= b << (2 * m);
                ++m;
            }
        }
    } else {
        for (int b = 0; b < 4; ++b) {
            if ((double)bv_sel4u(depth, b) / (int)total >= min_af) {  // basetype.cpp:137
                act |= b << (2 * m);
                ++m;
            }
        }
    }
    if (m == 0) return;
    if (m == 1) {  // one active base, no phred-0 call: see bv_lrt
        const int b = act & 3;
        o.m = 1; o.first = b; o.em_iters = 1; o.n_em = 1;
        if (b != ref_code) { o.n_alt = 1; o.alt_packed = b; o.af[0] = 1.0; }
        return;
    }
    const double n_cov = (double)total;
    const int m0 = m;
    // (the start frequencies depth / total are formed where a subset starts, basetype.cpp:99: four divisions per EM run
    // instead of eight registers held through all of them -- the kernel sits at its register limit)
    double fr0 = 0., fr1 = 0., fr2 = 0., fr3 = 0.;  // active_bases_freq
    double lr_alt = 0., chi = 0.;
    for (int n = m0; n > 0; --n) {
        const bool top = (n == m0);
        const int ncomb = top ? 1 : m;
        // closed-form log-likelihoods of the four single-base subsets (see bv_lrt): one sweep, at the level that needs them
        double single0 = 0., single1 = 0., single2 = 0., single3 = 0.;
        if (!top && m == 2) {
            double a_ = 0., g0 = 0., g1 = 0., g2 = 0., g3 = 0.;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                uint32_t b, q;
                double c;
                if (bv_g16_bin(B, s, b, q, c)) {
                    // (q as a value of THIS level: the table addresses are invariant over the levels, get hoisted out of their
                    // loop and, at the register limit, parked in scratch memory -- a memory trip per slot to save a shift and an add)
                    asm volatile("" : "+v"(q));
                    const double lm = B.logmiss[q], dh = B.loghit[q] - lm;
                    a_ += c * lm;
                    g0 += (b == 0) ? c * dh : 0.;
                    g1 += (b == 1) ? c * dh : 0.;
                    g2 += (b == 2) ? c * dh : 0.;
                    g3 += (b == 3) ? c * dh : 0.;
                }
            }
            a_ = bv_g16_sum<L>(a_);
            single0 = a_ + bv_g16_sum<L>(g0); single1 = a_ + bv_g16_sum<L>(g1);
            single2 = a_ + bv_g16_sum<L>(g2); single3 = a_ + bv_g16_sum<L>(g3);
        }
        double best_v = 0., best_lr = 0., bf0 = 0., bf1 = 0., bf2 = 0., bf3 = 0.;
        int best_c = 0;
        for (int c = 0; c < ncomb; ++c) {
            const int drop = top ? -1 : m - 1 - c;
            unsigned in_set = 0;
            for (int k = 0; k < m; ++k)
                if (k != drop) in_set |= 1u << ((act >> (2 * k)) & 3);
            double f[4];
            f[0] = (in_set & 1u) ? (double)depth[0] / n_cov : 0.; f[1] = (in_set & 2u) ? (double)depth[1] / n_cov : 0.;
            f[2] = (in_set & 4u) ? (double)depth[2] / n_cov : 0.; f[3] = (in_set & 8u) ? (double)depth[3] / n_cov : 0.;
            double s = 0.;
            s += f[0]; s += f[1]; s += f[2]; s += f[3];
            double lr;
            int it;
            if (s == 0.) {
                // basetype.cpp:113-115: the reference throws; flagged, and the subset scored like the wave solver's generic
                // path would is not reproducible here -- kernel A sends such sites (min_af == 0 only) to the wave solver
                o.zero_freq = true;
                lr = 0.; it = 0;
            } else if ((in_set & (in_set - 1u)) == 0u) {
                const int b1 = __builtin_ctz(in_set);
                lr = bv_sel4(single0, single1, single2, single3, b1);
                it = (bv_sel4(f[0], f[1], f[2], f[3], b1) < 0.36787944117144233) ? 2 : 1;
                f[0] = (b1 == 0) ? 1.0 : 0.; f[1] = (b1 == 1) ? 1.0 : 0.;
                f[2] = (b1 == 2) ? 1.0 : 0.; f[3] = (b1 == 3) ? 1.0 : 0.;
            } else {
                it = bv_em_g16<NS, TWO, L>(B, f, in_set, n_cov, &lr);
            }
            o.em_iters += it;
            o.n_em += 1;
            const double v = 2 * (lr_alt - lr);
            // Which of two subsets that score (nearly) alike wins is decided by the rounding of the reference's per-sample
            // sums, which sums over bins do not reproduce: the caller replays such an item in sample order if it can
            // (pop-groups of at most BV_ORD_MAX covered samples, bv_p2g_solve16_kernel).  Comparing with the RUNNING minimum is
            // enough to catch every score within the tolerance of the level's final minimum: if the minimum comes later it is
            // compared with a running value that lies between the two; if it came earlier it is the running value.
            if (SPEC && !top && c > 0 && fabs(v - best_v) <= BV_TIE_TOL * (1.0 + fabs(best_v))) o.tie_risk = true;
            if (top || c == 0 || v < best_v) {  // first minimum, algorithm.h:24-27
                best_v = v; best_c = c; best_lr = lr;
                bf0 = f[0]; bf1 = f[1]; bf2 = f[2]; bf3 = f[3];
            }
        }
        lr_alt = best_lr;
        bool accept = top;
        if (!top) {
            chi = best_v;
            accept = chi < 24;  // LRT_THRESHOLD, basetype.h:21
            // (a chi-square within rounding of the threshold: whether the allele is kept hangs on the reference's sums too)
            if (SPEC && fabs(chi - 24.0) <= BV_TIE_TOL * 25.0) o.tie_risk = true;
            if (accept) {
                const int drop = m - 1 - best_c;
                const int low = act & ((1 << (2 * drop)) - 1), high = act >> (2 * drop + 2);
                act = low | (high << (2 * drop));
                m = n;
            }
        }
        if (!accept) break;
        fr0 = bf0; fr1 = bf1; fr2 = bf2; fr3 = bf3;
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

// ------------------------------------------------------------------ Wilcoxon rank sum, 16 values per window
__device__ inline unsigned long long bv_ranksum_window_g16(uint32_t ref_v, uint32_t alt_v, unsigned long long n,
                                                           unsigned long long &below) {
    const uint32_t t = ref_v + alt_v;
    const uint32_t incl = bv_g16_incl_scan_u32(t);
    const unsigned long long below_v = below + (incl - t);
    const unsigned long long term = (unsigned long long)ref_v * (2ull * n - 2ull * below_v - t + 1ull);
    const unsigned long long s = bv_g16_sum_u64(term);
    below += (unsigned long long)bv_g16_sum_u32(t);
    return s;
}

// ------------------------------------------------------------------ one site on one group, in two phases
// The solve of a candidate is split over two kernels so that each kernel's code fits the instruction cache (64 KB per pair
// of CUs; as one kernel the solve was 207 KB and 1.2 % of its instruction fetches missed):
//   phase 1 (bv_site_lrt_g16)   BaseType::lrt -- active set, EMs, LRT, alt set and AF -- and the record's LRT fields;
//   phase 2 (bv_site_tail_g16)  QUAL / QD / CAF, the base-quality rank sum and the two strand-bias tests, patched into the record.
// Between them the record itself carries the state; two facts of the LRT that are no record fields ride in spare status bits.
#define BV_G16_STASH_M_SHIFT 16      /* bits 16-18: final active-set size (BvLrtOut::m)      */
#define BV_G16_STASH_FIRST_SHIFT 20  /* bits 20-21: active_bases[0]        (BvLrtOut::first)  */
#define BV_G16_STASH_MASK 0x00370000u

// what phase 2 needs of phase 1, when both run in one kernel (else phase 2 reads it back from the record)
struct BvG16Lrt {
    uint32_t status;    // with the stash bits
    uint32_t aw0, aw1;  // the record's bytes n_alt, alt[0..3] (+ n_em, em_iters): as stored at bv_site_result::n_alt
    double chi2;
    int ref;            // the reference base's code (0-3; 4: not ACGT)
};

// Phase 1.  `scratch`: the group's BV_G16_GRP_WORDS words of LDS -- the EM's previous marginals (B.pm points into it), then
// the staging of the record.  Returns whether the site is a variant site.
__device__ inline bool bv_site_lrt_g16(const BvSolveArgs &a, uint32_t site, const uint32_t depth[4], uint32_t total, uint32_t badq,
                                       const BvG16Bins &B, uint32_t *scratch, int lane, BvG16Lrt *pre = nullptr) {
    const int gl = lane & 15;
    constexpr int REC_WORDS = (int)(sizeof(bv_site_result) / 4);
    bv_site_result *res = reinterpret_cast<bv_site_result *>(scratch);
    int ref = a.ref_base[site];
    if (ref > 4) ref = 4;
    const double qnan = __builtin_nan("");
    uint32_t flags = BV_SITE_COVERED | (badq ? BV_SITE_BAD_QUAL : 0u);
    BvLrtOut L;
    L.n_alt = 0; L.alt_packed = 0; L.af[0] = L.af[1] = L.af[2] = L.af[3] = 0.;
    L.m = 0; L.first = 0; L.chi2 = 0.; L.em_iters = 0; L.n_em = 0; L.zero_freq = false;
    if (!(a.flags & BV_FLAG_SKIP_LRT)) bv_lrt_g16(B, depth, total, ref, a.min_af, L);
    if (L.zero_freq) flags |= BV_SITE_ZERO_FREQ;
    if (L.n_alt > 0) flags |= BV_SITE_VARIANT;
    flags |= ((uint32_t)L.m << BV_G16_STASH_M_SHIFT) | ((uint32_t)L.first << BV_G16_STASH_FIRST_SHIFT);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the EM is done with the scratch
    for (int i = gl; i < REC_WORDS; i += 16) scratch[i] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (gl == 0) {
#pragma unroll
        for (int b = 0; b < 4; ++b) res->depth[b] = depth[b];
        res->total_depth = total;
        // (both phases in one kernel: the stash travels in `pre`, and the record's status word is final but for the bits that
        // phase 2 and the rank-sum kernels OR into it)
        res->status = pre != nullptr ? (flags & ~BV_G16_STASH_MASK) : flags;
        res->n_alt = (uint8_t)L.n_alt;
#pragma unroll
        for (int k = 0; k < BV_MAX_ALT; ++k) {
            if (k < L.n_alt) {
                res->alt[k] = (uint8_t)bv_alt_at(L, k);
                res->af[k] = L.af[k];
            }
        }
        res->chi2 = L.chi2;
        res->em_iters = (uint16_t)L.em_iters;
        res->n_em = (uint8_t)L.n_em;
        res->mq_ranksum = qnan;
        res->rpr_ranksum = qnan;
        res->bq_ranksum = qnan;
        if (L.zero_freq) atomicAdd(&a.counters[BV_CTR_ZEROFREQ], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int i = gl; i < REC_WORDS; i += 16) reinterpret_cast<uint32_t *>(&a.out[site])[i] = scratch[i];
    if (pre != nullptr) {
        pre->status = flags;
        pre->aw0 = scratch[offsetof(bv_site_result, n_alt) / 4];
        pre->aw1 = scratch[offsetof(bv_site_result, n_alt) / 4 + 1];
        pre->chi2 = L.chi2;
        pre->ref = ref;
    }
    return L.n_alt > 0;
}

// Phase 2.  `S`: the site's strand totals; `bins` / `nb`: its exported bins in device memory (bv_g16_bin layout), or, when both
// phases run in one kernel, `w_held`: the lane's bin words as phase 1 held them (B.w: no second trip to memory); `cls`: the
// group's scratch, here 2 x 128 REF / ALT counts per phred for the rank sum.  Everything it computes is patched into the
// record phase 1 wrote, by the group's first lane.
__device__ inline void bv_site_tail_g16(const BvSolveArgs &a, uint32_t site, const BvSiteSums &S, const uint32_t *bins, uint32_t nb,
                                        uint32_t *cls, int lane, const BvG16Lrt *pre = nullptr, const uint32_t *w_held = nullptr) {
    const int gl = lane & 15;
    bv_site_result *rec = &a.out[site];
    uint32_t depth[4], total = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        depth[b] = S.fwd[b] + S.rev[b];
        total += depth[b];
    }
    int ref = pre ? pre->ref : (int)a.ref_base[site];
    if (ref > 4) ref = 4;
    // what phase 1 left: status (with the stash), n_alt + alt[4] (8 bytes), chi2
    const uint32_t st = pre ? pre->status : rec->status;
    const uint2 aw = pre ? make_uint2(pre->aw0, pre->aw1) : *reinterpret_cast<const uint2 *>(&rec->n_alt);
    const double chi2 = pre ? pre->chi2 : rec->chi2;
    const int n_alt = (int)(aw.x & 0xFFu);
    const int m = (int)((st >> BV_G16_STASH_M_SHIFT) & 7u), first = (int)((st >> BV_G16_STASH_FIRST_SHIFT) & 3u);
    uint32_t flags = st & ~BV_G16_STASH_MASK;
    const double qnan = __builtin_nan("");
    uint32_t c_rf = 0, c_rr = 0, c_af = 0, c_ar = 0;  // caller.cpp:1236-1245
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (b == ref) { c_rf += S.fwd[b]; c_rr += S.rev[b]; } else { c_af += S.fwd[b]; c_ar += S.rev[b]; }
    }
    double bq_ranksum = qnan, qual = 0., qd = 0.;
    uint32_t v_rf = 0, v_rr = 0, v_af = 0, v_ar = 0;
    const bool have_var = n_alt > 0;
    double caf0 = 0., caf1 = 0., caf2 = 0., caf3 = 0.;
    if (have_var) {
        uint32_t alt_mask = 0, ad_sum_u = 0;
#pragma unroll
        for (int k = 0; k < BV_MAX_ALT; ++k) {
            if (k < n_alt) {
                const int b = (int)((k < 3 ? (aw.x >> (8 * (k + 1))) : aw.y) & 3u);
                alt_mask |= 1u << b;
                ad_sum_u += bv_sel4u(depth, b);
            }
        }
        // REF / ALT counts per phred value for the base-quality rank sum, scattered from the bins -- first, so that bin words
        // held in registers (w_held) are done with before QUAL and the strand-bias tests want theirs
        {
            for (int i = gl; i < 2 * 128; i += 16) cls[i] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int s = 0; s < BV_G16_SLOTS; ++s) {
                const uint32_t i = (uint32_t)(s * 16 + gl);
                const uint32_t ws = w_held ? w_held[s] : (i < nb ? bins[i] : 0u);
                if (ws & 0xFFFFu) {
                    const uint32_t q = (ws >> 16) & 127u, b = ws >> 23;
                    // bins are unique per (base, phred); several ALT bases can share a phred: add, one lane at a time per word
                    if ((int)b == ref) cls[q] = ws & 0xFFFFu;
                    else if ((alt_mask >> b) & 1u) atomicAdd(&cls[128 + q], ws & 0xFFFFu);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        // QUAL / QD / CAF (basetype.cpp:180-196, caller.cpp:1113-1122, 1160-1161)
        {
            const double r = (double)bv_sel4u(depth, first) / (double)total;
            if (m == 1 && total > 10 && r > 0.5) qual = 5000.0;
            else qual = bv_qual_from_chi2(chi2);
            double ad_sum = 0;
#pragma unroll
            for (int k = 0; k < BV_MAX_ALT; ++k) {
                if (k < n_alt) {
                    const uint32_t d = bv_sel4u(depth, (int)((k < 3 ? (aw.x >> (8 * (k + 1))) : aw.y) & 3u));
                    ad_sum = ad_sum + (double)d;
                    const double cf = (double)d / (int)total;
                    if (k == 0) caf0 = cf; else if (k == 1) caf1 = cf; else if (k == 2) caf2 = cf; else caf3 = cf;
                }
            }
            qd = qual / ad_sum;
            if (qd == 0) qd = 0.0;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {  // caller.cpp:1164
            if (b == ref) { v_rf += S.fwd[b]; v_rr += S.rev[b]; }
            else if ((alt_mask >> b) & 1u) { v_af += S.fwd[b]; v_ar += S.rev[b]; }
        }
        // base-quality rank sum (caller.cpp:1157), from the REF / ALT counts per phred value scattered above
        {
            const unsigned long long n1 = (ref < 4) ? bv_sel4u(depth, ref) : 0ull, n2 = ad_sum_u;
            unsigned long long below = 0, twoR = 0;
#pragma unroll
            for (int wi = 0; wi < 8; ++wi) {
                // (a window without a count adds nothing to the sum nor to `below`: skipped per group)
                const uint32_t rv = cls[wi * 16 + gl], av = cls[128 + wi * 16 + gl];
                if (bv_g16_ballot((rv | av) != 0u, lane) != 0u) twoR += bv_ranksum_window_g16(rv, av, n1 + n2, below);
            }
            bq_ranksum = bv_ranksum_phred(twoR, n1, n2);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // before the next site's zeroing
        }
    }
    // strand bias: FS / SOR of the CVG table, then of the VCF table unless it is the same 2 x 2 table.  ONE call site for the
    // test (its code is the bulk of this kernel)
    double c_fs = 0, c_sor = 0, v_fs = 0, v_sor = 0;
    if (!(a.flags & BV_FLAG_SKIP_FISHER)) {
        const bool same = have_var && v_rf == c_rf && v_rr == c_rr && v_af == c_af && v_ar == c_ar;
        const int ntab = (have_var && !same) ? 2 : 1;
#pragma unroll 1
        for (int t = 0; t < ntab; ++t) {
            double fs, sor;
            bv_strand_bias_g16(t ? v_rf : c_rf, t ? v_rr : c_rr, t ? v_af : c_af, t ? v_ar : c_ar, lane, a.lnfact, &fs, &sor, &flags);
            if (t) { v_fs = fs; v_sor = sor; } else { c_fs = fs; c_sor = sor; }
        }
        if (same) { v_fs = c_fs; v_sor = c_sor; }
    }
    if (gl == 0) {
        // (with `pre` the rank sums of the site may be on their way into the record already: their BV_SITE_RANKSUM must survive)
        if (pre == nullptr) rec->status = flags;
        else if (flags & ~st & BV_SITE_SOR_OVERFLOW) atomicOr(&rec->status, BV_SITE_SOR_OVERFLOW);
        if (!(a.flags & BV_FLAG_SKIP_FISHER)) {
            *reinterpret_cast<uint2 *>(&rec->cvg_sb[0]) = make_uint2(c_rf, c_rr);
            *reinterpret_cast<uint2 *>(&rec->cvg_sb[2]) = make_uint2(c_af, c_ar);
            rec->cvg_fs = c_fs; rec->cvg_sor = c_sor;
            if (have_var) {
                *reinterpret_cast<uint2 *>(&rec->var_sb[0]) = make_uint2(v_rf, v_rr);
                *reinterpret_cast<uint2 *>(&rec->var_sb[2]) = make_uint2(v_af, v_ar);
                rec->var_fs = v_fs; rec->var_sor = v_sor;
            }
        }
        if (n_alt > 0) { rec->caf[0] = caf0; rec->qual = qual; rec->qd = qd; rec->bq_ranksum = bq_ranksum; }
        if (n_alt > 1) rec->caf[1] = caf1;
        if (n_alt > 2) rec->caf[2] = caf2;
        if (n_alt > 3) rec->caf[3] = caf3;
    }
}
