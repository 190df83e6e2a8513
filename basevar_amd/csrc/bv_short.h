// bv_short.h -- pieces shared by the short-row pass-1 kernels (bv_pass1_short.hip: streaming kernel + solve kernel;
// bv_pass1_fused.hip: both in one persistent kernel): the histogram geometry, the per-lane Fisher test and the one-lane
// finish of a non-candidate site, the wave solver's scratch.
#pragma once

#include "bv_kernels.h"
#include "bv_solver.h"
#include "bv_solver16.h"
#include "bv_tally.h"

#define BV_S_HROWQ 128                       /* phred axis of the short-row histogram */
#define BV_S_HWORDS (BV_ROWS * BV_S_HROWQ)   /* 1024 words = 4 KiB */
#define BV_S_OVF 8                           /* + per-row counts of covered cells with phred >= 128 (invalid input) */
#define BV_S_SLOT_WORDS 512                  /* one slot of U = 1: 1 KiB of calls, then 1 KiB of phreds (U x that for U chunks per lane) */
#define BV_S_SIMPLE_MAX_TABLES 32            /* a non-candidate site's strand table has at most this many Fisher tables */


// ------------------------------------------------------------------------------ per-lane Fisher test
// kt_fisher_exact (two-sided), htslib/kfunc.c:245-313, one table family per LANE: the walk of kfunc.c:291-307 with its
// incremental hypergeo_acc (kfunc.c:220-243: multiplicative update, re-seeded from log-factorials whenever n11 % 11 == 0
// or the table's n22 is 0).  log(k!) comes from the engine's table of the host's lgamma (the reference's own values).
struct BvHgAcc {
    int n11, n1_, n_1, n;
    double p;
};
__device__ __forceinline__ double bv_lbinom_lane(const BvLnTab &T, int n, int k) {
    if (k == 0 || n == k) return 0;
    return bv_lnfact(T, n) - bv_lnfact(T, k) - bv_lnfact(T, n - k);
}
__device__ __forceinline__ double bv_hypergeo_lane(const BvLnTab &T, int n11, int n1_, int n_1, int n) {
    return exp(bv_lbinom_lane(T, n1_, n11) + bv_lbinom_lane(T, n - n1_, n_1 - n11) - bv_lbinom_lane(T, n, n_1));
}
__device__ __forceinline__ double bv_hgacc_step(const BvLnTab &T, int n11, BvHgAcc &x) {  // hypergeo_acc(n11, 0, 0, 0, aux)
    if (n11 % 11 && n11 + x.n - x.n1_ - x.n_1) {
        if (n11 == x.n11 + 1) {
            x.p *= (double)(x.n1_ - x.n11) / n11 * (x.n_1 - x.n11) / (n11 + x.n - x.n1_ - x.n_1);
            x.n11 = n11;
            return x.p;
        }
        if (n11 == x.n11 - 1) {
            x.p *= (double)x.n11 / (x.n1_ - n11) * (x.n11 + x.n - x.n1_ - x.n_1) / (x.n_1 - n11);
            x.n11 = n11;
            return x.p;
        }
    }
    x.n11 = n11;
    x.p = bv_hypergeo_lane(T, x.n11, x.n1_, x.n_1, x.n);
    return x.p;
}
__device__ inline double bv_fisher_two_sided_lane(int n11, int n12, int n21, int n22, const BvLnTab &T) {
    const int n1_ = n11 + n12, n_1 = n11 + n21, n = n11 + n12 + n21 + n22;
    const int max = (n_1 < n1_) ? n_1 : n1_;
    int min = n1_ + n_1 - n;
    if (min < 0) min = 0;
    if (min == max) return 1.;
    BvHgAcc x;
    x.n11 = n11; x.n1_ = n1_; x.n_1 = n_1; x.n = n;
    x.p = bv_hypergeo_lane(T, n11, n1_, n_1, n);
    const double q = x.p;
    if (q == 0.0) return 0.0;  // kfunc.c:260-289: two = 0
    double p, left, right;
    int i, j;
    p = bv_hgacc_step(T, min, x);
    for (left = 0., i = min + 1; p < 0.99999999 * q && i <= max; ++i) { left += p; p = bv_hgacc_step(T, i, x); }
    if (p < 1.00000001 * q) left += p;
    p = bv_hgacc_step(T, max, x);
    for (right = 0., j = max - 1; p < 0.99999999 * q && j >= 0; --j) { right += p; p = bv_hgacc_step(T, j, x); }
    if (p < 1.00000001 * q) right += p;
    double two = left + right;
    if (two > 1.) two = 1.;
    return two;
}

// ------------------------------------------------------------------------------ solve kernels
// A non-candidate site (hom-ref or uncovered), finished by ONE LANE: depths, flags, one small Fisher test.
// Three 16-byte loads served by the L2 (sc1: the CU's vector L1 is bypassed), waited for inside the statement.  For data that
// another wave of the SAME launch has stored (the fused kernel's hand-offs): summaries are 48 bytes, neighbours share a
// 128-byte line, and a line fetched for one site would otherwise be served stale from the L1 for the next.
__device__ __forceinline__ void bv_load3_l2(const void *p, uint4 &r0, uint4 &r1, uint4 &r2) {
    bv_u32x4 a, b, c;
    asm volatile(
        "global_load_dwordx4 %0, %3, off sc1\n\t"
        "global_load_dwordx4 %1, %3, off offset:16 sc1\n\t"
        "global_load_dwordx4 %2, %3, off offset:32 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c)
        : "v"(p)
        : "memory");
    r0 = make_uint4(a.x, a.y, a.z, a.w); r1 = make_uint4(b.x, b.y, b.z, b.w); r2 = make_uint4(c.x, c.y, c.z, c.w);
}
// L2: the summaries were stored by other waves of this launch (bv_pass1_fused.hip)
template <bool L2 = false>
__device__ __forceinline__ void bv_p1s_simple_site(const BvP1ShortArgs &a, const BvLnTab &lnfact, uint32_t site) {
    const double qnan = __builtin_nan("");
    uint4 s0, s1, s2;
    if (L2) bv_load3_l2(&a.summ[site], s0, s1, s2);
    else {
        const uint4 *sp = reinterpret_cast<const uint4 *>(&a.summ[site]);
        s0 = sp[0]; s1 = sp[1]; s2 = sp[2];
    }
    if (s2.y & BV_SUM_CAND) return;
    const uint32_t fwd[4] = {s0.x, s0.y, s0.z, s0.w}, rev[4] = {s1.x, s1.y, s1.z, s1.w};
    bv_site_result r;
    {
        uint32_t *w = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(r) / 4); ++i) w[i] = 0u;
    }
    uint32_t total = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) { r.depth[b] = fwd[b] + rev[b]; total += r.depth[b]; }
    r.total_depth = total;
    if (a.flags & BV_FLAG_TALLY_ONLY) {
        // diagnostic: depths only
    } else if (total == 0) {
        r.mq_ranksum = r.rpr_ranksum = r.bq_ranksum = qnan;  // caller.cpp:718 / basetype.cpp:132
    } else {
        int ref = a.ref_base[site];
        if (ref > 4) ref = 4;
        uint32_t flags = BV_SITE_COVERED | ((s2.y & BV_SUM_BADQ) ? BV_SITE_BAD_QUAL : 0u);
        uint32_t c_rf = 0, c_rr = 0, c_af = 0, c_ar = 0;  // caller.cpp:1236-1245
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b == ref) { c_rf += fwd[b]; c_rr += rev[b]; } else { c_af += fwd[b]; c_ar += rev[b]; }
        }
        if (!(a.flags & BV_FLAG_SKIP_FISHER)) {
            // strand_bias tail, src/basetype.cpp:277-286 (see bv_strand_bias_wave for the SOR overflow note)
            double fs = -10 * log10(bv_fisher_two_sided_lane((int)c_rf, (int)c_rr, (int)c_af, (int)c_ar, lnfact));
            if (isinf(fs)) fs = 10000;
            else if (fs == 0) fs = 0.0;
            const int den = (int)(c_rr * c_af), num = (int)(c_rf * c_ar);
            if ((unsigned long long)c_rr * c_af > 0x7fffffffull || (unsigned long long)c_rf * c_ar > 0x7fffffffull)
                flags |= BV_SITE_SOR_OVERFLOW;
            r.cvg_fs = fs;
            r.cvg_sor = (c_rr != 0u && c_af != 0u) ? (double)num / (double)den : 10000;
            r.cvg_sb[0] = c_rf; r.cvg_sb[1] = c_rr; r.cvg_sb[2] = c_af; r.cvg_sb[3] = c_ar;
        }
        r.status = flags;
        // lrt() with one active base, the reference base: no ALT, chi2 0, one EM run of one iteration (bv_lrt)
        const bool lrt_ran = !(a.flags & BV_FLAG_SKIP_LRT);
        r.em_iters = lrt_ran ? 1 : 0;
        r.n_em = lrt_ran ? 1 : 0;
        r.mq_ranksum = r.rpr_ranksum = r.bq_ranksum = qnan;
    }
    uint4 *dst = reinterpret_cast<uint4 *>(&a.out[site]);
    const uint4 *src = reinterpret_cast<const uint4 *>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(r) / 16); ++i) dst[i] = src[i];
}

// Candidates that need the wave solver (shallow sites, phred-0 calls, > 128 bins, min_af <= 0): one wave per site.
#define BV_P1S_RAW_WORDS (2 * BV_SLOTS * BV_WAVE + 4 * 128)  /* the wave solver's per-wave bins: codes [384], counts [384], merged counts [4][128] */
// per wave: the four groups' scratch of the 16-lane solver -- or, while the wave works off the (rare) candidates that need the
// wave solver, that solver's bins and scratch in the same bytes
union __attribute__((aligned(16))) BvP1sWaveScratch {
    uint32_t grp[4][BV_G16_GRP_WORDS];  // per group: bv_site_lrt_g16 / bv_site_tail_g16
    struct {
        uint32_t raw[BV_P1S_RAW_WORDS];  // bin codes [384], bin counts [384], merged (base, phred) counts [4][128]
        BvSolverScratch sc;
    } w;
};
