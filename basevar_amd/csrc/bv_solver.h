// bv_solver.h -- the per-site solver on one wave: everything the reference computes for a site
// from its (strand, base, phred) histogram in LDS.  Shared by the streaming kernels
// (bv_pass1.hip) and the sample-axis tile mode (bv_tiles.hip).
#pragma once
#include <type_traits>

#include "bv_kernels.h"

#define BV_H2_ROWQ 256                         /* phred axis: the raw phred byte indexes the row */
#define BV_H2_WORDS (BV_ROWS * BV_H2_ROWQ)     /* 2048 x u32 = 8 KiB per histogram               */

struct BvSolverScratch {
    BvLrtShared lrt;
    bv_site_result res;  // staged record, stored with one coalesced write
    alignas(8) uint16_t ord[BV_ORD_ALLOC];  // shallow sites: the covered cells in sample order (bv_gather_ordered) + bv_em_ordered's scratch
};
struct BvSolverShared {
    uint32_t bin_code[BV_SLOTS * BV_WAVE];  // compacted non-empty (base<<7 | phred) bins
    uint32_t bin_cnt[BV_SLOTS * BV_WAVE];
    BvSolverScratch sc;
};
// ALIAS mode (short-row kernel, LDS-limited): the compacted bins live in the histogram itself, in
// the upper halves (phred 128..255) of rows 0-2 (codes) and 3-5 (counts).  Valid input never
// touches those words; they are read (bad-phred check, depth) before the bins overwrite them.
#define BV_ALIAS_CODE_OFF (0 * BV_H2_ROWQ + 128)
#define BV_ALIAS_CNT_OFF (3 * BV_H2_ROWQ + 128)


// ------------------------------------------------------------------------------ solver
// strand/base row sums and deterministic compaction of the non-empty (base, phred) bins
template <bool ALIAS>
__device__ __forceinline__ void bv_prologue_wave(const uint32_t *hist, uint32_t *bin_code, uint32_t *bin_cnt, int lane,
                                                 uint32_t fwd[4], uint32_t rev[4], uint32_t *nb_out, uint32_t *badq_out) {
    uint32_t nb = 0;
    bool bad = false;
    uint32_t facc[4], racc[4];
    // phred 128..255 first: only invalid input puts counts there (they still belong to the depth)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        facc[b] = 0; racc[b] = 0;
#pragma unroll
        for (int qr = 2; qr < 4; ++qr) {
            const int q = (qr << 6) | lane;
            uint32_t f = hist[(b << 8) | q], v = hist[((b | 4) << 8) | q];
            facc[b] += f;
            racc[b] += v;
            bad |= (f | v) != 0;
        }
    }
    if (ALIAS) bv_lrt_sync<0>();  // every lane has read the upper halves before bins land in them
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int qr = 0; qr < 2; ++qr) {  // phred 0..127 hold every valid bin
            const int q = (qr << 6) | lane;
            uint32_t f = hist[(b << 8) | q], v = hist[((b | 4) << 8) | q];
            facc[b] += f;
            racc[b] += v;
            uint32_t c = f + v;
            bad |= (c != 0) && (q >= BV_NQ_VALID);
            bool valid = (c != 0) && (q < BV_NQ_VALID);
            unsigned long long m = __ballot(valid);
            uint32_t pos = nb + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (valid) {
                const uint32_t at = ALIAS ? pos + (pos & ~127u) : pos;
                bin_code[at] = ((uint32_t)b << 7) | (uint32_t)q;
                bin_cnt[at] = c;
            }
            nb += (uint32_t)__popcll(m);
        }
    }
    {
        const uint32_t v[8] = {facc[0], facc[1], facc[2], facc[3], racc[0], racc[1], racc[2], racc[3]};
        uint32_t t[8];
        bv_wave_sum8_u32(v, t, lane);
#pragma unroll
        for (int b = 0; b < 4; ++b) { fwd[b] = t[b]; rev[b] = t[4 + b]; }
    }
    *nb_out = nb;
    *badq_out = (__ballot(bad) != 0ull) ? 1u : 0u;
}

// what the solver needs from the launch arguments (passed by value, in registers)
struct BvSolveArgs {
    const uint8_t *ref_base;
    bv_site_result *out;
    uint32_t *var_list;
    uint32_t *counters;
    double min_af;
    uint32_t flags;
    BvLnTab lnfact;  // log-factorial table (BvTables::lnfact)
    const double *loghit, *logmiss;  // BvTables::loghit / logmiss (device memory)
    // the planes the site's row lives in, for the ordered replay of shallow sites (bs == NULL: not available, e.g. the
    // per-site-tally realisation of the tile mode)
    const uint8_t *bs, *q;
    uint64_t pitch;
    uint32_t n_samples;
};

#define BV_LDS __attribute__((address_space(3)))

// Merged-strand (base, phred) counts as the base-quality rank sum wants them: from the pass-1 histogram
// (rows of 256 phred words, forward rows 0-3 and reverse rows 4-7), or from a [4][128] table (bv_pass1_short.hip).
struct BvHqFromHist {
    const uint32_t *hist;
    __device__ __forceinline__ uint32_t operator()(int b, int q) const { return hist[(b << 8) | q] + hist[((b | 4) << 8) | q]; }
};
struct BvHqMerged {
    const uint32_t *hq;
    __device__ __forceinline__ uint32_t operator()(int b, int q) const { return hq[(b << 7) | q]; }
};

// What the tally of a site boils down to before the solve: strand x base totals, the number of compacted
// (base, phred) bins, and two facts about the phred values seen.
struct BvSiteSums {
    uint32_t fwd[4], rev[4];
    uint32_t nb;       // compacted bins
    uint32_t badq;     // a covered cell had phred > 93
    uint32_t q0_mask;  // bit b: base b has a phred-0 call
};

// WHO runs a site's EM runs and its two Fisher tests.  BvSoloWork: the calling wave itself (every kernel; the arithmetic of a
// site is defined by this form).  bv_pass1.hip adds a team form for the last sites of a long-row launch, where the same
// one-wave runs are dealt to the workgroup's idle waves -- same values, shorter critical path.
struct BvSoloWork {
    __device__ __forceinline__ void lrt(const BvBins &B, const uint32_t depth[4], uint32_t total, int nspec, int ref,
                                        double min_af, BvLrtShared *sh, int lane, BvLrtOut &L, uint32_t q0_mask) const {
        bv_lrt<0>(B, depth, total, /*A,C,G,T*/ 0 | (1 << 3) | (2 << 6) | (3 << 9), nspec, ref, min_af, sh, 0, lane, L, q0_mask);
    }
    // the VCF table is known (ntab == 2: it differs from the CVG table and needs its own test)
    __device__ __forceinline__ void tables_known(const uint32_t[4], int, int) const {}
    // FS / SOR of the CVG table c[] and, if ntab == 2, of the VCF table v[] ({ref_fwd, ref_rev, alt_fwd, alt_rev})
    __device__ __forceinline__ void strand_bias(const uint32_t c[4], const uint32_t v[4], int ntab, int lane, const BvLnTab &T,
                                                double &c_fs, double &c_sor, double &v_fs, double &v_sor,
                                                uint32_t &flags) const {
#pragma unroll 1
        for (int t = 0; t < ntab; ++t) {
            double fs, sor;
            bv_strand_bias_wave(t ? v[0] : c[0], t ? v[1] : c[1], t ? v[2] : c[2], t ? v[3] : c[3], lane, T, &fs, &sor, &flags);
            if (t) { v_fs = fs; v_sor = sor; } else { c_fs = fs; c_sor = sor; }
        }
    }
};

// A WORK policy may bring its own source of a shallow site's covered cells in sample order (`uint32_t ordered(site, ord, lane)`:
// bv_tiles.hip, which has no rows left when it solves); every other policy takes them from the site's row (bv_gather_ordered).
template <class W, class = void>
struct bv_work_has_ordered : std::false_type {};
template <class W>
struct bv_work_has_ordered<W, std::void_t<decltype(&W::ordered)>> : std::true_type {};

// Everything the reference computes for one site, on one wave, from the site's totals, its compacted bins
// (bin_code / bin_cnt, layout given by ALIAS) and the merged (base, phred) counts `hq`.
// Register-pressure note: this body sits inside the persistent loop of the kernel.  With
// MachineLICM enabled, every libm polynomial constant of log/exp is hoisted out of that loop
// and kept live around it -> 240 VGPRs, 2 waves/SIMD.  The kernel files are therefore compiled with
// `-mllvm -disable-machine-licm` (see Makefile): 121 VGPRs, 4 waves/SIMD.  (A noinline call
// is no way out: device functions are register-allocated without an occupancy target.)
// DEFER_LIST: the caller appends variant sites to var_list itself, in batches (one returning atomic per SITE on the single
// counter address serialises at ~88 M/s: 0.23 ms per 20 k variant sites, measured in the short-row solve kernel).
// Returns whether the site is a variant site.
template <bool ALIAS, class HQ, bool DEFER_LIST = false, class WORK = BvSoloWork>
__device__ __forceinline__ bool bv_site_solve(const BvSolveArgs &a, uint32_t site, const BvSiteSums &S, uint32_t *bin_code,
                                              uint32_t *bin_cnt, const HQ &hq, BvSolverScratch *sv, const double *tab_hit,
                                              const double *tab_miss, int lane, const WORK &work = WORK()) {
    constexpr int REC_WORDS = (int)(sizeof(bv_site_result) / 4);
    uint32_t *res_words = reinterpret_cast<uint32_t *>(&sv->res);
    uint32_t depth[4], total = 0;
    const uint32_t *fwd = S.fwd, *rev = S.rev;
    const uint32_t nb = S.nb, badq = S.badq, q0_mask = S.q0_mask;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        depth[b] = fwd[b] + rev[b];
        total += depth[b];
    }
    int ref = a.ref_base[site];
    if (ref > 4) ref = 4;
    const double qnan = __builtin_nan("");
    bool is_variant = false;

    if (a.flags & BV_FLAG_TALLY_ONLY) {  // diagnostic: streaming part only
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) sv->res.depth[b] = depth[b];
            sv->res.total_depth = total;
        }
    } else if (total == 0) {
        // nothing to call (caller.cpp:718 / basetype.cpp:132): the record stays zero
        if (lane == 0) sv->res.mq_ranksum = sv->res.rpr_ranksum = sv->res.bq_ranksum = qnan;
    } else {
        uint32_t flags = BV_SITE_COVERED | (badq ? BV_SITE_BAD_QUAL : 0u);

        // ---- CVG strand-bias table: alt = every non-ref ACGT base (caller.cpp:1236-1245).  The two Fisher tests of a
        // site (this one and the VCF one, caller.cpp:1164) run further down through ONE call site: the test is
        // ~45 KB of code, and a second inlined copy was a third of the kernel (162 KB against a 64 KB I-cache).
        uint32_t c_rf = 0, c_rr = 0, c_af = 0, c_ar = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b == ref) { c_rf += fwd[b]; c_rr += rev[b]; } else { c_af += fwd[b]; c_ar += rev[b]; }
        }

        // ---- lrt() over ACGT (basetype.h:115)
        BvBins B;
        B.code = bin_code; B.cnt = bin_cnt; B.skip_mask = ALIAS ? ~127u : 0u; B.hit = tab_hit; B.miss = tab_miss;
        B.loghit = a.loghit; B.logmiss = a.logmiss;
        B.nb = (int)nb;
        B.ord = nullptr; B.n_ord = 0;
        bool have_src = a.bs != nullptr;
        if constexpr (bv_work_has_ordered<WORK>::value) have_src = true;
        if (have_src && total >= 2u && total <= (uint32_t)BV_ORD_MAX && !(a.flags & BV_FLAG_SKIP_LRT)) {
            // shallow site: the reference's per-sample order decides ties -- replay it (bv_em_ordered)
            uint32_t got;
            if constexpr (bv_work_has_ordered<WORK>::value) got = work.ordered(site, sv->ord, lane);
            else got = bv_gather_ordered(a.bs + (size_t)site * a.pitch, a.q + (size_t)site * a.pitch, a.n_samples, sv->ord, lane);
            bv_lrt_sync<0>();
            if (got == total) { B.ord = sv->ord; B.n_ord = (int)total; }
            if (B.ord != nullptr && bv_hostlog_of(a.logmiss) == nullptr) flags |= BV_SITE_LOG_APPROX;  // loud: see basevar_amd.h
        }
        BvLrtOut L;
        // q0_mask: bases that hold a phred-0 call (1 - eps == 0) keep the generic EM path, because
        // the reference's 0/0 there yields NaN frequencies that must be reproduced
        work.lrt(B, depth, total, (a.flags & BV_FLAG_SKIP_LRT) ? 0 : 4, ref, a.min_af, &sv->lrt, lane, L, q0_mask);
        if (L.zero_freq) flags |= BV_SITE_ZERO_FREQ;

        double bq_ranksum = qnan;
        uint32_t v_rf = 0, v_rr = 0, v_af = 0, v_ar = 0;
        bool have_var = false;
        if (L.n_alt > 0) {
            flags |= BV_SITE_VARIANT;
            uint32_t alt_mask = 0, ad_sum_u = 0;
#pragma unroll
            for (int k = 0; k < BV_MAX_ALT; ++k) {
                if (k < L.n_alt) {
                    alt_mask |= 1u << bv_alt_at(L, k);
                    ad_sum_u += bv_sel4u(depth, bv_alt_at(L, k));
                }
            }
            // QUAL / QD / AF / CAF (basetype.cpp:180-196, caller.cpp:1113-1122, 1160-1161)
            {
                double r = (double)bv_sel4u(depth, L.first) / (double)total;
                double qual;
                if (L.m == 1 && total > 10 && r > 0.5) qual = 5000.0;
                else qual = bv_qual_from_chi2(L.chi2);
                double ad_sum = 0;
#pragma unroll
                for (int k = 0; k < BV_MAX_ALT; ++k) {
                    if (k < L.n_alt) {
                        const uint32_t d = bv_sel4u(depth, bv_alt_at(L, k));
                        ad_sum = ad_sum + (double)d;
                        if (lane == 0) {
                            sv->res.alt[k] = (uint8_t)bv_alt_at(L, k);
                            sv->res.af[k] = L.af[k];
                            sv->res.caf[k] = (double)d / (int)total;
                        }
                    }
                }
                double qd = qual / ad_sum;
                if (qd == 0) qd = 0.0;
                if (lane == 0) {
                    sv->res.n_alt = (uint8_t)L.n_alt;
                    sv->res.qual = qual;
                    sv->res.qd = qd;
                }
            }
            // VCF strand-bias table w.r.t. the chosen ALTs (caller.cpp:1164)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b == ref) { v_rf += fwd[b]; v_rr += rev[b]; }
                else if ((alt_mask >> b) & 1u) { v_af += fwd[b]; v_ar += rev[b]; }
            }
            have_var = true;
            {
                const uint32_t vt[4] = {v_rf, v_rr, v_af, v_ar};
                work.tables_known(vt, (v_rf == c_rf && v_rr == c_rr && v_af == c_af && v_ar == c_ar) ? 1 : 2, lane);
            }
            // base-quality rank sum from the (base, phred) counts this pass already holds (caller.cpp:1157)
            {
                unsigned long long n1 = (ref < 4) ? bv_sel4u(depth, ref) : 0ull, n2 = ad_sum_u;
                unsigned long long below = 0, twoR = 0;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    int q = w * 64 + lane;
                    uint32_t rv = 0, av = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        uint32_t c = hq(b, q);
                        if (b == ref) rv += c;
                        else if ((alt_mask >> b) & 1u) av += c;
                    }
                    twoR += bv_ranksum_window(rv, av, n1 + n2, below, lane);
                }
                bq_ranksum = bv_ranksum_phred(twoR, n1, n2);
            }
        }
        // ---- strand bias: FS / SOR of the CVG table, then of the VCF table unless it is the same 2x2 table (it is
        // whenever the chosen ALTs are all the non-ref bases seen)
        if (!(a.flags & BV_FLAG_SKIP_FISHER)) {
            const bool same = have_var && v_rf == c_rf && v_rr == c_rr && v_af == c_af && v_ar == c_ar;
            const int ntab = (have_var && !same) ? 2 : 1;
            double c_fs = 0, c_sor = 0, v_fs = 0, v_sor = 0;
            {
                const uint32_t ct[4] = {c_rf, c_rr, c_af, c_ar}, vt[4] = {v_rf, v_rr, v_af, v_ar};
                work.strand_bias(ct, vt, ntab, lane, a.lnfact, c_fs, c_sor, v_fs, v_sor, flags);
            }
            if (same) { v_fs = c_fs; v_sor = c_sor; }
            if (lane == 0) {
                sv->res.cvg_sb[0] = c_rf; sv->res.cvg_sb[1] = c_rr; sv->res.cvg_sb[2] = c_af; sv->res.cvg_sb[3] = c_ar;
                sv->res.cvg_fs = c_fs;
                sv->res.cvg_sor = c_sor;
                if (have_var) {
                    sv->res.var_sb[0] = v_rf; sv->res.var_sb[1] = v_rr; sv->res.var_sb[2] = v_af; sv->res.var_sb[3] = v_ar;
                    sv->res.var_fs = v_fs;
                    sv->res.var_sor = v_sor;
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) sv->res.depth[b] = depth[b];
            sv->res.total_depth = total;
            sv->res.status = flags;
            sv->res.chi2 = L.chi2;
            sv->res.em_iters = (uint16_t)L.em_iters;
            sv->res.n_em = (uint8_t)L.n_em;
            sv->res.mq_ranksum = qnan;
            sv->res.rpr_ranksum = qnan;
            sv->res.bq_ranksum = bq_ranksum;
            if (!DEFER_LIST && L.n_alt > 0) {
                uint32_t slot = atomicAdd(&a.counters[BV_CTR_VARIANTS], 1u);
                a.var_list[slot] = site;
            }
            if (L.zero_freq) atomicAdd(&a.counters[BV_CTR_ZEROFREQ], 1u);
        }
        is_variant = L.n_alt > 0;
    }
    bv_lrt_sync<0>();
    if (lane < REC_WORDS) reinterpret_cast<uint32_t *>(&a.out[site])[lane] = res_words[lane];
    return is_variant;
}

// Tally histogram -> record: the prologue (totals + bins) and the solve, on one wave.
template <bool ALIAS, class WORK = BvSoloWork>
__device__ __forceinline__ void bv_solve_site_wave(BvSolveArgs a, uint32_t site, BV_LDS uint32_t *hist_l,
                                                  BV_LDS uint32_t *bin_code_l, BV_LDS uint32_t *bin_cnt_l,
                                                  BV_LDS BvSolverScratch *sv_l, BV_LDS const double *tab_hit_l,
                                                  BV_LDS const double *tab_miss_l, int lane, const WORK &work = WORK()) {
    uint32_t *hist = (uint32_t *)hist_l;
    uint32_t *bin_code = ALIAS ? hist + BV_ALIAS_CODE_OFF : (uint32_t *)bin_code_l;
    uint32_t *bin_cnt = ALIAS ? hist + BV_ALIAS_CNT_OFF : (uint32_t *)bin_cnt_l;
    BvSolverScratch *sv = (BvSolverScratch *)sv_l;
    constexpr int REC_WORDS = (int)(sizeof(bv_site_result) / 4);
    uint32_t *res_words = reinterpret_cast<uint32_t *>(&sv->res);
    if (lane < REC_WORDS) res_words[lane] = 0u;

    BvSiteSums S;
    // phred-0 calls per base, read before ALIAS-mode bins can overwrite anything (they never touch
    // phred < 128, but keep every histogram read of the prologue in one place)
    S.q0_mask = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        if (hist[b << 8] + hist[(b | 4) << 8]) S.q0_mask |= 1u << b;
    bv_prologue_wave<ALIAS>(hist, bin_code, bin_cnt, lane, S.fwd, S.rev, &S.nb, &S.badq);
    bv_lrt_sync<0>();  // bin_code / bin_cnt / res zeroing visible to every lane
    BvHqFromHist hq{hist};
    bv_site_solve<ALIAS, BvHqFromHist, false, WORK>(a, site, S, bin_code, bin_cnt, hq, sv, (const double *)tab_hit_l, (const double *)tab_miss_l, lane, work);
}
