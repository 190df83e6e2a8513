// bv_device.h -- gfx950 device building blocks of the per-site basetype solver.
//
// Everything here operates on a per-site (base x phred) histogram that the tally loop
// builds in LDS.  The key identity: the reference's per-sample
// likelihood row depends only on (first base in ACGT, phred q) -- src/basetype.cpp:47-64 --
// so its n_cov x 4 EM collapses exactly onto <= 4 x 94 weighted bins.
//
// Execution model: wave64.  Every solver routine is a *wave-level* function: the 64 lanes
// of one wavefront cooperate through DPP scans (lane 63's total is broadcast with
// v_readlane, so every lane holds the bit-identical sum), and independent routines
// (the up-to-4 EM runs of one LRT level, Fisher tests, rank sums) are spread over the
// waves of the workgroup.  No MFMA: this is categorical-table arithmetic in FP64.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/basevar_amd.h"
#include "../../include/basevar_amd_diag.h"  // diagnostic / A-B flag bits the kernels honour

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
#define BV_HOSTLOG_N 274  // ln2hi, ln2lo, A[5], B[11], 128 x {invc, logc}: glibc's __log_data

// native 16-byte vector: one lane's share of a coalesced 1 KiB wave load
typedef uint32_t bv_u32x4 __attribute__((ext_vector_type(4)));

// (1 - eps_q) and eps_q / 3 for q = 0..127, filled on the host with glibc exp() so that
// eps is bit-identical to the reference's exp((qchar - 33) * MLN10TO10), basetype.cpp:47.
struct BvTables {
    double hit[BV_QBINS];
    double miss[BV_QBINS];
    const double *lnfact;  // lnfact[k] = lgamma(k + 1) (host glibc), k < lnfact_n; device memory
    uint32_t lnfact_n;
    uint32_t pad_;
    double loghit[BV_QBINS];   // log(hit[q]), log(miss[q]) with the host's log(): the per-sample log-marginals of a
    double logmiss[BV_QBINS];  // single-base subset (algorithm.h:243 with f == 1), see bv_lrt
    // The host libm's own log() data, found in the loaded libm at engine creation and accepted only after the
    // restated algorithm (bv_log_host below) reproduced the host's log() bit for bit on a few hundred thousand
    // arguments: ln2hi, ln2lo, A[5], B[11], then 128 x {invc, logc}.  hostlog[BV_HOSTLOG_N] != 0 <=> usable.
    // Must follow logmiss directly: the ordered replay finds it as logmiss + BV_QBINS (bv_hostlog_of).
    double hostlog[BV_HOSTLOG_N + 2];
};
static_assert(offsetof(BvTables, hostlog) == offsetof(BvTables, logmiss) + sizeof(double) * BV_QBINS, "hostlog follows logmiss");

// `logmiss` (BvTables::logmiss in device memory, or NULL) -> the host-log table, or NULL when the engine could not
// verify it (the ordered replay then uses the device library's log, ulps from the host's)
__device__ __forceinline__ const double *bv_hostlog_of(const double *logmiss) {
    if (!logmiss) return nullptr;
    const double *t = logmiss + BV_QBINS;
    return t[BV_HOSTLOG_N] != 0. ? t : nullptr;
}

// log(x) the way the host's glibc computes it on an FMA-capable x86-64 (sysdeps/ieee754/dbl-64/e_log.c, the variant
// its ifunc picks there): a 128-entry table of (1/c, log c), a degree-5 polynomial in r = z/c - 1, the near-1 branch
// with the split r = rhi + rlo, every fused multiply-add exactly where the compiled routine has one.  The reference's
// EM compares and truncates sums of such logs (algorithm.h:238-255), so at tie-prone shallow sites the last bit matters.
// T: BvTables::hostlog.  The same restatement runs on the host against log() before the table is accepted.
__device__ __forceinline__ double bv_log_host(double x, const double *__restrict__ T) {
    const double *A = T + 2, *Bc = T + 7, *tab = T + 18;
    uint64_t ix = (uint64_t)__double_as_longlong(x);
    const uint64_t LO = 0x3fee000000000000ull, HI = 0x3ff1090000000000ull;  // 1 - 2^-4, 1 + 0x1.09p-4
    if (ix - LO < HI - LO) {
        if (ix == 0x3ff0000000000000ull) return 0.;
        const double r = __dadd_rn(x, -1.0), r2 = __dmul_rn(r, r), r3 = __dmul_rn(r, r2);
        const double pA = __fma_rn(r2, Bc[3], __fma_rn(r, Bc[2], Bc[1]));
        const double pB = __fma_rn(r2, Bc[6], __fma_rn(r, Bc[5], Bc[4]));
        const double pC = __fma_rn(r3, Bc[10], __fma_rn(r2, Bc[9], __fma_rn(r, Bc[8], Bc[7])));
        const double p = __fma_rn(__fma_rn(pC, r3, pB), r3, pA);
        const double t = __fma_rn(r, 0x1p27, r), rhi = __fma_rn(-0x1p27, r, t), rlo = __dadd_rn(r, -rhi);
        const double s = __dmul_rn(rhi, rhi);
        const double hi = __fma_rn(s, Bc[0], r);
        const double lo = __fma_rn(s, Bc[0], __dadd_rn(r, -hi));
        const double lo2 = __fma_rn(__dmul_rn(Bc[0], rlo), __dadd_rn(rhi, r), lo);
        return __dadd_rn(__fma_rn(p, r3, lo2), hi);
    }
    const uint32_t top = (uint32_t)(ix >> 48);
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        if ((ix << 1) == 0ull) return -__longlong_as_double(0x7ff0000000000000ll);
        if (ix == 0x7ff0000000000000ull) return x;
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return __longlong_as_double(0x7ff8000000000000ll);
        ix = (uint64_t)__double_as_longlong(__dmul_rn(x, 0x1p52));  // subnormal: scale up, take it off the exponent
        ix -= 52ull << 52;
    }
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const uint32_t i = (uint32_t)(tmp >> 45) & 127u;
    const int k = (int)((int64_t)tmp >> 52);
    const double z = __longlong_as_double((long long)(ix - (tmp & (0xfffull << 52))));
    const double invc = tab[2u * i], logc = tab[2u * i + 1u], kd = (double)k;
    const double r = __fma_rn(z, invc, -1.0);
    const double w = __fma_rn(kd, T[0], logc), hi = __dadd_rn(r, w);
    const double lo = __fma_rn(kd, T[1], __dadd_rn(__dadd_rn(w, -hi), r));
    const double r2 = __dmul_rn(r, r), r3 = __dmul_rn(r, r2);
    const double p = __fma_rn(__fma_rn(r, A[4], A[3]), r2, __fma_rn(r, A[2], A[1]));
    return __dadd_rn(__fma_rn(r3, p, __fma_rn(r2, A[0], lo)), hi);
}

// ------------------------------------------------------------------ wave reductions
// All cross-lane sums use DPP (data-parallel primitives: a VALU operand read from another
// lane), not ds_bpermute: a bpermute round trip costs an LDS latency per butterfly step,
// six dependent steps per reduction, and the solver does several reductions per EM pass.
// Pattern (gfx9 family): inclusive scan inside each row of 16 lanes by row_shr 1,2,4,8,
// then row_bcast15 / row_bcast31 carry the row totals across rows; lane 63 ends up holding
// the wave total, which v_readlane broadcasts.  Fixed order => deterministic FP sums, and
// every lane receives the bit-identical value.
#define BV_DPP_ROW_SHR(n) (0x110 + (n))
#define BV_DPP_BCAST15 0x142
#define BV_DPP_BCAST31 0x143

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int bv_dpp_i32(int ident, int v) {
    return __builtin_amdgcn_update_dpp(ident, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double bv_dpp_f64(double v) {  // identity 0.0
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = bv_dpp_i32<CTRL, ROW_MASK>(0, lo);
    hi = bv_dpp_i32<CTRL, ROW_MASK>(0, hi);
    return __hiloint2double(hi, lo);
}
// the same with identity 1.0 (for products): lanes without a source read 1.0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double bv_dpp_f64_one(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = bv_dpp_i32<CTRL, ROW_MASK>(0, lo);
    hi = bv_dpp_i32<CTRL, ROW_MASK>(0x3FF00000, hi);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long bv_dpp_u64(unsigned long long v) {
    int lo = (int)(uint32_t)v, hi = (int)(uint32_t)(v >> 32);
    lo = bv_dpp_i32<CTRL, ROW_MASK>(0, lo);
    hi = bv_dpp_i32<CTRL, ROW_MASK>(0, hi);
    return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ int bv_readlane63_i32(int v) { return __builtin_amdgcn_readlane(v, 63); }

// inclusive prefix sums over the 64 lanes
__device__ __forceinline__ double bv_wave_incl_scan_f64(double v) {
    v += bv_dpp_f64<BV_DPP_ROW_SHR(1), 0xf>(v);
    v += bv_dpp_f64<BV_DPP_ROW_SHR(2), 0xf>(v);
    v += bv_dpp_f64<BV_DPP_ROW_SHR(4), 0xf>(v);
    v += bv_dpp_f64<BV_DPP_ROW_SHR(8), 0xf>(v);
    v += bv_dpp_f64<BV_DPP_BCAST15, 0xa>(v);
    v += bv_dpp_f64<BV_DPP_BCAST31, 0xc>(v);
    return v;
}
// inclusive prefix PRODUCTS over the 64 lanes
__device__ __forceinline__ double bv_wave_incl_scan_prod_f64(double v) {
    v *= bv_dpp_f64_one<BV_DPP_ROW_SHR(1), 0xf>(v);
    v *= bv_dpp_f64_one<BV_DPP_ROW_SHR(2), 0xf>(v);
    v *= bv_dpp_f64_one<BV_DPP_ROW_SHR(4), 0xf>(v);
    v *= bv_dpp_f64_one<BV_DPP_ROW_SHR(8), 0xf>(v);
    v *= bv_dpp_f64_one<BV_DPP_BCAST15, 0xa>(v);
    v *= bv_dpp_f64_one<BV_DPP_BCAST31, 0xc>(v);
    return v;
}
__device__ __forceinline__ uint32_t bv_wave_incl_scan_u32(uint32_t v, int /*lane*/) {
    v += (uint32_t)bv_dpp_i32<BV_DPP_ROW_SHR(1), 0xf>(0, (int)v);
    v += (uint32_t)bv_dpp_i32<BV_DPP_ROW_SHR(2), 0xf>(0, (int)v);
    v += (uint32_t)bv_dpp_i32<BV_DPP_ROW_SHR(4), 0xf>(0, (int)v);
    v += (uint32_t)bv_dpp_i32<BV_DPP_ROW_SHR(8), 0xf>(0, (int)v);
    v += (uint32_t)bv_dpp_i32<BV_DPP_BCAST15, 0xa>(0, (int)v);
    v += (uint32_t)bv_dpp_i32<BV_DPP_BCAST31, 0xc>(0, (int)v);
    return v;
}
__device__ __forceinline__ unsigned long long bv_wave_incl_scan_u64(unsigned long long v) {
    v += bv_dpp_u64<BV_DPP_ROW_SHR(1), 0xf>(v);
    v += bv_dpp_u64<BV_DPP_ROW_SHR(2), 0xf>(v);
    v += bv_dpp_u64<BV_DPP_ROW_SHR(4), 0xf>(v);
    v += bv_dpp_u64<BV_DPP_ROW_SHR(8), 0xf>(v);
    v += bv_dpp_u64<BV_DPP_BCAST15, 0xa>(v);
    v += bv_dpp_u64<BV_DPP_BCAST31, 0xc>(v);
    return v;
}
__device__ __forceinline__ double bv_wave_sum(double v) {
    v = bv_wave_incl_scan_f64(v);
    int lo = bv_readlane63_i32(__double2loint(v)), hi = bv_readlane63_i32(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint32_t bv_wave_sum_u32(uint32_t v) {
    return (uint32_t)bv_readlane63_i32((int)bv_wave_incl_scan_u32(v, 0));
}
__device__ __forceinline__ unsigned long long bv_wave_sum_u64(unsigned long long v) {
    v = bv_wave_incl_scan_u64(v);
    uint32_t lo = (uint32_t)bv_readlane63_i32((int)(uint32_t)v), hi = (uint32_t)bv_readlane63_i32((int)(uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
// Packed sums: gfx950's v_permlane32_swap / v_permlane16_swap fold two (four) per-lane values into
// one register whose halves (rows of 16 lanes) carry the partial sums of different values, so two
// (four) wave totals cost one scan instead of two (four).  Fixed order => deterministic.
__device__ __forceinline__ double bv_fold32_f64(double a, double b) {
    // lanes 0-31: a[l] + a[l+32];  lanes 32-63: b[l-32] + b[l]
    unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    auto rl = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto rh = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ __forceinline__ double bv_fold16_f64(double p, double q) {
    // rows of 16 lanes: [p0+p1, q0+q1, p2+p3, q2+q3]
    unsigned plo = (unsigned)__double2loint(p), phi = (unsigned)__double2hiint(p);
    unsigned qlo = (unsigned)__double2loint(q), qhi = (unsigned)__double2hiint(q);
    auto rl = __builtin_amdgcn_permlane16_swap(plo, qlo, false, false);
    auto rh = __builtin_amdgcn_permlane16_swap(phi, qhi, false, false);
    return __hiloint2double((int)rh[0], (int)rl[0]) + __hiloint2double((int)rh[1], (int)rl[1]);
}
__device__ __forceinline__ double bv_readlane_f64(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void bv_wave_sum2(double a, double b, double &sa, double &sb) {
    double v = bv_fold32_f64(a, b);
    v += bv_dpp_f64<BV_DPP_ROW_SHR(1), 0xf>(v);
    v += bv_dpp_f64<BV_DPP_ROW_SHR(2), 0xf>(v);
    v += bv_dpp_f64<BV_DPP_ROW_SHR(4), 0xf>(v);
    v += bv_dpp_f64<BV_DPP_ROW_SHR(8), 0xf>(v);
    v += bv_dpp_f64<BV_DPP_BCAST15, 0xa>(v);
    sa = bv_readlane_f64(v, 31);
    sb = bv_readlane_f64(v, 63);
}
__device__ __forceinline__ void bv_wave_sum4(double a, double b, double c, double d, double &sa, double &sb, double &sc,
                                             double &sd) {
    double v = bv_fold16_f64(bv_fold32_f64(a, b), bv_fold32_f64(c, d));  // rows: [a, c, b, d]
    v += bv_dpp_f64<BV_DPP_ROW_SHR(1), 0xf>(v);
    v += bv_dpp_f64<BV_DPP_ROW_SHR(2), 0xf>(v);
    v += bv_dpp_f64<BV_DPP_ROW_SHR(4), 0xf>(v);
    v += bv_dpp_f64<BV_DPP_ROW_SHR(8), 0xf>(v);
    sa = bv_readlane_f64(v, 15);
    sc = bv_readlane_f64(v, 31);
    sb = bv_readlane_f64(v, 47);
    sd = bv_readlane_f64(v, 63);
}
// Eight u32 wave totals with one scan: 32-lane fold, 16-lane fold, 8-lane fold, then a 3-step scan
// inside groups of 8 lanes (26 VALU instead of 8 x 13).
__device__ __forceinline__ uint32_t bv_fold32_u32(uint32_t a, uint32_t b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return r[0] + r[1];
}
__device__ __forceinline__ uint32_t bv_fold16_u32(uint32_t p, uint32_t q) {
    auto r = __builtin_amdgcn_permlane16_swap(p, q, false, false);
    return r[0] + r[1];
}
#define BV_DPP_ROW_ROR(n) (0x120 + (n))
__device__ __forceinline__ void bv_wave_sum8_u32(const uint32_t v[8], uint32_t out[8], int lane) {
    uint32_t t0 = bv_fold16_u32(bv_fold32_u32(v[0], v[1]), bv_fold32_u32(v[2], v[3]));  // rows: v0, v2, v1, v3
    uint32_t t1 = bv_fold16_u32(bv_fold32_u32(v[4], v[5]), bv_fold32_u32(v[6], v[7]));  // rows: v4, v6, v5, v7
    t0 += (uint32_t)bv_dpp_i32<BV_DPP_ROW_ROR(8), 0xf>(0, (int)t0);  // both halves of a row: its 8 pair sums
    t1 += (uint32_t)bv_dpp_i32<BV_DPP_ROW_ROR(8), 0xf>(0, (int)t1);
    uint32_t w = (lane & 8) ? t1 : t0;
    w += (uint32_t)bv_dpp_i32<BV_DPP_ROW_SHR(1), 0xf>(0, (int)w);
    w += (uint32_t)bv_dpp_i32<BV_DPP_ROW_SHR(2), 0xf>(0, (int)w);
    w += (uint32_t)bv_dpp_i32<BV_DPP_ROW_SHR(4), 0xf>(0, (int)w);
    out[0] = (uint32_t)__builtin_amdgcn_readlane((int)w, 7);
    out[4] = (uint32_t)__builtin_amdgcn_readlane((int)w, 15);
    out[2] = (uint32_t)__builtin_amdgcn_readlane((int)w, 23);
    out[6] = (uint32_t)__builtin_amdgcn_readlane((int)w, 31);
    out[1] = (uint32_t)__builtin_amdgcn_readlane((int)w, 39);
    out[5] = (uint32_t)__builtin_amdgcn_readlane((int)w, 47);
    out[3] = (uint32_t)__builtin_amdgcn_readlane((int)w, 55);
    out[7] = (uint32_t)__builtin_amdgcn_readlane((int)w, 63);
}
// min / max: same scan shape with the matching identity
__device__ __forceinline__ int bv_wave_min_i32(int v) {
    const int I = 0x7fffffff;
    v = min(v, bv_dpp_i32<BV_DPP_ROW_SHR(1), 0xf>(I, v));
    v = min(v, bv_dpp_i32<BV_DPP_ROW_SHR(2), 0xf>(I, v));
    v = min(v, bv_dpp_i32<BV_DPP_ROW_SHR(4), 0xf>(I, v));
    v = min(v, bv_dpp_i32<BV_DPP_ROW_SHR(8), 0xf>(I, v));
    v = min(v, bv_dpp_i32<BV_DPP_BCAST15, 0xa>(I, v));
    v = min(v, bv_dpp_i32<BV_DPP_BCAST31, 0xc>(I, v));
    return bv_readlane63_i32(v);
}
__device__ __forceinline__ int bv_wave_max_i32(int v) {
    const int I = (int)0x80000000;
    v = max(v, bv_dpp_i32<BV_DPP_ROW_SHR(1), 0xf>(I, v));
    v = max(v, bv_dpp_i32<BV_DPP_ROW_SHR(2), 0xf>(I, v));
    v = max(v, bv_dpp_i32<BV_DPP_ROW_SHR(4), 0xf>(I, v));
    v = max(v, bv_dpp_i32<BV_DPP_ROW_SHR(8), 0xf>(I, v));
    v = max(v, bv_dpp_i32<BV_DPP_BCAST15, 0xa>(I, v));
    v = max(v, bv_dpp_i32<BV_DPP_BCAST31, 0xc>(I, v));
    return bv_readlane63_i32(v);
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
// updating p(i) multiplicatively and re-seeding it from lgamma() every 11 tables
// (hypergeo_acc, kfunc.c:220-243).  Here every table probability is evaluated on its own,
// one table per lane, from log-factorials:
//   p(i) = exp( lbinom(n1_, i) + lbinom(n - n1_, n_1 - i) - lbinom(n, n_1) )   (kfunc.c:197-212)
// log(n!) == lgamma(n + 1) is the only way kfunc.c:197-201 uses lgamma; the engine keeps a table of
// it in HBM (L2-resident), filled on the host with glibc's lgamma -- the reference's own values --
// for every n up to the largest possible depth.  Arguments beyond the table fall back to the
// Stirling series (7 terms, |error| < 1e-16 relative for n + 1 >= 17).  Values differ from the
// reference's multiplicative walk at the 1e-13 relative level (parity bar: 1e-6).
struct BvLnTab {
    const double *t;  // t[k] = lgamma(k + 1), k < n
    int n;
};
__device__ __forceinline__ double bv_lnfact_series(int n) {
    if (n < 16) {
        const double T0 = 0.0, T2 = 0.693147180559945, T3 = 1.7917594692280554, T4 = 3.178053830347945,
                     T5 = 4.787491742782047, T6 = 6.579251212010102, T7 = 8.525161361065415, T8 = 10.604602902745249,
                     T9 = 12.801827480081467, T10 = 15.104412573075514, T11 = 17.502307845873887,
                     T12 = 19.987214495661885, T13 = 22.55216385312342, T14 = 25.191221182738683,
                     T15 = 27.89927138384089;
        // select chain instead of a memory table: no scratch / constant-memory traffic
        double lo = n < 2 ? T0 : (n == 2 ? T2 : (n == 3 ? T3 : (n == 4 ? T4 : (n == 5 ? T5 : (n == 6 ? T6 : T7)))));
        double hi = n == 8 ? T8 : (n == 9 ? T9 : (n == 10 ? T10 : (n == 11 ? T11 : (n == 12 ? T12 : (n == 13 ? T13 : (n == 14 ? T14 : T15))))));
        return n < 8 ? lo : hi;
    }
    const double x = (double)n + 1.0;
    const double xi = 1.0 / x, x2 = xi * xi;
    double s = 1.0 / 156.0;
    s = s * x2 + (-691.0 / 360360.0);
    s = s * x2 + (1.0 / 1188.0);
    s = s * x2 + (-1.0 / 1680.0);
    s = s * x2 + (1.0 / 1260.0);
    s = s * x2 + (-1.0 / 360.0);
    s = s * x2 + (1.0 / 12.0);
    s *= xi;
    return (x - 0.5) * log(x) - x + 0.91893853320467274178 + s;
}
__device__ __forceinline__ double bv_lnfact(const BvLnTab &T, int n) {
#ifdef BV_LNFACT_TABLE_ONLY
    // short-row kernels: a depth is at most 65,535 there and the engine's table never has fewer than 65,536 entries
    // (bv_engine_create), so the series -- a log() and a select chain inlined at every one of ~60 call sites -- stays out
    return T.t[n < T.n ? n : T.n - 1];
#else
    if (n < T.n) return T.t[n];
    return bv_lnfact_series(n);
#endif
}

// One 2x2 table family: margins fixed, n11 = i varies over [imin, imax].
// lbinom(n, k) = lnfact(n) - lnfact(k) - lnfact(n - k); its k == 0 / k == n special case
// (kfunc.c:199) needs no branch here because lnfact(0) == 0 makes the difference exactly 0.
struct BvHyper {
    BvLnTab T;
    int n1_, n_1, n, n22off;
    double lf_n1;   // lnfact(n1_)
    double lf_n2;   // lnfact(n - n1_)
    double lb3;     // lbinom(n, n_1)
};
__device__ __forceinline__ double bv_hyper_logp(const BvHyper &h, int i) {
    double a = h.lf_n1 - bv_lnfact(h.T, i) - bv_lnfact(h.T, h.n1_ - i);
    double b = h.lf_n2 - bv_lnfact(h.T, h.n_1 - i) - bv_lnfact(h.T, i + h.n22off);
    return a + b - h.lb3;
}
__device__ __forceinline__ double bv_hyper_p(const BvHyper &h, int i) { return exp(bv_hyper_logp(h, i)); }
// p(i + 1) / p(i): the multiplicative step of the reference's own walk (hypergeo_acc, kfunc.c:226-231)
__device__ __forceinline__ double bv_hyper_ratio(const BvHyper &h, int i) {
    return ((double)(h.n1_ - i) * (double)(h.n_1 - i)) / ((double)(i + 1) * (double)(i + 1 + h.n22off));
}
__device__ __forceinline__ void bv_hyper_init(BvHyper &h, const BvLnTab &T, int n1_, int n_1, int n) {
    h.T = T;
    h.n1_ = n1_; h.n_1 = n_1; h.n = n; h.n22off = n - n1_ - n_1;
    h.lf_n1 = bv_lnfact(T, n1_);
    h.lf_n2 = bv_lnfact(T, n - n1_);
    h.lb3 = bv_lnfact(T, n) - bv_lnfact(T, n_1) - bv_lnfact(T, n - n_1);
}

// kt_fisher_exact (two-sided), wave-parallel.  Semantics of the reference's two loops
// (kfunc.c:291-307): with lo = 0.99999999 q and hi = 1.00000001 q,
//   L* = first table from the left  with p >= lo,  left  = sum_{i < L*} p(i) + (p(L*) < hi ? p(L*) : 0)
//   R* = first table from the right with p >= lo,  right = sum_{i > R*} p(i) + (p(R*) < hi ? p(R*) : 0)
//   two = min(1, left + right)
// The pmf is unimodal, so the tables with p < lo are exactly those left of L* and right of R*:
//   two = sum_{p(i) < lo} p(i) + [p(L*) < hi] p(L*) + [p(R*) < hi] p(R*),
// with L* / R* the lowest / highest lane of a ballot.
// Regimes:
//   a row margin <= 12   product form, no log-factorials (the usual all-sites CVG case);
//   R <= 64 tables       one table per lane;
//   more                 rounds of 64 tables over [imin, imax]; beyond 256 tables a 64-point probe of
//                        log p on either side first skips the far tails whose terms are below
//                        q * 2^-86 (they cannot change a double-precision sum that is >= q).
__device__ inline double bv_fisher_two_sided_wave(int n11, int n12, int n21, int n22, int lane, const BvLnTab &T) {
    const int n1_ = n11 + n12, n_1 = n11 + n21, n = n11 + n12 + n21 + n22;
    const int imax = (n_1 < n1_) ? n_1 : n1_;
    int imin = n1_ + n_1 - n;
    if (imin < 0) imin = 0;
    if (imin == imax) return 1.;
    {
        // ---- a row margin of at most 12 reads (the usual case for the all-sites CVG test: a hom-ref
        // site carries only a handful of non-reference reads).  With m = that row's total and j = how
        // many of them sit in column 1, the table probability is a ratio of two products of m terms
        //     p(j) = C(m, j) * (n_1)_j * (n_2)_{m-j} / (n)_m        ((x)_k: falling factorial)
        // -- exact to ~1e-15.  The tables are walked in j instead of n11; the two-sided sum is
        // symmetric under that relabelling.  At most 13 tables: they sit in the first row of lanes.
        const int n2_ = n21 + n22, n_2 = n - n_1;
        const bool alt_row = n2_ <= n1_;
        const int m = alt_row ? n2_ : n1_, jobs = alt_row ? n21 : n11;
        if (m <= 12) {
            const int jmin = max(0, m - n_2), jmax = min(m, n_1);  // jmin < jmax because imin < imax
            const int j = jmin + lane;
            const bool have = j <= jmax;
            double pn = 1.0, pd = 1.0;
            for (int t = 0; t < m; ++t) {
                const double num = (t < j) ? (double)(n_1 - t) * (double)(m - t) : (double)(n_2 - (t - j));
                const double den = (t < j) ? (double)(n - t) * (double)(t + 1) : (double)(n - t);
                pn *= num;
                pd *= den;
            }
            const double p = have ? pn / pd : 0.;
            const double q = bv_readlane_f64(p, jobs - jmin);
            if (q == 0.0) return 0.0;
            const double lo = 0.99999999 * q, hi = 1.00000001 * q;
            const bool viol = have && !(p < lo);
            const unsigned long long vm = __ballot(viol);  // never empty: the observed table is in it
            double v = viol ? 0. : p;                        // lanes 0..15 hold everything
            v += bv_dpp_f64<BV_DPP_ROW_SHR(1), 0xf>(v);
            v += bv_dpp_f64<BV_DPP_ROW_SHR(2), 0xf>(v);
            v += bv_dpp_f64<BV_DPP_ROW_SHR(4), 0xf>(v);
            v += bv_dpp_f64<BV_DPP_ROW_SHR(8), 0xf>(v);
            double two = bv_readlane_f64(v, 15);
            const double pL = bv_readlane_f64(p, (int)__builtin_ctzll(vm)), pR = bv_readlane_f64(p, 63 - (int)__builtin_clzll(vm));
            if (pL < hi) two += pL;
            if (pR < hi) two += pR;
            return two > 1. ? 1. : two;
        }
    }
    BvHyper h;
    bv_hyper_init(h, T, n1_, n_1, n);
    const int R = imax - imin + 1;

    if (R <= BV_WAVE) {
        // ---- one table per lane; q is the value of the lane that holds the observed table
        const int i = imin + lane;
        const bool have = i <= imax;
        const double pe = bv_hyper_p(h, have ? i : imax);
        const double p = have ? pe : 0.;
        const double q = bv_readlane_f64(p, n11 - imin);
        if (q == 0.0) return 0.0;  // kfunc.c:260-289
        const double lo = 0.99999999 * q, hi = 1.00000001 * q;
        const bool viol = have && !(p < lo);
        const unsigned long long vm = __ballot(viol);
        double two = bv_wave_sum(viol ? 0. : p);
        const double pL = bv_readlane_f64(p, (int)__builtin_ctzll(vm)), pR = bv_readlane_f64(p, 63 - (int)__builtin_clzll(vm));
        if (pL < hi) two += pL;
        if (pR < hi) two += pR;
        return two > 1. ? 1. : two;
    }

    const double logq = bv_hyper_logp(h, n11);
    const double q = exp(logq);
    if (q == 0.0) return 0.0;  // kfunc.c:260-289
    const double lo = 0.99999999 * q, hi = 1.00000001 * q;
    const int INF = 0x7fffffff;
    int wl = imin, wr = imax;
    if (R > 4 * BV_WAVE) {
        // ---- many tables: skip the far tails
        const double cut = logq - 60.0;  // exp(-60) ~ 2^-86.6
        {
            // probes on [imin, n11]: tables before the last probe that is still below the cut are negligible
            // (rising side, or between mode and n11 where p >= q): a prefix by unimodality
            const int span = n11 - imin, step = span / 63 + 1;
            const int i = imin + lane * step;
            const bool below = (i <= n11) && (bv_hyper_logp(h, min(i, n11)) < cut);
            const int last = bv_wave_max_i32(below ? i : -1);
            if (last >= 0) wl = last;
        }
        {
            const int span = imax - n11, step = span / 63 + 1;
            const int i = imax - lane * step;
            const bool below = (i >= n11) && (bv_hyper_logp(h, max(i, n11)) < cut);
            const int first = bv_wave_min_i32(below ? i : INF);
            if (first != INF) wr = first;
        }
    }
    // ---- rounds of 64 tables, ascending over [wl, wr].  Only the first table of the range comes from log-factorials
    // (four table lookups + exp, an L2 round trip); the others follow by the reference's own multiplicative step
    // p(i+1) = p(i) * ratio(i), as prefix products across the lanes -- no memory access inside the loop.  (~1e-16 per
    // step; the reference re-seeds every 11 tables, which matters at 1e-13, not at the 1e-6 bar.)
    double tail = 0., pL = 0., pR = 0.;
    bool seen = false;
    double pbase = bv_hyper_p(h, wl);
    for (int w = wl; w <= wr; w += BV_WAVE) {
        const int i = w + lane;
        const bool have = i <= wr;
        const double step = (lane == 0 || !have) ? 1.0 : bv_hyper_ratio(h, i - 1);
        const double pe = pbase * bv_wave_incl_scan_prod_f64(step);
        pbase = bv_readlane_f64(pe, 63) * bv_hyper_ratio(h, w + 63);  // first table of the next round (unused after the last)
        const double p = have ? pe : 0.;
        const bool viol = have && !(p < lo);
        tail += viol ? 0. : p;
        const unsigned long long vm = __ballot(viol);
        if (vm != 0ull) {
            if (!seen) pL = bv_readlane_f64(p, (int)__builtin_ctzll(vm));
            seen = true;
            pR = bv_readlane_f64(p, 63 - (int)__builtin_clzll(vm));
        }
    }
    double two = bv_wave_sum(tail);
    if (pL < hi) two += pL;
    if (pR < hi) two += pR;
    return two > 1. ? 1. : two;
}

// strand_bias tail, src/basetype.cpp:277-286
__device__ inline void bv_strand_bias_wave(uint32_t ref_fwd, uint32_t ref_rev, uint32_t alt_fwd, uint32_t alt_rev,
                                           int lane, const BvLnTab &T, double *fs_out, double *sor_out, uint32_t *flags) {
    double fs = -10 * log10(bv_fisher_two_sided_wave((int)ref_fwd, (int)ref_rev, (int)alt_fwd, (int)alt_rev, lane, T));
    if (isinf(fs)) fs = 10000;
    else if (fs == 0) fs = 0.0;
    // basetype.cpp:286 multiplies `int`s; past 2^31 that is UB in the reference.  Its compiled
    // form (gcc -O3 x86-64, pinned by tests/golden/deep_sor.npz) folds the guard
    // `ref_rev * alt_fwd > 0` into "both factors non-zero" and divides the 32-bit wrapped products.
    // Reproduced exactly, and flagged so that a caller can tell such a value is not meaningful.
    int den = (int)(ref_rev * alt_fwd), num = (int)(ref_fwd * alt_rev);
    if ((unsigned long long)ref_rev * alt_fwd > 0x7fffffffull || (unsigned long long)ref_fwd * alt_rev > 0x7fffffffull)
        *flags |= BV_SITE_SOR_OVERFLOW;
    double sor = (ref_rev != 0u && alt_fwd != 0u) ? (double)num / (double)den : 10000;
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
    // (a window without a count adds nothing to the sum nor to `below`: mapq stops at 60, read-position ranks at the read length,
    // so most windows of most rows are empty, and a scan, a 64-bit product and a 64-bit wave sum are ~50 instructions)
    if (__ballot(t != 0u) == 0ull) return 0ull;
    uint32_t incl = bv_wave_incl_scan_u32(t, lane);
    unsigned long long below_v = below + (incl - t);
    unsigned long long term = (unsigned long long)ref_v * (2ull * n - 2ull * below_v - t + 1ull);
    unsigned long long s = bv_wave_sum_u64(term);
    below += (unsigned long long)(uint32_t)bv_readlane63_i32((int)incl);
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
    const uint32_t *code;  // LDS; bin i is at code[bv_bin_at(B, i)]
    const uint32_t *cnt;   // LDS
    uint32_t skip_mask;    // 0: dense arrays; ~127u: 128 entries, then 128 words skipped, ... (bins
                           // stored in the unused upper halves of the histogram rows, bv_pass1.hip)
    const double *hit;     // LDS copy of BvTables::hit  (1 - eps)
    const double *miss;    // LDS copy of BvTables::miss (eps / 3)
    const double *loghit;  // BvTables::loghit / logmiss (device memory, L2-resident), or NULL: single-base subsets then
    const double *logmiss; // run the iterative EM like every other subset
    int nb;                // number of bins (wave-uniform)
    const uint16_t *ord;   // LDS: the site's covered cells in SAMPLE ORDER (call << 8 | phred), or NULL.  When given (shallow
    int n_ord;             // sites), every EM of the LRT replays the reference's per-sample arithmetic literally (bv_em_ordered)
};

__device__ __forceinline__ int bv_bin_at(const BvBins &B, int i) { return i + (int)((uint32_t)i & B.skip_mask); }

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
// Bin i lives in lane i % 64, slot i / 64; only the per-bin marginal of the previous pass
// is carried (in registers).  (A branch-free variant that kept three bins per lane in flight
// for ILP, with the log-marginals in LDS, was measured 13 % SLOWER end to end: the solver wave
// shares its SIMD with tally waves, so its instruction count matters more than its latency.)
__device__ __forceinline__ int bv_em_wave_generic(const BvBins &B, double f[4], double n_cov, double *lr_out, int lane) {
    const double epsilon = (double)0.001f;  // `const float epsilon=0.001`, algorithm.h:213
    const int nslots = (B.nb + BV_WAVE - 1) / BV_WAVE;
    // The reference takes log(marginal) of every sample in every pass, but uses the values only (a) in
    // the convergence term |int(llh_new - llh_old)| (algorithm.h:245, integer abs: zero unless the two logs
    // differ by >= 1) and (b) from the LAST pass, as the log-likelihood sum.  So the previous pass's
    // MARGINAL is carried instead of its log; a pass takes logs only for the bins whose marginal moved
    // by a factor outside (0.37, 2.7) -- |delta log| < 0.995 otherwise, which truncates to 0 whatever the
    // rounding of log() -- and the sum (b) is formed once after the loop.  NaN / 0 / inf marginals fail both
    // comparisons and take the exact path.
    double pm[BV_SLOTS];
#pragma unroll
    for (int s = 0; s < BV_SLOTS; ++s) pm[s] = 1.0;
    int iters = 0;
    for (int k = 0; k <= 100; ++k) {
        double pf0 = 0., pf1 = 0., pf2 = 0., pf3 = 0., delta = 0.;
#pragma unroll
        for (int s = 0; s < BV_SLOTS; ++s) {
            if (s < nslots) {
                const int i = s * BV_WAVE + lane;
                if (i < B.nb) {
                    const int at = bv_bin_at(B, i);
                    const uint32_t code = B.code[at];
                    const double c = (double)B.cnt[at];
                    const uint32_t b = code >> 7;
                    const double hit = B.hit[code & 127u], miss = B.miss[code & 127u];
                    double L0 = (b == 0 ? hit : miss) * f[0];
                    double L1 = (b == 1 ? hit : miss) * f[1];
                    double L2 = (b == 2 ? hit : miss) * f[2];
                    double L3 = (b == 3 ? hit : miss) * f[3];
                    double marg = L0;
                    marg += L1;
                    marg += L2;
                    marg += L3;
                    double r = 1.0 / marg;
                    pf0 += c * (L0 * r);
                    pf1 += c * (L1 * r);
                    pf2 += c * (L2 * r);
                    pf3 += c * (L3 * r);
                    const double old = pm[s];
                    if (k > 0 && !(marg < old * 2.7 && marg > old * 0.37))
                        delta += c * bv_int_abs_trunc(log(marg) - log(old));
                    pm[s] = marg;
                }
            }
        }
        f[0] = bv_wave_sum(pf0) / n_cov;
        f[1] = bv_wave_sum(pf1) / n_cov;
        f[2] = bv_wave_sum(pf2) / n_cov;
        f[3] = bv_wave_sum(pf3) / n_cov;
        if (k == 0) continue;
        delta = bv_wave_sum(delta);
        ++iters;
        if (delta < epsilon) break;
    }
    // log-likelihood sum at the marginals of the last pass (algorithm.h:243, basetype.cpp:123-125)
    double lr = 0.;
#pragma unroll
    for (int s = 0; s < BV_SLOTS; ++s) {
        if (s < nslots) {
            const int i = s * BV_WAVE + lane;
            if (i < B.nb) lr += (double)B.cnt[bv_bin_at(B, i)] * log(pm[s]);
        }
    }
    *lr_out = bv_wave_sum(lr);
    return iters;
}

// The same EM when no call of the site has phred 0 and the start frequencies are not all zero (every
// likelihood and marginal is then a positive finite number).  Bases outside the subset keep
// frequency +0.0 through every pass in the reference as well (0 * x, 0 + x, 0 / n are exact), so their
// terms are skipped rather than computed; the per-base sums of a pass are reduced together
// (bv_wave_sum2/4); the convergence sum is only formed when some bin took the exact-log path; and
// sum / n_cov is sum * (1 / n_cov) (<= 1 ulp; the parity bar is 1e-6).  `in_set`: bit b = base b.
__device__ __forceinline__ int bv_em_wave(const BvBins &B, double f[4], unsigned in_set, double n_cov, double *lr_out,
                                          int lane) {
    const double epsilon = (double)0.001f;
    const int nslots = (B.nb + BV_WAVE - 1) / BV_WAVE;
    const double inv_n = 1.0 / n_cov;
    const bool s0 = in_set & 1u, s1 = in_set & 2u, s2 = in_set & 4u, s3 = in_set & 8u;
    const int nset = __popc(in_set);
    double pm[BV_SLOTS];
#pragma unroll
    for (int s = 0; s < BV_SLOTS; ++s) pm[s] = 1.0;
    int iters = 0;
    for (int k = 0; k <= 100; ++k) {
        double pf0 = 0., pf1 = 0., pf2 = 0., pf3 = 0., delta = 0.;
#pragma unroll
        for (int s = 0; s < BV_SLOTS; ++s) {
            if (s < nslots) {
                const int i = s * BV_WAVE + lane;
                if (i < B.nb) {
                    const int at = bv_bin_at(B, i);
                    const uint32_t code = B.code[at];
                    const double c = (double)B.cnt[at];
                    const uint32_t b = code >> 7;
                    const double hit = B.hit[code & 127u], miss = B.miss[code & 127u];
                    double L0 = 0., L1 = 0., L2 = 0., L3 = 0., marg = 0.;
                    if (s0) { L0 = (b == 0 ? hit : miss) * f[0]; marg += L0; }
                    if (s1) { L1 = (b == 1 ? hit : miss) * f[1]; marg += L1; }
                    if (s2) { L2 = (b == 2 ? hit : miss) * f[2]; marg += L2; }
                    if (s3) { L3 = (b == 3 ? hit : miss) * f[3]; marg += L3; }
                    const double r = c / marg;
                    if (s0) pf0 += L0 * r;
                    if (s1) pf1 += L1 * r;
                    if (s2) pf2 += L2 * r;
                    if (s3) pf3 += L3 * r;
                    const double old = pm[s];
                    if (k > 0 && !(marg < old * 2.7 && marg > old * 0.37))
                        delta += c * bv_int_abs_trunc(log(marg) - log(old));
                    pm[s] = marg;
                }
            }
        }
        if (nset == 1) {
            const double t = bv_wave_sum(s0 ? pf0 : (s1 ? pf1 : (s2 ? pf2 : pf3)));
            pf0 = pf1 = pf2 = pf3 = t;
        } else if (nset == 2) {
            // the two members, in base order
            const double x = s0 ? pf0 : (s1 ? pf1 : pf2);
            const double y = s3 ? pf3 : ((s2 && (s0 || s1)) ? pf2 : pf1);
            double tx, ty;
            bv_wave_sum2(x, y, tx, ty);
            pf0 = tx;                                  // only read when s0
            pf1 = s0 ? ty : tx;                        // s1: second member iff s0 is the first
            pf2 = (s0 || s1) ? ty : tx;                // s2: second member iff an earlier base is in the set
            pf3 = ty;                                  // s3 is always the second member
        } else {
            bv_wave_sum4(pf0, pf1, pf2, pf3, pf0, pf1, pf2, pf3);
        }
        f[0] = s0 ? pf0 * inv_n : 0.;
        f[1] = s1 ? pf1 * inv_n : 0.;
        f[2] = s2 ? pf2 * inv_n : 0.;
        f[3] = s3 ? pf3 * inv_n : 0.;
        if (k == 0) continue;
        ++iters;
        if (__ballot(delta != 0.) == 0ull) break;  // every bin's log-marginal moved by < 1: the sum is 0
        if (bv_wave_sum(delta) < epsilon) break;
    }
    double lr = 0.;
#pragma unroll
    for (int s = 0; s < BV_SLOTS; ++s) {
        if (s < nslots) {
            const int i = s * BV_WAVE + lane;
            if (i < B.nb) lr += (double)B.cnt[bv_bin_at(B, i)] * log(pm[s]);
        }
    }
    *lr_out = bv_wave_sum(lr);
    return iters;
}

// ------------------------------------------------------------------ shallow sites: the reference's own order
// A histogram cannot see the ORDER of the samples, and when two allele subsets have mathematically equal likelihood
// (two bases seen once each at the same phred ...) the reference's pick depends on it: its log-likelihood is a
// sequential sum of per-sample terms (algorithm.h:24-41, basetype.cpp:123).  All such ties sit at shallow sites, so
// sites with at most BV_ORD_MAX covered samples replay the reference literally: the covered cells are gathered in
// sample order, one sample per lane, e_step / m_step as algorithm.h:148-198 (per-sample products, the marginal summed
// over A, C, G, T in that order, four IEEE divisions, the m_step's sums taken over the samples in order, no fused
// multiply-add), the convergence term and the final sum in sample order, and log() as the host's libm computes it
// (bv_log_host; the device library's log only when the engine could not verify the host's table).
#define BV_ORD_MAX 64
// An ordered-cell buffer in LDS: BV_ORD_MAX cells (uint16_t) followed by the scratch of bv_em_ordered's in-order sums
// (BV_SEQ_N quantities x BV_SEQ_STRIDE doubles); declare it `alignas(8) uint16_t ord[BV_ORD_ALLOC]`.
#define BV_SEQ_N 5
#define BV_SEQ_STRIDE 65 /* 64 + 1: the BV_SEQ_N lanes that walk the rows read different banks */
#define BV_ORD_ALLOC (BV_ORD_MAX + 4 * BV_SEQ_N * BV_SEQ_STRIDE)

// covered cells of one row in sample order -> ord[] (at most BV_ORD_MAX); returns how many the row holds.  With `gid`:
// only the samples of pop-group `g` (gid: one byte per sample, readable in 16-byte chunks up to the row's last chunk).
__device__ __noinline__ uint32_t bv_gather_ordered(const uint8_t *bs_row, const uint8_t *q_row, uint32_t n_samples, uint16_t *ord,
                                                  int lane, const uint8_t *gid = nullptr, uint32_t g = 0) {
    const uint32_t n_chunks = (n_samples + 15u) >> 4;
    uint32_t count = 0;
    // a round is 64 chunks of 16 cells, one per lane: calls, phreds (and group ids) as 16-byte loads, the next round's issued
    // before this round's cells are placed -- the walk is one wave's, its latency is the whole cost
    bv_u32x4 nb_ = bv_u32x4{0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u}, nq_ = bv_u32x4{0u, 0u, 0u, 0u}, ng_ = bv_u32x4{0u, 0u, 0u, 0u};
    auto fetch = [&](uint32_t c0) {
        const uint32_t chunk = c0 + (uint32_t)lane;
        nb_ = bv_u32x4{0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u};
        if (chunk < n_chunks) {
            nb_ = *reinterpret_cast<const bv_u32x4 *>(bs_row + (size_t)chunk * 16u);
            nq_ = *reinterpret_cast<const bv_u32x4 *>(q_row + (size_t)chunk * 16u);
            if (gid) ng_ = *reinterpret_cast<const bv_u32x4 *>(gid + (size_t)chunk * 16u);
        }
    };
    fetch(0);
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += BV_WAVE) {
        const uint32_t chunk = c0 + (uint32_t)lane;
        const bv_u32x4 vb = nb_, vq = nq_, vg = ng_;
        if (c0 + BV_WAVE < n_chunks) fetch(c0 + BV_WAVE);
        uint32_t mask = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t wbk = (k >> 2) == 0 ? vb.x : (k >> 2) == 1 ? vb.y : (k >> 2) == 2 ? vb.z : vb.w;
            const uint32_t wgk = (k >> 2) == 0 ? vg.x : (k >> 2) == 1 ? vg.y : (k >> 2) == 2 ? vg.z : vg.w;
            const uint32_t c = (wbk >> (8 * (k & 3))) & 0xFFu;
            const bool mine = !gid || ((wgk >> (8 * (k & 3))) & 0xFFu) == g;
            if (c < 8u && mine && chunk * 16u + (uint32_t)k < n_samples) mask |= 1u << k;
        }
        const uint32_t cnt = (uint32_t)__popc(mask);
        const uint32_t incl = bv_wave_incl_scan_u32(cnt, lane);
        uint32_t pos = count + incl - cnt;
        while (mask) {
            const int k = __builtin_ctz(mask);
            mask &= mask - 1u;
            if (pos < (uint32_t)BV_ORD_MAX) {
                const int wi = k >> 2, sh8 = 8 * (k & 3);
                const uint32_t wbk = wi == 0 ? vb.x : wi == 1 ? vb.y : wi == 2 ? vb.z : vb.w;
                const uint32_t wqk = wi == 0 ? vq.x : wi == 1 ? vq.y : wi == 2 ? vq.z : vq.w;
                ord[pos] = (uint16_t)((((wbk >> sh8) & 0xFFu) << 8) | ((wqk >> sh8) & 0xFFu));
            }
            ++pos;
        }
        count += (uint32_t)bv_readlane63_i32((int)incl);
    }
    return count;
}

// EM, algorithm.h:210-255, literally, on n <= BV_ORD_MAX samples (lane i = sample i).  f: initial freqs in, final out.
// hipcc contracts a * b + c into fused multiply-adds (-ffp-contract=fast: across statements, pragmas ignored) and
// __dmul_rn / __dadd_rn are plain * and + in its headers; the reference's marginal (algorithm.h:164-165) is a sum of
// ROUNDED products, so every product passes through an empty asm statement before it is added.
__device__ __noinline__ int bv_em_ordered(const uint16_t *ord, int n, const double *hit, const double *miss, double f[4],
                                          double *lr_out, int lane, const double *hostlog) {
    const double epsilon = (double)0.001f;
    const bool have = lane < n;
    const uint32_t w = have ? ord[lane] : 0u;
    const uint32_t b = (w >> 8) & 3u, qi = min(w & 0xFFu, 127u);
    const double hv = hit[qi], mv = miss[qi];
    const double lh0 = b == 0 ? hv : mv, lh1 = b == 1 ? hv : mv, lh2 = b == 2 ? hv : mv, lh3 = b == 3 ? hv : mv;
    double p0 = 0., p1 = 0., p2 = 0., p3 = 0., marg = 1.;
    auto e_step = [&]() {  // algorithm.h:161-171
        double L0 = __dmul_rn(lh0, f[0]), L1 = __dmul_rn(lh1, f[1]), L2 = __dmul_rn(lh2, f[2]), L3 = __dmul_rn(lh3, f[3]);
        asm volatile("" : "+v"(L0), "+v"(L1), "+v"(L2), "+v"(L3));  // rounded products: not to be fused into the sum
        marg = __dadd_rn(__dadd_rn(__dadd_rn(L0, L1), L2), L3);
        p0 = __ddiv_rn(L0, marg); p1 = __ddiv_rn(L1, marg); p2 = __ddiv_rn(L2, marg); p3 = __ddiv_rn(L3, marg);
    };
    // Sums over the samples IN SAMPLE ORDER, from 0.0 (algorithm.h:190-193, :29-33), of up to BV_SEQ_N quantities at once:
    // every lane parks its values in LDS, then lane k alone adds quantity k of samples 0 .. n-1 one after the other (the
    // loads do not depend on the running sum, so they pipeline; the k chains run side by side).  A loop of v_readlane +
    // add per quantity took five times the instructions, all of them on the critical path.
    typedef __attribute__((address_space(3))) double bv_lds_f64;
    bv_lds_f64 *seq = (bv_lds_f64 *)reinterpret_cast<double *>(const_cast<uint16_t *>(ord) + BV_ORD_MAX);  // `ord` is LDS
    double sums[BV_SEQ_N];
    auto wave_fence = [&]() {  // LDS operations of one wave execute in order: only the compiler has to be held back
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto seq_sums = [&](const double (&v)[BV_SEQ_N], int nq) {
#pragma unroll
        for (int k = 0; k < BV_SEQ_N; ++k)
            if (k < nq) seq[k * BV_SEQ_STRIDE + lane] = v[k];
        wave_fence();
        double acc = 0.;
        if (lane < nq) {
            const bv_lds_f64 *row = seq + lane * BV_SEQ_STRIDE;
            for (int i = 0; i < n; ++i) acc = __dadd_rn(acc, row[i]);
        }
#pragma unroll
        for (int k = 0; k < BV_SEQ_N; ++k) sums[k] = (k < nq) ? bv_readlane_f64(acc, k) : 0.;
        wave_fence();
    };
    const double dn = (double)n;
    auto ln = [&](double v) { return hostlog ? bv_log_host(v, hostlog) : log(v); };  // algorithm.h:243
    e_step();
    double llh = ln(marg);
    {
        const double v[BV_SEQ_N] = {p0, p1, p2, p3, 0.};
        seq_sums(v, 4);  // m_step, algorithm.h:184-198
        f[0] = __ddiv_rn(sums[0], dn); f[1] = __ddiv_rn(sums[1], dn); f[2] = __ddiv_rn(sums[2], dn); f[3] = __ddiv_rn(sums[3], dn);
    }
    int iters = 0;
    for (int it = 0; it < 100; ++it) {
        e_step();
        const double now = ln(marg);
        const double d = have ? bv_int_abs_trunc(now - llh) : 0.;
        llh = now;
        const double v[BV_SEQ_N] = {p0, p1, p2, p3, d};
        seq_sums(v, 5);  // the m_step's four sums and the convergence term, one walk
        f[0] = __ddiv_rn(sums[0], dn); f[1] = __ddiv_rn(sums[1], dn); f[2] = __ddiv_rn(sums[2], dn); f[3] = __ddiv_rn(sums[3], dn);
        ++iters;
        if (sums[4] < epsilon) break;
    }
    {
        const double v[BV_SEQ_N] = {have ? llh : 0., 0., 0., 0., 0.};
        seq_sums(v, 1);
    }
    *lr_out = sums[0];
    return iters;
}

// ------------------------------------------------------------------ LRT
// BaseType::lrt + _f, src/basetype.cpp:105-199.  The EM runs of one level (the n-subsets of
// the current active set, Combinations order: external/combinations.h:55-69) are independent.
//   NW >= 1: "team mode" -- the runs of a level are dealt to NW cooperating waves of a workgroup (NOT necessarily all of
//            its waves: no s_barrier); results meet in LDS behind a counting barrier in `sh` (which must then be a
//            BvLrtTeamShared shared by the team), every wave replays the cheap, uniform argmin / threshold decision.
//            Each run is the same one-wave arithmetic as in wave mode, so the outcome is bit-identical to it.
//            (Used at the tail of a long-row launch, bv_pass1.hip: the idle tally waves join the solver wave.)
//   NW == 0: "wave mode"  -- the calling wave does every run itself (used for the pop-group
//            calls of pass 2, where each wave owns a different group); `sh` is wave-private.
struct BvLrtShared {
    double f[2][4][4];  // [level parity][combination][base]
    double lr[2][4];
    int iters[2][4];
    double single[4];   // closed-form log-likelihoods of the four single-base subsets (wave mode; kept here, not in
                        // registers: the solver runs at the 128-VGPR limit)
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
    bool tie_risk;   // bv_lrt_g16<true> only: two subsets of one level scored within BV_TIE_TOL of each other (see there)
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
    static_assert(NW == 0, "team mode synchronises through bv_team_barrier");
    // same wave writes (lane 0) then reads (all lanes): LDS is in-order per wave; just
    // stop the compiler from moving the accesses across this point
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Team mode: the LRT scratch plus a counting barrier for the NW waves of the team.  `bar` counts arrivals since the
// team's owner last zeroed it (before it hands out a site); every spin is bounded like the kernel's other hand-offs.
struct BvLrtTeamShared {
    BvLrtShared lrt;  // first member: bv_lrt() receives &team->lrt
    uint32_t bar;
    uint32_t *err;    // a.counters + BV_CTR_TIMEOUT
};
__device__ __forceinline__ void bv_team_barrier(BvLrtShared *sh, uint32_t target, int lane) {
    BvLrtTeamShared *t = reinterpret_cast<BvLrtTeamShared *>(sh);
    // one arrival per wave; its earlier LDS writes are ahead of the atomic in the wave's in-order LDS queue
    if (lane == 0) __hip_atomic_fetch_add(&t->bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t spins = 0;
    while (__hip_atomic_load(&t->bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 24)) {
            atomicOr(t->err, 0x10000000u);  // BV_TMO_RING (bv_kernels.h)
            break;
        }
    }
}

// `specific_packed`: the candidate bases in reference order, 3 bits each (value 4 = a base
// that is not ACGT, never active); `nspec` of them.
template <int NW>
__device__ inline void bv_lrt(const BvBins &B, const uint32_t depth[4], uint32_t total, int specific_packed,
                              int nspec, int ref_code, double min_af, BvLrtShared *sh, int wave, int lane,
                              BvLrtOut &o, uint32_t q0_mask = 0xFu) {
    o.n_alt = 0; o.alt_packed = 0; o.af[0] = o.af[1] = o.af[2] = o.af[3] = 0.;
    o.m = 0; o.first = 0; o.chi2 = 0.; o.em_iters = 0; o.n_em = 0; o.zero_freq = false; o.tie_risk = false;
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
    if (m == 1 && !((q0_mask >> (act & 3)) & 1u)) {
        // One active base b and none of its calls has phred 0.  The reference's EM is then exact
        // and needs no arithmetic: every row's posterior for b is L_b / L_b == 1.0, the m_step
        // gives f_b == n/n == 1.0, the first delta is |int(-log(f_init))| == 0 because
        // f_init > 1 - 3*min_af, so the loop stops after one iteration; there is no smaller subset
        // to test (chi stays 0, basetype.cpp:146-151) and the log-likelihood sum is never read.
        const int b = act & 3;
        o.m = 1; o.first = b; o.chi2 = 0.; o.em_iters = 1; o.n_em = 1;
        if (b != ref_code) { o.n_alt = 1; o.alt_packed = b; o.af[0] = 1.0; }
        return;
    }
    const double n_cov = (double)total;
    const int m0 = m;
    const int first_c = (NW > 0) ? wave : 0, step_c = (NW > 0) ? NW : 1;
    // initial frequency of every base, depth/total (basetype.cpp:99)
    const double d0 = (double)depth[0] / n_cov, d1 = (double)depth[1] / n_cov, d2 = (double)depth[2] / n_cov,
                 d3 = (double)depth[3] / n_cov;

    double fr0 = 0., fr1 = 0., fr2 = 0., fr3 = 0.;  // active_bases_freq
    double lr_alt = 0., chi = 0.;
    bool have_single = false;  // sh->single[] holds the closed-form log-likelihoods of the single-base subsets
    int par = 0;
    uint32_t team_epoch = 0;  // team mode: barriers passed for this site
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
            int it;
            if (B.ord != nullptr && s != 0.) {
                // (copies for the call: bv_em_ordered is not inlined, and an array whose address leaves the function lives in
                // scratch memory -- with f and lr handed over themselves, EVERY subset of every site paid memory trips for them,
                // also in the kernels that never replay)
                double fo[4] = {f[0], f[1], f[2], f[3]}, lro;
                it = bv_em_ordered(B.ord, B.n_ord, B.hit, B.miss, fo, &lro, lane, bv_hostlog_of(B.logmiss));
                f[0] = fo[0]; f[1] = fo[1]; f[2] = fo[2]; f[3] = fo[3];
                lr = lro;
            } else if (B.loghit != nullptr && q0_mask == 0u && s != 0. && (in_set & (in_set - 1u)) == 0u) {
                // A single-base subset {b} with every likelihood positive: the reference's EM needs no arithmetic.  Its
                // first e_step gives every sample the posterior L/L == 1.0 for b, the m_step f_b == n/n == 1.0, and from
                // then on every marginal is the likelihood itself (lh * 1.0), so the reported log-likelihood is
                //     sum_i log(lh_i[b]) = sum_bins c * (bin's base == b ? log(1 - eps_q) : log(eps_q / 3))
                // with the host's log() of the host's table values -- the reference's own per-sample terms.  One sweep
                // yields the sums of all four bases: lr_b = sum c * logmiss + sum_{bins of b} c * (loghit - logmiss).
                // The loop runs twice when the start frequency is below 1/e (the first delta is |int(-log f_init)|
                // per sample, algorithm.h:245), else once.
                if (!have_single) {
                    double a_ = 0., g0 = 0., g1 = 0., g2 = 0., g3 = 0.;
                    const int nslots = (B.nb + BV_WAVE - 1) / BV_WAVE;
#pragma unroll
                    for (int sl = 0; sl < BV_SLOTS; ++sl) {
                        if (sl < nslots) {
                            const int i = sl * BV_WAVE + lane;
                            if (i < B.nb) {
                                const int at = bv_bin_at(B, i);
                                const uint32_t code = B.code[at];
                                const double cc = (double)B.cnt[at];
                                const double lm = B.logmiss[code & 127u], dh = B.loghit[code & 127u] - lm;
                                const uint32_t bb = code >> 7;
                                a_ += cc * lm;
                                g0 += (bb == 0) ? cc * dh : 0.;
                                g1 += (bb == 1) ? cc * dh : 0.;
                                g2 += (bb == 2) ? cc * dh : 0.;
                                g3 += (bb == 3) ? cc * dh : 0.;
                            }
                        }
                    }
                    a_ = bv_wave_sum(a_);
                    bv_wave_sum4(g0, g1, g2, g3, g0, g1, g2, g3);
                    // (team mode: every wave that meets a single-base subset computes the four sums itself and reads back
                    // its own stores -- the other waves write the same values -- so no team barrier is needed here)
                    if (lane == 0) {
                        sh->single[0] = a_ + g0; sh->single[1] = a_ + g1;
                        sh->single[2] = a_ + g2; sh->single[3] = a_ + g3;
                    }
                    bv_lrt_sync<0>();
                    have_single = true;
                }
                const int b1 = __builtin_ctz(in_set);
                lr = sh->single[b1];
                const double f_init = bv_sel4(f[0], f[1], f[2], f[3], b1);
                it = (f_init < 0.36787944117144233) ? 2 : 1;
                f[0] = (b1 == 0) ? 1.0 : 0.; f[1] = (b1 == 1) ? 1.0 : 0.;
                f[2] = (b1 == 2) ? 1.0 : 0.; f[3] = (b1 == 3) ? 1.0 : 0.;
            } else {
                // phred-0 calls (1 - eps == 0) and all-zero starts make 0/0 in the reference: exact replay
                it = (q0_mask != 0u || s == 0.) ? bv_em_wave_generic(B, f, n_cov, &lr, lane)
                                                : bv_em_wave(B, f, in_set, n_cov, &lr, lane);
            }
            if (lane == 0) {
                sh->f[par][c][0] = f[0]; sh->f[par][c][1] = f[1];
                sh->f[par][c][2] = f[2]; sh->f[par][c][3] = f[3];
                sh->lr[par][c] = lr;
                sh->iters[par][c] = (s == 0.) ? -it - 1 : it;  // negative marks basetype.cpp:113-115
            }
        }
        if (NW > 0) bv_team_barrier(sh, (uint32_t)NW * (++team_epoch), lane);
        else bv_lrt_sync<0>();
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
