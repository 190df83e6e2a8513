// bv_tally.h -- the per-cell tally primitives shared by the streaming kernels (bv_pass1.hip, bv_pass1_short.hip).
#pragma once

#include "bv_kernels.h"

// ------------------------------------------------------------------------------ tally
// One cell = one byte of each plane.  v_perm_b32 glues the call byte and the phred byte into
// X = call << 8 | phred; covered calls are 0..7, so X < 0x800 is the coverage test and X
// is directly the histogram word index: 3 VALU + 1 ds_add_u32 per cell, and no phred byte can
// index outside its 256-wide row.
template <int J>
__device__ __forceinline__ uint32_t bv_cell_index(uint32_t w, uint32_t qq) {
    // selector bytes: result byte0 = qq.byte[J] (S1 is bytes 0-3), byte1 = w.byte[J] (S0 is 4-7)
    constexpr uint32_t SEL = 0x0C0C0000u | ((4u + J) << 8) | (uint32_t)J;
    return __builtin_amdgcn_perm(w, qq, SEL);
}
typedef __attribute__((address_space(3))) uint32_t bv_lds_u32;
// 16 predicated LDS increments: word (x[j] << SH) / 4 of `hist` gets +1 for every lane with x[j] < lim.  All 16 word addresses
// are formed first, unpredicated (independent VALU work that pipelines), the 16 compares go into 16 SGPR pairs, and only then
// come the 16 x (s_and_b64 exec, ds_add_u32): pure SALU + LDS issue, no VALU in the chain, no branches.  Hand-scheduled because
// the compiler either re-serialises the chains through VCC or branches around every add.  (LDS operations of one wave execute
// in order, so reads that follow need no extra wait.)
template <int SH>
__device__ __forceinline__ void bv_lds_add16(const uint32_t x[16], uint32_t *hist, uint32_t one, uint32_t lim) {
    const uint32_t hbase = (uint32_t)(uintptr_t)(bv_lds_u32 *)hist;
    uint32_t ad[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) ad[j] = hbase + (x[j] << SH);
    unsigned long long m[16], sv;
    asm volatile(
        "v_cmp_gt_u32_e64 %[m0], %[lim], %[x0]\n\t"
        "v_cmp_gt_u32_e64 %[m1], %[lim], %[x1]\n\t"
        "v_cmp_gt_u32_e64 %[m2], %[lim], %[x2]\n\t"
        "v_cmp_gt_u32_e64 %[m3], %[lim], %[x3]\n\t"
        "v_cmp_gt_u32_e64 %[m4], %[lim], %[x4]\n\t"
        "v_cmp_gt_u32_e64 %[m5], %[lim], %[x5]\n\t"
        "v_cmp_gt_u32_e64 %[m6], %[lim], %[x6]\n\t"
        "v_cmp_gt_u32_e64 %[m7], %[lim], %[x7]\n\t"
        "v_cmp_gt_u32_e64 %[m8], %[lim], %[x8]\n\t"
        "v_cmp_gt_u32_e64 %[m9], %[lim], %[x9]\n\t"
        "v_cmp_gt_u32_e64 %[m10], %[lim], %[x10]\n\t"
        "v_cmp_gt_u32_e64 %[m11], %[lim], %[x11]\n\t"
        "v_cmp_gt_u32_e64 %[m12], %[lim], %[x12]\n\t"
        "v_cmp_gt_u32_e64 %[m13], %[lim], %[x13]\n\t"
        "v_cmp_gt_u32_e64 %[m14], %[lim], %[x14]\n\t"
        "v_cmp_gt_u32_e64 %[m15], %[lim], %[x15]\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_and_b64 exec, %[sv], %[m0]\n\tds_add_u32 %[a0], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m1]\n\tds_add_u32 %[a1], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m2]\n\tds_add_u32 %[a2], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m3]\n\tds_add_u32 %[a3], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m4]\n\tds_add_u32 %[a4], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m5]\n\tds_add_u32 %[a5], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m6]\n\tds_add_u32 %[a6], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m7]\n\tds_add_u32 %[a7], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m8]\n\tds_add_u32 %[a8], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m9]\n\tds_add_u32 %[a9], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m10]\n\tds_add_u32 %[a10], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m11]\n\tds_add_u32 %[a11], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m12]\n\tds_add_u32 %[a12], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m13]\n\tds_add_u32 %[a13], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m14]\n\tds_add_u32 %[a14], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m15]\n\tds_add_u32 %[a15], %[one]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [m0] "=&s"(m[0]), [m1] "=&s"(m[1]), [m2] "=&s"(m[2]), [m3] "=&s"(m[3]), [m4] "=&s"(m[4]), [m5] "=&s"(m[5]), [m6] "=&s"(m[6]), [m7] "=&s"(m[7]), [m8] "=&s"(m[8]), [m9] "=&s"(m[9]), [m10] "=&s"(m[10]), [m11] "=&s"(m[11]), [m12] "=&s"(m[12]), [m13] "=&s"(m[13]), [m14] "=&s"(m[14]), [m15] "=&s"(m[15]), [sv] "=&s"(sv)
        : [x0] "v"(x[0]), [a0] "v"(ad[0]), [x1] "v"(x[1]), [a1] "v"(ad[1]), [x2] "v"(x[2]), [a2] "v"(ad[2]), [x3] "v"(x[3]), [a3] "v"(ad[3]), [x4] "v"(x[4]), [a4] "v"(ad[4]), [x5] "v"(x[5]), [a5] "v"(ad[5]), [x6] "v"(x[6]), [a6] "v"(ad[6]), [x7] "v"(x[7]), [a7] "v"(ad[7]), [x8] "v"(x[8]), [a8] "v"(ad[8]), [x9] "v"(x[9]), [a9] "v"(ad[9]), [x10] "v"(x[10]), [a10] "v"(ad[10]), [x11] "v"(x[11]), [a11] "v"(ad[11]), [x12] "v"(x[12]), [a12] "v"(ad[12]), [x13] "v"(x[13]), [a13] "v"(ad[13]), [x14] "v"(x[14]), [a14] "v"(ad[14]), [x15] "v"(x[15]), [a15] "v"(ad[15]), [one] "v"(one), [lim] "s"(lim)
        : "memory", "scc");
}


// Two tallies under ONE predicate: cell j adds to hist_a at x[j] and to hist_b at y[j] when x[j] < lim.  For pairs whose predicates are
// the same cell by cell -- the mapq and the read-position-rank histograms of pass 2: both X = class << 8 | value with the same class
// byte, "< 0x200" <=> the cell is a REF or an ALT read -- : 16 v_cmp and 16 exec writes per 16 cells instead of 32 and 32 (the
// variant rows' tally is issue-bound: ~124 VALU + 40 SALU + 32 ds_add per 1,024 cells and wave).
template <int SH>
__device__ __forceinline__ void bv_lds_add16x2(const uint32_t x[16], const uint32_t y[16], uint32_t *hist_a, uint32_t *hist_b, uint32_t one, uint32_t lim) {
    const uint32_t abase = (uint32_t)(uintptr_t)(bv_lds_u32 *)hist_a, bbase = (uint32_t)(uintptr_t)(bv_lds_u32 *)hist_b;
    uint32_t ad[16], bd[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        ad[j] = abase + (x[j] << SH);
        bd[j] = bbase + (y[j] << SH);
    }
    unsigned long long m[16], sv;
    asm volatile(
        "v_cmp_gt_u32_e64 %[m0], %[lim], %[x0]\n\t"
        "v_cmp_gt_u32_e64 %[m1], %[lim], %[x1]\n\t"
        "v_cmp_gt_u32_e64 %[m2], %[lim], %[x2]\n\t"
        "v_cmp_gt_u32_e64 %[m3], %[lim], %[x3]\n\t"
        "v_cmp_gt_u32_e64 %[m4], %[lim], %[x4]\n\t"
        "v_cmp_gt_u32_e64 %[m5], %[lim], %[x5]\n\t"
        "v_cmp_gt_u32_e64 %[m6], %[lim], %[x6]\n\t"
        "v_cmp_gt_u32_e64 %[m7], %[lim], %[x7]\n\t"
        "v_cmp_gt_u32_e64 %[m8], %[lim], %[x8]\n\t"
        "v_cmp_gt_u32_e64 %[m9], %[lim], %[x9]\n\t"
        "v_cmp_gt_u32_e64 %[m10], %[lim], %[x10]\n\t"
        "v_cmp_gt_u32_e64 %[m11], %[lim], %[x11]\n\t"
        "v_cmp_gt_u32_e64 %[m12], %[lim], %[x12]\n\t"
        "v_cmp_gt_u32_e64 %[m13], %[lim], %[x13]\n\t"
        "v_cmp_gt_u32_e64 %[m14], %[lim], %[x14]\n\t"
        "v_cmp_gt_u32_e64 %[m15], %[lim], %[x15]\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_and_b64 exec, %[sv], %[m0]\n\tds_add_u32 %[a0], %[one]\n\tds_add_u32 %[b0], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m1]\n\tds_add_u32 %[a1], %[one]\n\tds_add_u32 %[b1], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m2]\n\tds_add_u32 %[a2], %[one]\n\tds_add_u32 %[b2], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m3]\n\tds_add_u32 %[a3], %[one]\n\tds_add_u32 %[b3], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m4]\n\tds_add_u32 %[a4], %[one]\n\tds_add_u32 %[b4], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m5]\n\tds_add_u32 %[a5], %[one]\n\tds_add_u32 %[b5], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m6]\n\tds_add_u32 %[a6], %[one]\n\tds_add_u32 %[b6], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m7]\n\tds_add_u32 %[a7], %[one]\n\tds_add_u32 %[b7], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m8]\n\tds_add_u32 %[a8], %[one]\n\tds_add_u32 %[b8], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m9]\n\tds_add_u32 %[a9], %[one]\n\tds_add_u32 %[b9], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m10]\n\tds_add_u32 %[a10], %[one]\n\tds_add_u32 %[b10], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m11]\n\tds_add_u32 %[a11], %[one]\n\tds_add_u32 %[b11], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m12]\n\tds_add_u32 %[a12], %[one]\n\tds_add_u32 %[b12], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m13]\n\tds_add_u32 %[a13], %[one]\n\tds_add_u32 %[b13], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m14]\n\tds_add_u32 %[a14], %[one]\n\tds_add_u32 %[b14], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m15]\n\tds_add_u32 %[a15], %[one]\n\tds_add_u32 %[b15], %[one]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [m0] "=&s"(m[0]), [m1] "=&s"(m[1]), [m2] "=&s"(m[2]), [m3] "=&s"(m[3]), [m4] "=&s"(m[4]), [m5] "=&s"(m[5]), [m6] "=&s"(m[6]), [m7] "=&s"(m[7]), [m8] "=&s"(m[8]), [m9] "=&s"(m[9]), [m10] "=&s"(m[10]), [m11] "=&s"(m[11]), [m12] "=&s"(m[12]), [m13] "=&s"(m[13]), [m14] "=&s"(m[14]), [m15] "=&s"(m[15]), [sv] "=&s"(sv)
        : [x0] "v"(x[0]), [a0] "v"(ad[0]), [b0] "v"(bd[0]), [x1] "v"(x[1]), [a1] "v"(ad[1]), [b1] "v"(bd[1]), [x2] "v"(x[2]), [a2] "v"(ad[2]), [b2] "v"(bd[2]), [x3] "v"(x[3]), [a3] "v"(ad[3]), [b3] "v"(bd[3]), [x4] "v"(x[4]), [a4] "v"(ad[4]), [b4] "v"(bd[4]), [x5] "v"(x[5]), [a5] "v"(ad[5]), [b5] "v"(bd[5]), [x6] "v"(x[6]), [a6] "v"(ad[6]), [b6] "v"(bd[6]), [x7] "v"(x[7]), [a7] "v"(ad[7]), [b7] "v"(bd[7]), [x8] "v"(x[8]), [a8] "v"(ad[8]), [b8] "v"(bd[8]), [x9] "v"(x[9]), [a9] "v"(ad[9]), [b9] "v"(bd[9]), [x10] "v"(x[10]), [a10] "v"(ad[10]), [b10] "v"(bd[10]), [x11] "v"(x[11]), [a11] "v"(ad[11]), [b11] "v"(bd[11]), [x12] "v"(x[12]), [a12] "v"(ad[12]), [b12] "v"(bd[12]), [x13] "v"(x[13]), [a13] "v"(ad[13]), [b13] "v"(bd[13]), [x14] "v"(x[14]), [a14] "v"(ad[14]), [b14] "v"(bd[14]), [x15] "v"(x[15]), [a15] "v"(ad[15]), [b15] "v"(bd[15]), [one] "v"(one), [lim] "s"(lim)
        : "memory", "scc");
}


// bv_lds_add16 for tallies with ONE dominant value (the mapq histogram of a deep row: 80 % of a cohort's reads carry the
// aligner's top mapping quality and most reads of a site are REF reads, so at full coverage ~50 of a wave's 64 lanes add to
// the same LDS word, and LDS atomics on one address are served one lane at a time -- measured: pass 2 of 100,000-sample
// rows 6 x as long at coverage 1.0 as at 0.08).  D = the value of slot 0's first lane that passes (wave-uniform): per slot the
// lanes that hold D are taken out of the predicate and counted (v_cmp_ne -> s_andn2 / s_bcnt1 / s_add: one VALU + four SALU),
// the others add as before, and ONE lane adds the chunk's count to D's word at the end.  Any D gives the same histogram; a
// lucky one (the dominant value: three times out of four) removes the serialisation.  `D` is the caller's, carried from chunk
// to chunk of a row (start a row with BV_DOM_NONE): a value that collected an eighth of a chunk's cells is kept, anything less
// is dropped and the next chunk picks again -- a row pays for an unlucky pick once, not in a quarter of its chunks.  Callers use it for rows whose depth says
// they are dense (a wave-uniform choice per row): sparse rows keep bv_lds_add16, whose instruction count this exceeds by a third.
#define BV_DOM_NONE 0xFFFFFFFFu
template <int SH>
__device__ __forceinline__ void bv_lds_add16_dom(const uint32_t x[16], uint32_t *hist, uint32_t one, uint32_t lim, uint32_t &D) {
    const uint32_t hbase = (uint32_t)(uintptr_t)(bv_lds_u32 *)hist;
    uint32_t ad[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) ad[j] = hbase + (x[j] << SH);
    if (D == BV_DOM_NONE) {
        // slot 0's first passing lane names D; no such lane: a D nothing equals
        const unsigned long long m0 = __ballot(x[0] < lim);
        if (m0 != 0ull) D = (uint32_t)__builtin_amdgcn_readlane((int)x[0], (int)__builtin_ctzll(m0));
    }
    const uint32_t adD = hbase + ((D == BV_DOM_NONE ? 0u : D) << SH);
    unsigned long long m[16], n[4], t, sv;
    uint32_t cnt, c, va, vc;
    asm volatile(
        "v_cmp_gt_u32_e64 %[m0], %[lim], %[x0]\n\t"
        "v_cmp_gt_u32_e64 %[m1], %[lim], %[x1]\n\t"
        "v_cmp_gt_u32_e64 %[m2], %[lim], %[x2]\n\t"
        "v_cmp_gt_u32_e64 %[m3], %[lim], %[x3]\n\t"
        "v_cmp_gt_u32_e64 %[m4], %[lim], %[x4]\n\t"
        "v_cmp_gt_u32_e64 %[m5], %[lim], %[x5]\n\t"
        "v_cmp_gt_u32_e64 %[m6], %[lim], %[x6]\n\t"
        "v_cmp_gt_u32_e64 %[m7], %[lim], %[x7]\n\t"
        "v_cmp_gt_u32_e64 %[m8], %[lim], %[x8]\n\t"
        "v_cmp_gt_u32_e64 %[m9], %[lim], %[x9]\n\t"
        "v_cmp_gt_u32_e64 %[m10], %[lim], %[x10]\n\t"
        "v_cmp_gt_u32_e64 %[m11], %[lim], %[x11]\n\t"
        "v_cmp_gt_u32_e64 %[m12], %[lim], %[x12]\n\t"
        "v_cmp_gt_u32_e64 %[m13], %[lim], %[x13]\n\t"
        "v_cmp_gt_u32_e64 %[m14], %[lim], %[x14]\n\t"
        "v_cmp_gt_u32_e64 %[m15], %[lim], %[x15]\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b32 %[cnt], 0\n\t"
#define BV_DOM4(a, b, c_, d)                                                                                                     \
        "v_cmp_ne_u32_e64 %[n0], %[D], %[x" #a "]\n\t"                                                                           \
        "v_cmp_ne_u32_e64 %[n1], %[D], %[x" #b "]\n\t"                                                                           \
        "v_cmp_ne_u32_e64 %[n2], %[D], %[x" #c_ "]\n\t"                                                                          \
        "v_cmp_ne_u32_e64 %[n3], %[D], %[x" #d "]\n\t"                                                                           \
        "s_and_b64 %[m" #a "], %[m" #a "], %[n0]\n\ts_andn2_b64 %[t], %[sv], %[n0]\n\ts_bcnt1_i32_b64 %[c], %[t]\n\ts_add_u32 %[cnt], %[cnt], %[c]\n\t"   \
        "s_and_b64 %[m" #b "], %[m" #b "], %[n1]\n\ts_andn2_b64 %[t], %[sv], %[n1]\n\ts_bcnt1_i32_b64 %[c], %[t]\n\ts_add_u32 %[cnt], %[cnt], %[c]\n\t"   \
        "s_and_b64 %[m" #c_ "], %[m" #c_ "], %[n2]\n\ts_andn2_b64 %[t], %[sv], %[n2]\n\ts_bcnt1_i32_b64 %[c], %[t]\n\ts_add_u32 %[cnt], %[cnt], %[c]\n\t" \
        "s_and_b64 %[m" #d "], %[m" #d "], %[n3]\n\ts_andn2_b64 %[t], %[sv], %[n3]\n\ts_bcnt1_i32_b64 %[c], %[t]\n\ts_add_u32 %[cnt], %[cnt], %[c]\n\t"
        BV_DOM4(0, 1, 2, 3)
        BV_DOM4(4, 5, 6, 7)
        BV_DOM4(8, 9, 10, 11)
        BV_DOM4(12, 13, 14, 15)
#undef BV_DOM4
        "s_and_b64 exec, %[sv], %[m0]\n\tds_add_u32 %[a0], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m1]\n\tds_add_u32 %[a1], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m2]\n\tds_add_u32 %[a2], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m3]\n\tds_add_u32 %[a3], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m4]\n\tds_add_u32 %[a4], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m5]\n\tds_add_u32 %[a5], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m6]\n\tds_add_u32 %[a6], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m7]\n\tds_add_u32 %[a7], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m8]\n\tds_add_u32 %[a8], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m9]\n\tds_add_u32 %[a9], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m10]\n\tds_add_u32 %[a10], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m11]\n\tds_add_u32 %[a11], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m12]\n\tds_add_u32 %[a12], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m13]\n\tds_add_u32 %[a13], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m14]\n\tds_add_u32 %[a14], %[one]\n\t"
        "s_and_b64 exec, %[sv], %[m15]\n\tds_add_u32 %[a15], %[one]\n\t"
        // the chunk's count of D, by the wave's first active lane
        "s_ff1_i32_b64 %[c], %[sv]\n\t"
        "s_lshl_b64 %[t], 1, %[c]\n\t"
        "s_mov_b64 exec, %[t]\n\t"
        "v_mov_b32 %[va], %[adD]\n\t"
        "v_mov_b32 %[vc], %[cnt]\n\t"
        "ds_add_u32 %[va], %[vc]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [m0] "=&s"(m[0]), [m1] "=&s"(m[1]), [m2] "=&s"(m[2]), [m3] "=&s"(m[3]), [m4] "=&s"(m[4]), [m5] "=&s"(m[5]), [m6] "=&s"(m[6]), [m7] "=&s"(m[7]), [m8] "=&s"(m[8]), [m9] "=&s"(m[9]), [m10] "=&s"(m[10]), [m11] "=&s"(m[11]), [m12] "=&s"(m[12]), [m13] "=&s"(m[13]), [m14] "=&s"(m[14]), [m15] "=&s"(m[15]),
          [n0] "=&s"(n[0]), [n1] "=&s"(n[1]), [n2] "=&s"(n[2]), [n3] "=&s"(n[3]), [t] "=&s"(t), [sv] "=&s"(sv), [cnt] "=&s"(cnt), [c] "=&s"(c), [va] "=&v"(va), [vc] "=&v"(vc)
        : [x0] "v"(x[0]), [a0] "v"(ad[0]), [x1] "v"(x[1]), [a1] "v"(ad[1]), [x2] "v"(x[2]), [a2] "v"(ad[2]), [x3] "v"(x[3]), [a3] "v"(ad[3]), [x4] "v"(x[4]), [a4] "v"(ad[4]), [x5] "v"(x[5]), [a5] "v"(ad[5]), [x6] "v"(x[6]), [a6] "v"(ad[6]), [x7] "v"(x[7]), [a7] "v"(ad[7]), [x8] "v"(x[8]), [a8] "v"(ad[8]), [x9] "v"(x[9]), [a9] "v"(ad[9]), [x10] "v"(x[10]), [a10] "v"(ad[10]), [x11] "v"(x[11]), [a11] "v"(ad[11]), [x12] "v"(x[12]), [a12] "v"(ad[12]), [x13] "v"(x[13]), [a13] "v"(ad[13]), [x14] "v"(x[14]), [a14] "v"(ad[14]), [x15] "v"(x[15]), [a15] "v"(ad[15]),
          [one] "v"(one), [lim] "s"(lim), [D] "s"(D), [adD] "s"(adD)
        : "memory", "scc");
    if (cnt < 128u) D = BV_DOM_NONE;  // (of the chunk's 1,024 cells)
}

// The same into 16-bit counters (two per word): X = x[j] is the BYTE offset of the cell's half-word; the word X & ~3 gets
// 1 << 16 * ((X >> 1) & 1).  For rows of at most 65,535 cells, where no count can carry into its neighbour.
__device__ __forceinline__ void bv_lds_add16_half(const uint32_t x[16], uint32_t *hist, uint32_t one, uint32_t lim) {
    const uint32_t hbase = (uint32_t)(uintptr_t)(bv_lds_u32 *)hist;
    uint32_t ad[16], val[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        ad[j] = hbase + (x[j] & 0xFFFCu);
        val[j] = one << ((x[j] << 3) & 31u);  // x is even: bit 1 = the index's low bit -> shift 0 or 16
    }
    unsigned long long m[16], sv;
    asm volatile(
        "v_cmp_gt_u32_e64 %[m0], %[lim], %[x0]\n\t"
        "v_cmp_gt_u32_e64 %[m1], %[lim], %[x1]\n\t"
        "v_cmp_gt_u32_e64 %[m2], %[lim], %[x2]\n\t"
        "v_cmp_gt_u32_e64 %[m3], %[lim], %[x3]\n\t"
        "v_cmp_gt_u32_e64 %[m4], %[lim], %[x4]\n\t"
        "v_cmp_gt_u32_e64 %[m5], %[lim], %[x5]\n\t"
        "v_cmp_gt_u32_e64 %[m6], %[lim], %[x6]\n\t"
        "v_cmp_gt_u32_e64 %[m7], %[lim], %[x7]\n\t"
        "v_cmp_gt_u32_e64 %[m8], %[lim], %[x8]\n\t"
        "v_cmp_gt_u32_e64 %[m9], %[lim], %[x9]\n\t"
        "v_cmp_gt_u32_e64 %[m10], %[lim], %[x10]\n\t"
        "v_cmp_gt_u32_e64 %[m11], %[lim], %[x11]\n\t"
        "v_cmp_gt_u32_e64 %[m12], %[lim], %[x12]\n\t"
        "v_cmp_gt_u32_e64 %[m13], %[lim], %[x13]\n\t"
        "v_cmp_gt_u32_e64 %[m14], %[lim], %[x14]\n\t"
        "v_cmp_gt_u32_e64 %[m15], %[lim], %[x15]\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_and_b64 exec, %[sv], %[m0]\n\tds_add_u32 %[a0], %[v0]\n\t"
        "s_and_b64 exec, %[sv], %[m1]\n\tds_add_u32 %[a1], %[v1]\n\t"
        "s_and_b64 exec, %[sv], %[m2]\n\tds_add_u32 %[a2], %[v2]\n\t"
        "s_and_b64 exec, %[sv], %[m3]\n\tds_add_u32 %[a3], %[v3]\n\t"
        "s_and_b64 exec, %[sv], %[m4]\n\tds_add_u32 %[a4], %[v4]\n\t"
        "s_and_b64 exec, %[sv], %[m5]\n\tds_add_u32 %[a5], %[v5]\n\t"
        "s_and_b64 exec, %[sv], %[m6]\n\tds_add_u32 %[a6], %[v6]\n\t"
        "s_and_b64 exec, %[sv], %[m7]\n\tds_add_u32 %[a7], %[v7]\n\t"
        "s_and_b64 exec, %[sv], %[m8]\n\tds_add_u32 %[a8], %[v8]\n\t"
        "s_and_b64 exec, %[sv], %[m9]\n\tds_add_u32 %[a9], %[v9]\n\t"
        "s_and_b64 exec, %[sv], %[m10]\n\tds_add_u32 %[a10], %[v10]\n\t"
        "s_and_b64 exec, %[sv], %[m11]\n\tds_add_u32 %[a11], %[v11]\n\t"
        "s_and_b64 exec, %[sv], %[m12]\n\tds_add_u32 %[a12], %[v12]\n\t"
        "s_and_b64 exec, %[sv], %[m13]\n\tds_add_u32 %[a13], %[v13]\n\t"
        "s_and_b64 exec, %[sv], %[m14]\n\tds_add_u32 %[a14], %[v14]\n\t"
        "s_and_b64 exec, %[sv], %[m15]\n\tds_add_u32 %[a15], %[v15]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [m0] "=&s"(m[0]), [m1] "=&s"(m[1]), [m2] "=&s"(m[2]), [m3] "=&s"(m[3]), [m4] "=&s"(m[4]), [m5] "=&s"(m[5]), [m6] "=&s"(m[6]), [m7] "=&s"(m[7]), [m8] "=&s"(m[8]), [m9] "=&s"(m[9]), [m10] "=&s"(m[10]), [m11] "=&s"(m[11]), [m12] "=&s"(m[12]), [m13] "=&s"(m[13]), [m14] "=&s"(m[14]), [m15] "=&s"(m[15]), [sv] "=&s"(sv)
        : [x0] "v"(x[0]), [a0] "v"(ad[0]), [v0] "v"(val[0]), [x1] "v"(x[1]), [a1] "v"(ad[1]), [v1] "v"(val[1]), [x2] "v"(x[2]), [a2] "v"(ad[2]), [v2] "v"(val[2]), [x3] "v"(x[3]), [a3] "v"(ad[3]), [v3] "v"(val[3]), [x4] "v"(x[4]), [a4] "v"(ad[4]), [v4] "v"(val[4]), [x5] "v"(x[5]), [a5] "v"(ad[5]), [v5] "v"(val[5]), [x6] "v"(x[6]), [a6] "v"(ad[6]), [v6] "v"(val[6]), [x7] "v"(x[7]), [a7] "v"(ad[7]), [v7] "v"(val[7]), [x8] "v"(x[8]), [a8] "v"(ad[8]), [v8] "v"(val[8]), [x9] "v"(x[9]), [a9] "v"(ad[9]), [v9] "v"(val[9]), [x10] "v"(x[10]), [a10] "v"(ad[10]), [v10] "v"(val[10]), [x11] "v"(x[11]), [a11] "v"(ad[11]), [v11] "v"(val[11]), [x12] "v"(x[12]), [a12] "v"(ad[12]), [v12] "v"(val[12]), [x13] "v"(x[13]), [a13] "v"(ad[13]), [v13] "v"(val[13]), [x14] "v"(x[14]), [a14] "v"(ad[14]), [v14] "v"(val[14]), [x15] "v"(x[15]), [a15] "v"(ad[15]), [v15] "v"(val[15]), [lim] "s"(lim)
        : "memory", "scc");
}

// One 16-byte chunk = 16 cells of this lane: a per-cell perm -> cmp -> exec -> address -> ds_add chain cannot overlap with
// its neighbours because every link goes through VCC / EXEC, hence the batched form of bv_lds_add16.
// SH: log2 of the byte stride of index X (2: words indexed by X -- the 8 x 256 histogram; 1: X is twice the word index --
// the 8 x 128 histogram of bv_pass1_short.hip, whose phred bytes arrive pre-shifted by one bit).
// SWZ (rows the long-row kernel takes for dense -- a quarter of the cells covered): the phred byte is XORed with the cell's
// (strand, base) << 2 first.  A 4-byte LDS access is served in two groups of 32 lanes over 32 banks (bank = word index mod 32 =
// phred mod 32 here, whatever the histogram row), and a dense row puts most of a group's lanes on the twenty-odd phreds a
// sequencer emits: ~6 lanes on the busiest bank, six LDS cycles per group where a sparse row's handful of active lanes takes one.
// With the swizzle the forward and the reverse strand (and the bases) of one phred fall on different banks: 100,000-sample rows
// at full coverage, pass 1 0.59 -> 0.67 of the HBM peak.  (Bits 3-5, the first try, changed nothing: bit 5 is no bank bit.)
// The permutation stays inside the cell's histogram row for any byte; bv_hist_unswizzle puts the row back before anybody reads
// it.  Three VALU per four cells: sparse rows do not pay it.
template <int SH = 2, bool SWZ = false>
__device__ __forceinline__ void bv_tally_chunk(const bv_u32x4 &vb, const bv_u32x4 &vq_in, uint32_t *hist, uint32_t one) {
    bv_u32x4 vq = vq_in;
    if (SWZ) {
        constexpr uint32_t M = 0x1C1C1C1Cu << (2 - SH);  // SH 1: the phred bytes arrive shifted left by one
        vq.x ^= (vb.x << (2 + 2 - SH)) & M; vq.y ^= (vb.y << (2 + 2 - SH)) & M; vq.z ^= (vb.z << (2 + 2 - SH)) & M; vq.w ^= (vb.w << (2 + 2 - SH)) & M;
    }
    uint32_t x[16];
    x[0] = bv_cell_index<0>(vb.x, vq.x); x[1] = bv_cell_index<1>(vb.x, vq.x);
    x[2] = bv_cell_index<2>(vb.x, vq.x); x[3] = bv_cell_index<3>(vb.x, vq.x);
    x[4] = bv_cell_index<0>(vb.y, vq.y); x[5] = bv_cell_index<1>(vb.y, vq.y);
    x[6] = bv_cell_index<2>(vb.y, vq.y); x[7] = bv_cell_index<3>(vb.y, vq.y);
    x[8] = bv_cell_index<0>(vb.z, vq.z); x[9] = bv_cell_index<1>(vb.z, vq.z);
    x[10] = bv_cell_index<2>(vb.z, vq.z); x[11] = bv_cell_index<3>(vb.z, vq.z);
    x[12] = bv_cell_index<0>(vb.w, vq.w); x[13] = bv_cell_index<1>(vb.w, vq.w);
    x[14] = bv_cell_index<2>(vb.w, vq.w); x[15] = bv_cell_index<3>(vb.w, vq.w);
    // (Tried, round 6, for dense rows of binned qualities: counting the wave's TWO dominant values per chunk here too
    // (bv_lds_add16_dom with two values).  Rows without such values -- every row of an un-binned sequencer -- lost 12-40 % of pass 1
    // to it, even when the search was given up after two chunks: its registers alone cost the dense instance 12 %.  Not kept.)
    bv_lds_add16<SH>(x, hist, one, 0x800u);
}
// A histogram tallied with SWZ back in its plain order, in place, by one wave: `rows` rows of `1 << LOGW` words (column c of
// row r holds phred c ^ ((r & 7) << 3)).  Every lane reads its words, then writes them where they belong.
template <int LOGW, int ROWS>
__device__ __forceinline__ void bv_hist_unswizzle(uint32_t *hist, int lane) {
    constexpr int PER = (ROWS << LOGW) / 64;
    uint32_t v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = hist[i * 64 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t idx = (uint32_t)(i * 64 + lane), row = idx >> LOGW;
        hist[idx ^ ((row & 7u) << 2)] = v[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// cells at or beyond n_samples in the row's last chunk are forced to 'N'
__device__ __forceinline__ uint32_t bv_mask_tail_dword(uint32_t w, int keep) {
    if (keep >= 4) return w;
    if (keep <= 0) return 0x08080808u;
    uint32_t low = (1u << (8 * keep)) - 1u;
    return (w & low) | (0x08080808u & ~low);
}

// ------------------------------------------------------------------------------ LDS-DMA
// 64 lanes x 16 bytes from base + voff into LDS at lds_dst + lane * 16 (M0 saved / restored); counted on vmcnt like any load,
// invisible to the compiler's own wait insertion: the callers count their waits by hand.
__device__ __forceinline__ void bv_glds16(uint32_t lds_dst, const uint8_t *base, uint32_t voff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_dst), "s"(base)
                 : "memory");
}
// One wave-level fetch-and-add on an LDS word, result in an SGPR.  Done by lane 0 alone inside the asm statement, so the
// compiler sees no divergent branch (an `if (lane == 0)` around an atomic inside the streaming kernel's prefetch step made
// it keep the whole ring state in VGPRs) and no vector-memory operation (lgkmcnt only).
__device__ __forceinline__ uint32_t bv_lds_fetch_add_wave(uint32_t lds_addr, uint32_t v) {
    uint32_t r, t;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "v_mov_b32 %[t], %[val]\n\t"
        "ds_add_rtn_u32 %[t], %[adr], %[t]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %[r], %[t]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [r] "=&s"(r), [t] "=&v"(t), [sv] "=&s"(sv)
        : [adr] "v"(lds_addr), [val] "s"(v)
        : "memory");
    return r;
}
// a pointer the compiler knows to be wave-uniform (the "s" operand above)
__device__ __forceinline__ const uint8_t *bv_uniform_ptr(const uint8_t *p) {
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (const uint8_t *)(uintptr_t)(((uint64_t)hi << 32) | lo);
}

// ---- the perm form of the rank-sum tally (see bv_pass2_dma_kernel, bv_pass2.hip, for the derivation; also used by bv_pass1_fused.hip): class byte J << 8 | mapq byte J, and
// rank_hi << 16 | class byte J << 8 | rank_lo (rank H of the dword r2); "< 0x200" is the whole predicate, the value the word
// of a [class][256] histogram
template <int J>
__device__ __forceinline__ uint32_t bv_p2d_xm(uint32_t cls4, uint32_t mq4) {  // class byte J << 8 | mapq byte J
    constexpr uint32_t SEL = 0x0C0C0000u | ((4u + J) << 8) | (uint32_t)J;
    return __builtin_amdgcn_perm(cls4, mq4, SEL);
}
template <int J, int H>
__device__ __forceinline__ uint32_t bv_p2d_xr(uint32_t cls4, uint32_t r2) {  // class byte J << 8 | rank_lo (rank H of r2)
    // (the rank's high byte stays out: every caller ORs the high bytes of ALL rank words of a row and re-does a row that holds a
    // rank >= 256 by the window sweeps -- and in the tagged layout, BV_SLAB_RPR_TAGGED, that byte carries the cell's call)
    constexpr uint32_t SEL = 0x0C0C0000u | ((4u + J) << 8) | (uint32_t)(2 * H);
    return __builtin_amdgcn_perm(cls4, r2, SEL);
}

// ---- the tagged rank plane (BV_SLAB_RPR_TAGGED, include/basevar_amd.h): rank | base << 13 | nocall << 15 per 16-bit word.
// what the OR of a row's rank dwords must not hold in the perm form (a rank >= 256), per layout
__device__ __forceinline__ uint32_t bv_rpr_hi_mask(uint32_t tagged) { return tagged ? 0x1F001F00u : 0xFF00FF00u; }
__device__ __forceinline__ uint32_t bv_rpr_rank_mask(uint32_t tagged) { return tagged ? 0x1FFFu : 0xFFFFu; }
// Class bytes (REF 0x00, ALT 0x01, neither >= 0x7F: the values bv_p2d_xm / bv_p2d_xr take) of FOUR cells from the tags of their
// rank words (r01: cells 0, 1; r23: cells 2, 3) -- no call byte is read.  One v_perm gathers the four high bytes, a shift by
// five leaves tag = base | nocall << 2 in every byte (the five rank bits that would spill into the neighbour are zero in any
// row that stays in the perm form; a row with a rank >= 256 is re-done anyway, and whatever selector such a row produces only
// picks some class: the histogram index stays below 0x200), and the tag selects from {L, 0xFFFFFFFF}: a base -> its class
// byte in L (0x80 REF, 0x81 ALT, 0xFF neither), no call -> 0xFF.  4 VALU per four cells, against 2 with the call plane.
__device__ __forceinline__ uint32_t bv_p2t_class4(uint32_t L, uint32_t r01, uint32_t r23) {
    const uint32_t hb = __builtin_amdgcn_perm(r23, r01, 0x07050301u);
    return __builtin_amdgcn_perm(0xFFFFFFFFu, L, hb >> 5) ^ 0x80808080u;
}
// the cells of a row's last chunk at or beyond n_samples read as "no call": `keep` = cells of this rank dword that stay (<= 0: none)
__device__ __forceinline__ uint32_t bv_p2t_mask_tail(uint32_t r2, int keep) {
    return keep >= 2 ? r2 : (keep <= 0 ? 0x80008000u : ((r2 & 0xFFFFu) | 0x80000000u));
}

