// bv_pass2_sweep.h -- the window sweeps of pass 2 over one row (plain loads, branchy per-cell tally): what every pass-2 kernel
// falls back to for rows that do not fit its fast form (a read-position rank beyond its LDS window), and the generic
// workgroup-per-row kernels' own tally.  Shared by bv_pass2.hip and bv_pass1_fused.hip.
// Reference: ref_vs_alt_ranksumtest on mapq / read-position rank, src/basetype.cpp:201-242 via caller.cpp:1151-1154.
#pragma once

#include "bv_kernels.h"

#define BV_P2_U64 4      /* 16-byte chunks per thread and iteration when one wave sweeps a row (latency-bound: more loads in flight) */
#define BV_RPR_WIN 1024  /* read-position ranks per LDS window; longer reads take extra sweeps */

struct BvP2Ctx {
    uint32_t *hm, *hr, *hg;
    uint32_t lut;      // 2 bits per base: 0 REF, 1 ALT, 2 neither
    uint32_t win_lo;
    uint32_t n_groups;
    uint32_t maxr;     // per-lane running max of classified ranks
    bool half;         // hg holds 16-bit counters, two per word (rows of at most 65,535 samples; bv_pass2_kernel<.., HALF>)
    uint32_t rmask = 0xFFFFu;  // the rank bits of a word of the rpr plane (0x1FFF: BV_SLAB_RPR_TAGGED)
};

template <bool RANKS, bool MAPQ, bool GROUPS, int RW = BV_RPR_WIN>
__device__ __forceinline__ void bv_p2_dword(BvP2Ctx &cx, uint32_t w, uint32_t mq, uint32_t r01, uint32_t r23,
                                            uint32_t qq, uint32_t gg) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t c = (w >> (8 * j)) & 0xFFu;
        if (c < 8u) {  // a call byte is 0..7; N / indel tokens (8..10) and anything else count as "no call", as in pass 1
            uint32_t b = c & 3u;
            if (RANKS) {
                uint32_t cls = (cx.lut >> (2 * b)) & 3u;
                if (cls < 2u) {
                    if (MAPQ) atomicAdd(&cx.hm[cls * 256u + ((mq >> (8 * j)) & 0xFFu)], 1u);
                    uint32_t r = ((j < 2 ? r01 : r23) >> (16 * (j & 1))) & cx.rmask;
                    cx.maxr = max(cx.maxr, r);
                    uint32_t rr = r - cx.win_lo;
                    if (rr < (uint32_t)RW) atomicAdd(&cx.hr[cls * RW + rr], 1u);
                }
            }
            if (GROUPS) {
                uint32_t g = (gg >> (8 * j)) & 0xFFu;
                if (g < cx.n_groups) {
                    const uint32_t gi = ((g * 4u + b) << 7) | min((qq >> (8 * j)) & 0xFFu, 127u);  // phred >= 128: invalid bin 127 (in the depth, in no valid bin -- as in pass 1)
                    if (cx.half) atomicAdd(&cx.hg[gi >> 1], 1u << (16u * (gi & 1u)));
                    else atomicAdd(&cx.hg[gi], 1u);
                }
            }
        }
    }
}

__device__ __forceinline__ uint32_t bv_p2_mask_tail(uint32_t w, int keep) {
    if (keep >= 4) return w;
    if (keep <= 0) return 0x08080808u;
    uint32_t low = (1u << (8 * keep)) - 1u;
    return (w & low) | (0x08080808u & ~low);
}

// one sweep over the row; the first sweep (MAPQ/GROUPS as configured) fills everything,
// later sweeps (rank window > 0) only re-tally read-position ranks
template <int NT, bool RANKS, bool MAPQ, bool GROUPS, int RW = BV_RPR_WIN>
__device__ __forceinline__ void bv_p2_sweep(BvP2Ctx &cx, const BvPass2Args &a, uint32_t site, int tid) {
    const size_t row = (size_t)site * a.pitch;
    const bv_u32x4 *b4 = reinterpret_cast<const bv_u32x4 *>(a.bs + row);
    const bv_u32x4 *m4 = MAPQ ? reinterpret_cast<const bv_u32x4 *>(a.mapq + row) : nullptr;
    const bv_u32x4 *r4 = RANKS ? reinterpret_cast<const bv_u32x4 *>(a.rpr + row) : nullptr;
    const bv_u32x4 *q4 = GROUPS ? reinterpret_cast<const bv_u32x4 *>(a.q + row) : nullptr;
    const bv_u32x4 *g4 = GROUPS ? reinterpret_cast<const bv_u32x4 *>(a.group_id) : nullptr;
    const uint32_t n_chunks = (a.n_samples + 15u) >> 4;
    const int tail = (int)(a.n_samples & 15u);
    const bv_u32x4 zero = bv_u32x4{0u, 0u, 0u, 0u};
    // chunks per thread and iteration: short rows (one wave per site) are latency-bound -> more loads in flight
    constexpr int U = (NT == 64) ? BV_P2_U64 : 2;
    for (uint32_t base = 0; base < n_chunks; base += NT * U) {
        bv_u32x4 vb[U], vm[U], vr0[U], vr1[U], vq[U], vg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t idx = base + u * NT + tid;
            vb[u] = bv_u32x4{0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u};
            vm[u] = vr0[u] = vr1[u] = vq[u] = vg[u] = zero;
            if (idx < n_chunks) {
                vb[u] = __builtin_nontemporal_load(b4 + idx);
                if (MAPQ) vm[u] = __builtin_nontemporal_load(m4 + idx);
                if (RANKS) {
                    vr0[u] = __builtin_nontemporal_load(r4 + 2 * (size_t)idx);
                    vr1[u] = __builtin_nontemporal_load(r4 + 2 * (size_t)idx + 1);
                }
                if (GROUPS) {
                    vq[u] = __builtin_nontemporal_load(q4 + idx);
                    vg[u] = g4[idx];  // shared by every site: keep it cacheable
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t idx = base + u * NT + tid;
            if (tail && idx == n_chunks - 1) {
                vb[u].x = bv_p2_mask_tail(vb[u].x, tail);
                vb[u].y = bv_p2_mask_tail(vb[u].y, tail - 4);
                vb[u].z = bv_p2_mask_tail(vb[u].z, tail - 8);
                vb[u].w = bv_p2_mask_tail(vb[u].w, tail - 12);
            }
            bv_p2_dword<RANKS, MAPQ, GROUPS, RW>(cx, vb[u].x, vm[u].x, vr0[u].x, vr0[u].y, vq[u].x, vg[u].x);
            bv_p2_dword<RANKS, MAPQ, GROUPS, RW>(cx, vb[u].y, vm[u].y, vr0[u].z, vr0[u].w, vq[u].y, vg[u].y);
            bv_p2_dword<RANKS, MAPQ, GROUPS, RW>(cx, vb[u].z, vm[u].z, vr1[u].x, vr1[u].y, vq[u].z, vg[u].z);
            bv_p2_dword<RANKS, MAPQ, GROUPS, RW>(cx, vb[u].w, vm[u].w, vr1[u].z, vr1[u].w, vq[u].w, vg[u].w);
        }
    }
}

