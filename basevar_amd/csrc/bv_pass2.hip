// bv_pass2.hip -- pass 2 of the per-site basetype path: variant sites only.
//
// For every site that pass 1 flagged BV_SITE_VARIANT (index list in HBM, no host round
// trip) one workgroup re-reads the row's call plane together with the planes that only the
// VCF record needs, and finishes the record:
//   MQRankSum, ReadPosRankSum    ref_vs_alt_ranksumtest, src/basetype.cpp:201-233 via
//                                caller.cpp:1151-1154 (values are small integers, so the
//                                Wilcoxon statistic is computed from per-class histograms)
//   <group>_AF                   __gb(): BaseType(subset) + lrt([REF]+alts),
//                                caller.cpp:756-759, 767-797 -- one (base x phred)
//                                histogram per pop-group, each solved by one wave
// Traffic: 1 B (calls) + 1 B (mapq) + 2 B (rpr) per cell, + 1 B (phred) when groups exist;
// the group-id vector is shared by all sites and stays in L2.
#include "bv_kernels.h"

#ifndef BV_P2_U64
#define BV_P2_U64 4
#endif
#ifndef BV_P2_WIDE_GROUPS
#define BV_P2_WIDE_GROUPS 2 /* from this many pop-groups on, short rows also take the four-wave kernel: one wave per group */
#endif
#define BV_RPR_WIN 1024  /* read-position ranks per LDS window; longer reads take extra sweeps */

template <int NW>
struct __attribute__((aligned(16))) BvPass2Shared {
    uint32_t hm[2 * 256];         // [class][mapq]        class 0 = REF reads, 1 = ALT reads
    uint32_t hr[2 * BV_RPR_WIN];  // [class][rank - win_lo]
    uint32_t maxr[NW];
    uint32_t bin_code[NW][BV_SLOTS * BV_WAVE];
    uint32_t bin_cnt[NW][BV_SLOTS * BV_WAVE];
    BvLrtShared lrt[NW];
    double tab_hit[BV_QBINS], tab_miss[BV_QBINS];
    uint16_t ord[NW][BV_ORD_MAX];  // shallow pop-groups: the group's covered cells in sample order (bv_gather_ordered)
};

extern __shared__ __attribute__((aligned(16))) uint32_t bv_dyn_lds[];  // hg[n_groups][4][128]

struct BvP2Ctx {
    uint32_t *hm, *hr, *hg;
    uint32_t lut;      // 2 bits per base: 0 REF, 1 ALT, 2 neither
    uint32_t win_lo;
    uint32_t n_groups;
    uint32_t maxr;     // per-lane running max of classified ranks
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
                    uint32_t r = ((j < 2 ? r01 : r23) >> (16 * (j & 1))) & 0xFFFFu;
                    cx.maxr = max(cx.maxr, r);
                    uint32_t rr = r - cx.win_lo;
                    if (rr < (uint32_t)RW) atomicAdd(&cx.hr[cls * RW + rr], 1u);
                }
            }
            if (GROUPS) {
                uint32_t g = (gg >> (8 * j)) & 0xFFu;
                if (g < cx.n_groups) atomicAdd(&cx.hg[((g * 4u + b) << 7) | min((qq >> (8 * j)) & 0xFFu, 127u)], 1u);  // phred >= 128: invalid bin 127 (in the depth, in no valid bin -- as in pass 1)
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

template <int NT, bool RANKS, bool GROUPS>
__global__ __launch_bounds__(NT) void bv_pass2_kernel(BvPass2Args a) {
    constexpr int NW = NT / BV_WAVE;
    __shared__ BvPass2Shared<NW> sh;
    uint32_t *hg = bv_dyn_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_var = a.counters[BV_CTR_VARIANTS];
    if (GROUPS) {
        for (int i = tid; i < BV_QBINS; i += NT) {
            sh.tab_hit[i] = a.tables->hit[i];
            sh.tab_miss[i] = a.tables->miss[i];
        }
    }

    // one variant site per workgroup; surplus workgroups (the grid is sized for the worst case,
    // every site variant, because the count lives in HBM) leave at once.  No grid-stride loop:
    // see bv_pass1.hip.
    const uint32_t v = blockIdx.x;
    if (v >= n_var) return;
    {
        const uint32_t site = a.var_list[v];
        // ---- what pass 1 decided for this site
        const bv_site_result *res = &a.out[site];
        int ref = a.ref_base[site];
        if (ref > 4) ref = 4;
        const int n_alt = res->n_alt;
        uint32_t depth[4] = {res->depth[0], res->depth[1], res->depth[2], res->depth[3]};
        uint32_t lut = 0xAAu;  // every base "neither"
        unsigned long long n1 = 0, n2 = 0;
        if (ref < 4) { lut &= ~(3u << (2 * ref)); n1 = bv_sel4u(depth, ref); }
        int comb = ref, nc = 1;  // caller.cpp:750-753: [toupper(REF)] + alts, 3 bits per entry
#pragma unroll
        for (int k = 0; k < BV_MAX_ALT; ++k) {
            if (k < n_alt) {
                const int b = res->alt[k] & 3;
                lut = (lut & ~(3u << (2 * b))) | (1u << (2 * b));
                n2 += bv_sel4u(depth, b);
                comb |= b << (3 * nc);
                ++nc;
            }
        }

        // ---- clear histograms
        {
            // hm and hr are adjacent and 16-byte aligned: clear them with 128-bit stores
            uint4 *z = reinterpret_cast<uint4 *>(sh.hm);
            for (int i = tid; i < (2 * 256 + 2 * BV_RPR_WIN) / 4; i += NT) z[i] = make_uint4(0, 0, 0, 0);
        }
        if (GROUPS)
            for (uint32_t i = tid; i < a.n_groups * 512u; i += NT) hg[i] = 0u;
        __syncthreads();

        BvP2Ctx cx;
        cx.hm = sh.hm; cx.hr = sh.hr; cx.hg = hg;
        cx.lut = lut; cx.win_lo = 0; cx.n_groups = a.n_groups; cx.maxr = 0;
        bv_p2_sweep<NT, RANKS, RANKS, GROUPS>(cx, a, site, tid);
        if (RANKS) {
            uint32_t mx = (uint32_t)bv_wave_max_i32((int)cx.maxr);
            if (lane == 0) sh.maxr[wave] = mx;
        }
        __syncthreads();

        if (RANKS) {
            uint32_t maxr = 0;
            for (int w = 0; w < NW; ++w) maxr = max(maxr, sh.maxr[w]);
            // MQRankSum on wave 0
            if (wave == 0) {
                unsigned long long below = 0, twoR = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    twoR += bv_ranksum_window(sh.hm[w * 64 + lane], sh.hm[256 + w * 64 + lane], n1 + n2, below, lane);
                double ph = bv_ranksum_phred(twoR, n1, n2);
                if (lane == 0) a.out[site].mq_ranksum = ph;
            }
            // ReadPosRankSum on wave 1 (wave 0 when alone); extra sweeps for ranks >= BV_RPR_WIN
            unsigned long long below = 0, twoR = 0;
            for (uint32_t win_lo = 0;; win_lo += BV_RPR_WIN) {
                if (wave == (1 % NW)) {
                    // ranks beyond the row's largest classified rank hold nothing: stop at its 64-wide block
                    const int nblk = (maxr < win_lo + BV_RPR_WIN) ? (int)((maxr - win_lo) >> 6) + 1 : BV_RPR_WIN / 64;
                    for (int w = 0; w < nblk; ++w)
                        twoR += bv_ranksum_window(sh.hr[w * 64 + lane], sh.hr[BV_RPR_WIN + w * 64 + lane], n1 + n2,
                                                  below, lane);
                }
                if (maxr < win_lo + BV_RPR_WIN) break;
                // long reads: slide the window and re-tally the ranks only
                __syncthreads();
                for (int i = tid; i < 2 * BV_RPR_WIN; i += NT) sh.hr[i] = 0u;
                __syncthreads();
                cx.win_lo = win_lo + BV_RPR_WIN;
                bv_p2_sweep<NT, true, false, false>(cx, a, site, tid);
                __syncthreads();
            }
            if (wave == (1 % NW)) {
                double ph = bv_ranksum_phred(twoR, n1, n2);
                if (lane == 0) {
                    a.out[site].rpr_ranksum = ph;
                    atomicOr(&a.out[site].status, BV_SITE_RANKSUM);
                }
            }
        }

        if (GROUPS) {
            for (uint32_t g = wave; g < a.n_groups; g += NW) {
                const uint32_t *h = hg + g * 512u;
                uint32_t nb = 0, gdepth[4], gtotal = 0, q0_mask = 0;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int b = r >> 1;
                    const int q = ((r & 1) << 6) | lane;
                    uint32_t c = h[(b << 7) | q];
                    if (!(r & 1) && __builtin_amdgcn_readfirstlane((int)c) != 0) q0_mask |= 1u << b;  // phred-0 calls of base b
                    uint32_t cs = bv_wave_sum_u32(c);
                    if (r & 1) gdepth[b] += cs; else gdepth[b] = cs;
                    bool valid = (c != 0) && (q < BV_NQ_VALID);
                    unsigned long long m = __ballot(valid);
                    uint32_t pos = nb + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (valid) {
                        sh.bin_code[wave][pos] = ((uint32_t)b << 7) | (uint32_t)q;
                        sh.bin_cnt[wave][pos] = c;
                    }
                    nb += (uint32_t)__popcll(m);
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) gtotal += gdepth[b];
                bv_lrt_sync<0>();
                BvLrtOut L;
                L.n_alt = 0; L.alt_packed = 0; L.af[0] = L.af[1] = L.af[2] = L.af[3] = 0.;
                if (gtotal > 0) {
                    BvBins B;
                    B.code = sh.bin_code[wave]; B.cnt = sh.bin_cnt[wave]; B.skip_mask = 0u;
                    B.hit = sh.tab_hit; B.miss = sh.tab_miss; B.nb = (int)nb;
                    B.loghit = a.tables->loghit; B.logmiss = a.tables->logmiss; B.ord = nullptr; B.n_ord = 0;
                    const int n_seen = (gdepth[0] != 0) + (gdepth[1] != 0) + (gdepth[2] != 0) + (gdepth[3] != 0);
                    if (gtotal <= (uint32_t)BV_ORD_MAX && n_seen >= 2) {
                        // a shallow group with more than one base: the reference's per-sample order decides ties (bv_em_ordered)
                        const uint32_t got = bv_gather_ordered(a.bs + (size_t)site * a.pitch, a.q + (size_t)site * a.pitch,
                                                               a.n_samples, sh.ord[wave], lane, a.group_id, g);
                        bv_lrt_sync<0>();
                        if (got == gtotal) { B.ord = sh.ord[wave]; B.n_ord = (int)gtotal; }
                    }
                    bv_lrt<0>(B, gdepth, gtotal, comb, nc, ref, a.min_af, &sh.lrt[wave], wave, lane, L, q0_mask);
                }
                if (lane == 0) {
                    bv_group_result gr;
                    gr.n_alt = (uint8_t)L.n_alt;
                    gr.reserved[0] = gr.reserved[1] = gr.reserved[2] = 0;
                    gr.total_depth = gtotal;
                    gr.reserved2 = 0;
#pragma unroll
                    for (int k = 0; k < BV_MAX_ALT; ++k) {
                        gr.alt[k] = (k < L.n_alt) ? (uint8_t)bv_alt_at(L, k) : 0;
                        gr.af[k] = (k < L.n_alt) ? L.af[k] : 0.0;
                    }
                    a.gout[(size_t)site * a.n_groups + g] = gr;
                }
                bv_lrt_sync<0>();
            }
        }
    }
}

// ------------------------------------------------------------------------------ short rows, no pop-groups
// One workgroup launch per variant site costs more than the site itself when a row is a few tens of KB
// (10 k samples: 40 KB).  Here the grid is persistent: four independent waves per workgroup, each with its
// own 4 KiB of histograms (mapq 2 x 256, read-position ranks 2 x 256 per sweep), walk the variant list
// with a fixed stride.  Same arithmetic as bv_pass2_kernel<64, true, false>.
#define BV_P2S_WAVES 4
#define BV_P2S_RW 256
struct __attribute__((aligned(16))) BvPass2ShortShared {
    uint32_t h[BV_P2S_WAVES][2 * 256 + 2 * BV_P2S_RW];  // per wave: hm[2][256] then hr[2][BV_P2S_RW]
};
__global__ __launch_bounds__(BV_WAVE *BV_P2S_WAVES, 4) void bv_pass2_short_kernel(BvPass2Args a) {
    __shared__ BvPass2ShortShared sh;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t *hm = sh.h[wave], *hr = sh.h[wave] + 2 * 256;
    const uint32_t n_var = a.counters[BV_CTR_VARIANTS];
    const uint32_t stride = gridDim.x * BV_P2S_WAVES;
    for (uint32_t v = blockIdx.x * BV_P2S_WAVES + wave; v < n_var; v += stride) {
        const uint32_t site = a.var_list[v];
        const bv_site_result *res = &a.out[site];
        int ref = a.ref_base[site];
        if (ref > 4) ref = 4;
        const int n_alt = res->n_alt;
        uint32_t depth[4] = {res->depth[0], res->depth[1], res->depth[2], res->depth[3]};
        uint32_t lut = 0xAAu;  // every base "neither"
        unsigned long long n1 = 0, n2 = 0;
        if (ref < 4) { lut &= ~(3u << (2 * ref)); n1 = bv_sel4u(depth, ref); }
#pragma unroll
        for (int k = 0; k < BV_MAX_ALT; ++k) {
            if (k < n_alt) {
                const int b = res->alt[k] & 3;
                lut = (lut & ~(3u << (2 * b))) | (1u << (2 * b));
                n2 += bv_sel4u(depth, b);
            }
        }
        {
            uint4 *z = reinterpret_cast<uint4 *>(hm);
#pragma unroll
            for (int i = 0; i < (2 * 256 + 2 * BV_P2S_RW) / 4 / BV_WAVE; ++i) z[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
        }
        bv_lrt_sync<0>();
        BvP2Ctx cx;
        cx.hm = hm; cx.hr = hr; cx.hg = nullptr;
        cx.lut = lut; cx.win_lo = 0; cx.n_groups = 0; cx.maxr = 0;
        bv_p2_sweep<BV_WAVE, true, true, false, BV_P2S_RW>(cx, a, site, lane);
        const uint32_t maxr = (uint32_t)bv_wave_max_i32((int)cx.maxr);
        bv_lrt_sync<0>();
        {   // MQRankSum
            unsigned long long below = 0, twoR = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) twoR += bv_ranksum_window(hm[w * 64 + lane], hm[256 + w * 64 + lane], n1 + n2, below, lane);
            const double ph = bv_ranksum_phred(twoR, n1, n2);
            if (lane == 0) a.out[site].mq_ranksum = ph;
        }
        {   // ReadPosRankSum; extra sweeps of the rank plane for ranks >= BV_P2S_RW
            unsigned long long below = 0, twoR = 0;
            for (uint32_t win_lo = 0;; win_lo += BV_P2S_RW) {
                const int nblk = (maxr < win_lo + BV_P2S_RW) ? (int)((maxr - win_lo) >> 6) + 1 : BV_P2S_RW / 64;
                for (int w = 0; w < nblk; ++w)
                    twoR += bv_ranksum_window(hr[w * 64 + lane], hr[BV_P2S_RW + w * 64 + lane], n1 + n2, below, lane);
                if (maxr < win_lo + BV_P2S_RW) break;
                bv_lrt_sync<0>();
                {
                    uint4 *z = reinterpret_cast<uint4 *>(hr);
#pragma unroll
                    for (int i = 0; i < 2 * BV_P2S_RW / 4 / BV_WAVE; ++i) z[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
                }
                bv_lrt_sync<0>();
                cx.win_lo = win_lo + BV_P2S_RW;
                bv_p2_sweep<BV_WAVE, true, false, false, BV_P2S_RW>(cx, a, site, lane);
                bv_lrt_sync<0>();
            }
            const double ph = bv_ranksum_phred(twoR, n1, n2);
            if (lane == 0) {
                a.out[site].rpr_ranksum = ph;
                atomicOr(&a.out[site].status, BV_SITE_RANKSUM);
            }
        }
        bv_lrt_sync<0>();
    }
}

size_t bv_pass2_lds_bytes(uint32_t n_groups) { return (size_t)n_groups * 512u * sizeof(uint32_t); }

template <int NT>
static void bv_launch_pass2_nt(const BvPass2Args &a, hipStream_t stream) {
    const bool ranks = a.mapq != nullptr && a.rpr != nullptr;
    const bool groups = a.n_groups > 0 && a.group_id != nullptr && a.gout != nullptr;
    if (!ranks && !groups) return;
    uint32_t grid = a.n_sites;
    size_t dyn = groups ? bv_pass2_lds_bytes(a.n_groups) : 0;
    if (ranks && groups)
        hipLaunchKernelGGL((bv_pass2_kernel<NT, true, true>), dim3(grid), dim3(NT), dyn, stream, a);
    else if (ranks)
        hipLaunchKernelGGL((bv_pass2_kernel<NT, true, false>), dim3(grid), dim3(NT), dyn, stream, a);
    else
        hipLaunchKernelGGL((bv_pass2_kernel<NT, false, true>), dim3(grid), dim3(NT), dyn, stream, a);
}

void bv_launch_pass2(const BvPass2Args &a, hipStream_t stream) {
    const bool ranks = a.mapq != nullptr && a.rpr != nullptr;
    const bool groups = a.n_groups > 0 && a.group_id != nullptr && a.gout != nullptr;
    if (a.n_samples <= 16384u && ranks && !groups) {
        uint32_t grid = (a.n_cu ? a.n_cu : 256u) * 4u;  // 16 KiB of LDS and <= 128 VGPRs: 4 workgroups per CU
        const uint32_t need = (a.n_sites + BV_P2S_WAVES - 1) / BV_P2S_WAVES;
        if (grid > need) grid = need;
        hipLaunchKernelGGL(bv_pass2_short_kernel, dim3(grid), dim3(BV_WAVE * BV_P2S_WAVES), 0, stream, a);
        return;
    }
    if (a.n_samples <= 16384u && !(groups && a.n_groups >= BV_P2_WIDE_GROUPS)) bv_launch_pass2_nt<64>(a, stream);
    else bv_launch_pass2_nt<256>(a, stream);
}
