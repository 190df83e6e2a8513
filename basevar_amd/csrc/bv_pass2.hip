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
#include "bv_tally.h"
#include "bv_pass2_sweep.h"  // BvP2Ctx, bv_p2_sweep, BV_RPR_WIN, BV_P2_U64

#define BV_P2_WIDE_GROUPS 2 /* from this many pop-groups on, short rows also take the four-wave kernel: one wave per group */
#define BV_P2_BIG_GROUPS 12 /* from this many pop-groups on, long rows take workgroups of eight waves */

// INLINE: the kernel solves pop-groups itself (one wave per group) and needs the solver's LDS; otherwise every group leaves
// as an item for the group solve kernels and that LDS (and the solver's registers) are not taken.
// RANKS: the kernel forms the rank sums; without them (the group tallies behind a fused pass-1 kernel that has streamed the
// rank-sum rows itself) their 10 KiB are not taken either -- at 32 groups that is a fourth workgroup per CU.
template <int NW, bool INLINE = true, bool RANKS = true>
struct __attribute__((aligned(16))) BvPass2Shared {
    uint32_t hm[RANKS ? 2 * 256 : 4];         // [class][mapq]        class 0 = REF reads, 1 = ALT reads
    uint32_t hr[RANKS ? 2 * BV_RPR_WIN : 4];  // [class][rank - win_lo]
    uint32_t maxr[NW];
    uint32_t bin_code[INLINE ? NW : 1][INLINE ? BV_SLOTS * BV_WAVE : 4];
    uint32_t bin_cnt[INLINE ? NW : 1][INLINE ? BV_SLOTS * BV_WAVE : 4];
    BvLrtShared lrt[INLINE ? NW : 1];
    double tab_hit[INLINE ? BV_QBINS : 2], tab_miss[INLINE ? BV_QBINS : 2];
    alignas(8) uint16_t ord[INLINE ? NW : 1][BV_ORD_ALLOC];  // shallow pop-groups: the group's covered cells in sample order (bv_gather_ordered)
    uint32_t ghdr[INLINE ? 1 : BV_GROUPS_PER_ROUND];  // !INLINE: the header words of the site's items, until the site's kind of small-group solver is known
};

extern __shared__ __attribute__((aligned(16))) uint32_t bv_dyn_lds[];  // hg[n_groups][4][128]

// (BvP2Ctx, bv_p2_dword, bv_p2_mask_tail, bv_p2_sweep -- the window sweeps over a row: bv_pass2_sweep.h)

// (bv_p2d_xm / bv_p2d_xr, the perm form of the rank-sum tally: bv_tally.h)

// One sweep over the row in that form, by the whole workgroup into its shared mapq / rank histograms ([2][256] each; `hr` uses the
// first 512 words of the 1024-rank window) and, with GROUPS, into the per-group (base, phred) histograms `hg`.  Returns this
// thread's OR of what does not fit the form: non-zero somewhere in the workgroup = a rank >= 256 (or a call byte > 15 / a
// phred byte > 127) in the row, the caller re-does the row with the branchy window sweeps (bv_p2_sweep).  L: class table
// (byte b = 0x80 REF / 0x81 ALT / 0xFF neither).  ~6 VALU + 2 predicated ds_add per cell for the rank sums, ~4 + 1 for the
// groups; the branchy sweep takes ~18 VALU for the rank sums alone.
// TAG (rank sums without pop-groups, BV_SLAB_RPR_TAGGED): the class of a cell comes from the tag in its rank word
// (bv_p2t_class4) and the call plane is not read at all -- 3 bytes per cell, SURVEY 8d's figure, instead of 4.
// DOM: the mapq tally takes a dominant value out of the LDS adds (bv_lds_add16_dom): chosen per row by its depth.
template <int NT, bool RANKS, bool GROUPS, bool HALF = false, bool TAG = false, bool DOM = false>
__device__ __forceinline__ uint32_t bv_p2_fast_sweep(const BvPass2Args &a, uint32_t site, int tid, uint32_t L, uint32_t *hm, uint32_t *hr, uint32_t *hg,
                                                     uint32_t *q64 = nullptr /* with GROUPS: this thread's OR of bit 6 of the row's phred bytes */) {
    const size_t row = (size_t)site * a.pitch;
    const bv_u32x4 *b4 = reinterpret_cast<const bv_u32x4 *>(a.bs + row);
    const bv_u32x4 *m4 = RANKS ? reinterpret_cast<const bv_u32x4 *>(a.mapq + row) : nullptr;
    const bv_u32x4 *r4 = RANKS ? reinterpret_cast<const bv_u32x4 *>(a.rpr + row) : nullptr;
    const bv_u32x4 *q4 = GROUPS ? reinterpret_cast<const bv_u32x4 *>(a.q + row) : nullptr;
    const bv_u32x4 *g4 = GROUPS ? reinterpret_cast<const bv_u32x4 *>(a.gidp) : nullptr;  // g << 2, or 0x80: no group
    const uint32_t n_chunks = (a.n_samples + 15u) >> 4;
    const int tail = (int)(a.n_samples & 15u);
    const bv_u32x4 zero = bv_u32x4{0u, 0u, 0u, 0u}, none = bv_u32x4{0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u};
    uint32_t one;
    asm volatile("v_mov_b32 %0, 1" : "=v"(one));
    static_assert(!TAG || (RANKS && !GROUPS), "the tagged form: rank sums without pop-groups");
    const bv_u32x4 nocall = bv_u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
    const uint32_t hi_mask = TAG ? 0x1F001F00u : bv_rpr_hi_mask(a.rpr_tag);
    uint32_t hi_acc = 0, dom = BV_DOM_NONE, q64_acc = 0;
    // chunks per thread and trip: without the rank planes three loads per chunk, so four chunks -- a 10,000-sample row is then
    // ONE round of loads for a workgroup of 256 (the kernel waits for memory, not for instructions)
    constexpr int U = RANKS ? 2 : 4;
    for (uint32_t base = 0; base < n_chunks; base += NT * U) {
        bv_u32x4 vb[U], vm[U], vr0[U], vr1[U], vq[U], vg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t idx = base + u * NT + tid;
            vb[u] = none; vm[u] = vr0[u] = vr1[u] = vq[u] = vg[u] = zero;
            if (TAG) vr0[u] = vr1[u] = nocall;
            if (idx < n_chunks) {
                if (!TAG) vb[u] = __builtin_nontemporal_load(b4 + idx);
                if (RANKS) {
                    vm[u] = __builtin_nontemporal_load(m4 + idx);
                    vr0[u] = __builtin_nontemporal_load(r4 + 2 * (size_t)idx);
                    vr1[u] = __builtin_nontemporal_load(r4 + 2 * (size_t)idx + 1);
                }
                if (GROUPS) {
                    vq[u] = __builtin_nontemporal_load(q4 + idx);
                    vg[u] = g4[idx];  // shared by every row: cacheable
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t idx = base + u * NT + tid;
            if (tail && idx == n_chunks - 1) {
                if (TAG) {
                    vr0[u].x = bv_p2t_mask_tail(vr0[u].x, tail); vr0[u].y = bv_p2t_mask_tail(vr0[u].y, tail - 2);
                    vr0[u].z = bv_p2t_mask_tail(vr0[u].z, tail - 4); vr0[u].w = bv_p2t_mask_tail(vr0[u].w, tail - 6);
                    vr1[u].x = bv_p2t_mask_tail(vr1[u].x, tail - 8); vr1[u].y = bv_p2t_mask_tail(vr1[u].y, tail - 10);
                    vr1[u].z = bv_p2t_mask_tail(vr1[u].z, tail - 12); vr1[u].w = bv_p2t_mask_tail(vr1[u].w, tail - 14);
                } else {
                    vb[u].x = bv_p2_mask_tail(vb[u].x, tail); vb[u].y = bv_p2_mask_tail(vb[u].y, tail - 4);
                    vb[u].z = bv_p2_mask_tail(vb[u].z, tail - 8); vb[u].w = bv_p2_mask_tail(vb[u].w, tail - 12);
                }
            }
            uint32_t x[16];
            if (RANKS) {
                const bv_u32x4 r0 = vr0[u], r1 = vr1[u], vmq = vm[u];
                uint32_t c0, c1, c2, c3;
                if (TAG) {
                    c0 = bv_p2t_class4(L, r0.x, r0.y); c1 = bv_p2t_class4(L, r0.z, r0.w);
                    c2 = bv_p2t_class4(L, r1.x, r1.y); c3 = bv_p2t_class4(L, r1.z, r1.w);
                } else {
                    c0 = __builtin_amdgcn_perm(L, L, vb[u].x) ^ 0x80808080u; c1 = __builtin_amdgcn_perm(L, L, vb[u].y) ^ 0x80808080u;
                    c2 = __builtin_amdgcn_perm(L, L, vb[u].z) ^ 0x80808080u; c3 = __builtin_amdgcn_perm(L, L, vb[u].w) ^ 0x80808080u;
                }
                hi_acc |= (r0.x | r0.y | r0.z | r0.w | r1.x | r1.y | r1.z | r1.w) & hi_mask;
                x[0] = bv_p2d_xm<0>(c0, vmq.x); x[1] = bv_p2d_xm<1>(c0, vmq.x); x[2] = bv_p2d_xm<2>(c0, vmq.x); x[3] = bv_p2d_xm<3>(c0, vmq.x);
                x[4] = bv_p2d_xm<0>(c1, vmq.y); x[5] = bv_p2d_xm<1>(c1, vmq.y); x[6] = bv_p2d_xm<2>(c1, vmq.y); x[7] = bv_p2d_xm<3>(c1, vmq.y);
                x[8] = bv_p2d_xm<0>(c2, vmq.z); x[9] = bv_p2d_xm<1>(c2, vmq.z); x[10] = bv_p2d_xm<2>(c2, vmq.z); x[11] = bv_p2d_xm<3>(c2, vmq.z);
                x[12] = bv_p2d_xm<0>(c3, vmq.w); x[13] = bv_p2d_xm<1>(c3, vmq.w); x[14] = bv_p2d_xm<2>(c3, vmq.w); x[15] = bv_p2d_xm<3>(c3, vmq.w);
                uint32_t y[16];
                y[0] = bv_p2d_xr<0, 0>(c0, r0.x); y[1] = bv_p2d_xr<1, 1>(c0, r0.x); y[2] = bv_p2d_xr<2, 0>(c0, r0.y); y[3] = bv_p2d_xr<3, 1>(c0, r0.y);
                y[4] = bv_p2d_xr<0, 0>(c1, r0.z); y[5] = bv_p2d_xr<1, 1>(c1, r0.z); y[6] = bv_p2d_xr<2, 0>(c1, r0.w); y[7] = bv_p2d_xr<3, 1>(c1, r0.w);
                y[8] = bv_p2d_xr<0, 0>(c2, r1.x); y[9] = bv_p2d_xr<1, 1>(c2, r1.x); y[10] = bv_p2d_xr<2, 0>(c2, r1.y); y[11] = bv_p2d_xr<3, 1>(c2, r1.y);
                y[12] = bv_p2d_xr<0, 0>(c3, r1.z); y[13] = bv_p2d_xr<1, 1>(c3, r1.z); y[14] = bv_p2d_xr<2, 0>(c3, r1.w); y[15] = bv_p2d_xr<3, 1>(c3, r1.w);
                // (both histograms under one predicate -- the class byte is the same in x and y -- unless the row is deep and its mapq
                // tally counts the dominant value)
                if (DOM) {
                    bv_lds_add16_dom<2>(x, hm, one, 0x200u, dom);
                    bv_lds_add16<2>(y, hr, one, 0x200u);
                } else bv_lds_add16x2<2>(x, y, hm, hr, one, 0x200u);
            }
            if (GROUPS) {
                // the group tally of bv_p2g_stream_kernel: byte = group << 2 | base, bit 7 for "no call" / "no group"; X = byte << 8 |
                // phred << 1 is twice the word index of hg[group][base][128].  Call bytes above 15 and phred bytes above 127 do not
                // fit the packed index: reported like a long read, the row is then re-done by the branchy sweep
                bv_u32x4 q2 = vq[u];
                hi_acc |= ((vb[u].x | vb[u].y | vb[u].z | vb[u].w) & 0xF0F0F0F0u) | ((q2.x | q2.y | q2.z | q2.w) & 0x80808080u);
                q64_acc |= (q2.x | q2.y | q2.z | q2.w) & 0x40404040u;
                const uint32_t y0 = (((vb[u].x & 0x08080808u) << 4) | (vb[u].x & 0x03030303u)) | vg[u].x, y1 = (((vb[u].y & 0x08080808u) << 4) | (vb[u].y & 0x03030303u)) | vg[u].y;
                const uint32_t y2 = (((vb[u].z & 0x08080808u) << 4) | (vb[u].z & 0x03030303u)) | vg[u].z, y3 = (((vb[u].w & 0x08080808u) << 4) | (vb[u].w & 0x03030303u)) | vg[u].w;
                q2.x = (q2.x & 0x7F7F7F7Fu) << 1; q2.y = (q2.y & 0x7F7F7F7Fu) << 1; q2.z = (q2.z & 0x7F7F7F7Fu) << 1; q2.w = (q2.w & 0x7F7F7F7Fu) << 1;
                x[0] = bv_cell_index<0>(y0, q2.x); x[1] = bv_cell_index<1>(y0, q2.x); x[2] = bv_cell_index<2>(y0, q2.x); x[3] = bv_cell_index<3>(y0, q2.x);
                x[4] = bv_cell_index<0>(y1, q2.y); x[5] = bv_cell_index<1>(y1, q2.y); x[6] = bv_cell_index<2>(y1, q2.y); x[7] = bv_cell_index<3>(y1, q2.y);
                x[8] = bv_cell_index<0>(y2, q2.z); x[9] = bv_cell_index<1>(y2, q2.z); x[10] = bv_cell_index<2>(y2, q2.z); x[11] = bv_cell_index<3>(y2, q2.z);
                x[12] = bv_cell_index<0>(y3, q2.w); x[13] = bv_cell_index<1>(y3, q2.w); x[14] = bv_cell_index<2>(y3, q2.w); x[15] = bv_cell_index<3>(y3, q2.w);
                if (HALF) bv_lds_add16_half(x, hg, one, 0x8000u);  // X is the byte offset of a 16-bit counter of hg[group][base][128]
                else bv_lds_add16<1>(x, hg, one, 0x8000u);
            }
        }
    }
    if (GROUPS && q64 != nullptr) *q64 = q64_acc;
    return hi_acc;
}

// HALF (with GROUPS, not INLINE; rows of at most 65,535 samples): the group histograms hold 16-bit counters, 1 KiB per group
// instead of 2 -- with 16-32 groups that is what decides how many workgroups a CU holds (32 groups: 45 KiB instead of 77).
template <int NT, bool RANKS, bool GROUPS, bool INLINE = true, bool HALF = false, bool TAG = false>
__global__ __launch_bounds__(NT) void bv_pass2_kernel(BvPass2Args a) {
    static_assert(!HALF || (GROUPS && !INLINE), "16-bit group counters: the item-exporting form only");
    constexpr uint32_t GW = HALF ? 256u : 512u;  // words of one group's histogram
    constexpr int NW = NT / BV_WAVE;
    __shared__ BvPass2Shared<NW, INLINE, RANKS> sh;
    uint32_t *hg = bv_dyn_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_var = a.counters[BV_CTR_VARIANTS];
    if (GROUPS && INLINE) {
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
        if (a.ch != nullptr) {  // a chained launch: the segment's (biased) planes and records
            const BvChainC ch = bv_chain_const(a.ch);
            const uint32_t sg = bv_chain_seg(ch, (uint32_t)__builtin_amdgcn_readfirstlane((int)site));
            a.bs = ch->bs[sg]; a.q = ch->q[sg]; a.mapq = ch->mapq[sg]; a.rpr = ch->rpr[sg];
            if (!a.ch_cat) { a.ref_base = ch->ref_base[sg]; a.out = ch->out[sg]; }
            if (GROUPS) a.gout = ch->gout[sg];
        }
        // ---- what pass 1 decided for this site
        const bv_site_result *res = &a.out[site];
        int ref = a.ref_base[site];
        if (ref > 4) ref = 4;
        const int n_alt = res->n_alt;
        uint32_t depth[4] = {res->depth[0], res->depth[1], res->depth[2], res->depth[3]};
        uint32_t lut = 0xAAu;  // every base "neither"
        uint32_t Ltab = 0xFFFFFFFFu;  // the same as a byte table for the perm form: 0x80 REF, 0x81 ALT, 0xFF neither
        unsigned long long n1 = 0, n2 = 0;
        if (ref < 4) { lut &= ~(3u << (2 * ref)); Ltab = (Ltab & ~(0xFFu << (8 * ref))) | (0x80u << (8 * ref)); n1 = bv_sel4u(depth, ref); }
        int comb = ref, nc = 1;  // caller.cpp:750-753: [toupper(REF)] + alts, 3 bits per entry
#pragma unroll
        for (int k = 0; k < BV_MAX_ALT; ++k) {
            if (k < n_alt) {
                const int b = res->alt[k] & 3;
                lut = (lut & ~(3u << (2 * b))) | (1u << (2 * b));
                Ltab = (Ltab & ~(0xFFu << (8 * b))) | (0x81u << (8 * b));
                n2 += bv_sel4u(depth, b);
                comb |= b << (3 * nc);
                ++nc;
            }
        }

        // ---- clear histograms
        if (RANKS) {
            // hm and hr are adjacent and 16-byte aligned: clear them with 128-bit stores
            uint4 *z = reinterpret_cast<uint4 *>(sh.hm);
            for (int i = tid; i < (2 * 256 + 2 * BV_RPR_WIN) / 4; i += NT) z[i] = make_uint4(0, 0, 0, 0);
        }
        if (GROUPS) {  // (16-byte stores: GW is a multiple of 4 and the dynamic block is 16-byte aligned)
            uint4 *z = reinterpret_cast<uint4 *>(hg);
            for (uint32_t i = tid; i < a.n_groups * (GW / 4u); i += NT) z[i] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();

        BvP2Ctx cx;
        cx.hm = sh.hm; cx.hr = sh.hr; cx.hg = hg;
        cx.lut = lut; cx.win_lo = 0; cx.n_groups = a.n_groups; cx.maxr = 0; cx.half = HALF; cx.rmask = bv_rpr_rank_mask(a.rpr_tag);
        // Rank sums without pop-groups: the perm form first (a third of the instructions; 256-rank window).  A row that holds a
        // rank >= 256 (long reads) is re-done by the window sweeps below.
        const bool FAST = !GROUPS || a.gidp != nullptr;
        bool fast_ok = false, lo_only = false;
        if (FAST) {
            // a deep row (an eighth of its cells are REF / ALT reads): most lanes of a wave add to the dominant mapq's word
            const bool deep = RANKS && !GROUPS && (n1 + n2) * 8ull >= (unsigned long long)a.n_samples && !(a.flags & BV_FLAG_NO_DOM);
            uint32_t q64 = 0;
            const uint32_t hi = deep ? bv_p2_fast_sweep<NT, RANKS, GROUPS, HALF, TAG, RANKS && !GROUPS>(a, site, tid, Ltab, sh.hm, sh.hr, hg)
                                     : bv_p2_fast_sweep<NT, RANKS, GROUPS, HALF, TAG, false>(a, site, tid, Ltab, sh.hm, sh.hr, hg, &q64);
            const bool any_hi = __ballot(hi != 0u) != 0ull, any_q64 = GROUPS && __ballot(q64 != 0u) != 0ull;
            if (lane == 0) sh.maxr[wave] = (any_hi ? 1u : 0u) | (any_q64 ? 2u : 0u);
            __syncthreads();
            uint32_t slow = 0;
            for (int w = 0; w < NW; ++w) slow |= sh.maxr[w];
            fast_ok = (slow & 1u) == 0u;
            // no phred byte of the row has bit 6 or 7 set (every Illumina row: phreds stop at 41): the upper halves of the group
            // histograms -- phred 64-127 -- are empty, and the export below does not read them
            lo_only = fast_ok && (slow & 2u) == 0u;
            __syncthreads();
            if (!fast_ok) {
                uint4 *z = reinterpret_cast<uint4 *>(sh.hm);
                if (RANKS)
                    for (int i = tid; i < (2 * 256 + 2 * BV_RPR_WIN) / 4; i += NT) z[i] = make_uint4(0, 0, 0, 0);
                if (GROUPS)
                    for (uint32_t i = tid; i < a.n_groups * GW; i += NT) hg[i] = 0u;
                __syncthreads();
            }
        }
        if (!fast_ok) {
            bv_p2_sweep<NT, RANKS, RANKS, GROUPS>(cx, a, site, tid);
            if (RANKS) {
                uint32_t mx = (uint32_t)bv_wave_max_i32((int)cx.maxr);
                if (lane == 0) sh.maxr[wave] = mx;
            }
            __syncthreads();
        }

        if (RANKS) {
            uint32_t maxr = 255u;  // the perm form: one window of 256 ranks, ALT counts at word 256
            const uint32_t alt_off = fast_ok ? 256u : (uint32_t)BV_RPR_WIN;
            if (!fast_ok) {
                maxr = 0;
                for (int w = 0; w < NW; ++w) maxr = max(maxr, sh.maxr[w]);
            }
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
                        twoR += bv_ranksum_window(sh.hr[w * 64 + lane], sh.hr[alt_off + w * 64 + lane], n1 + n2,
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
                const uint32_t *h = hg + g * GW;
                if (!INLINE) {
                    // every group leaves as an item (bv_p2g_solve16_kernel / bv_p2g_hard_kernel): bins straight from the histogram
                    uint32_t c[8], dpart[4], q0 = 0, cm = 0, nbv = 0;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        if ((r & 1) && lo_only) { c[r] = 0u; continue; }  // (wave-uniform: see lo_only)
                        if (HALF) c[r] = (h[(((r >> 1) << 7) | ((r & 1) << 6) | lane) >> 1] >> (16 * (lane & 1))) & 0xFFFFu;
                        else c[r] = h[((r >> 1) << 7) | ((r & 1) << 6) | lane];
                        cm = max(cm, c[r]);
                        if (!(r & 1) && __builtin_amdgcn_readfirstlane((int)c[r]) != 0) q0 |= 1u << (r >> 1);
                        const bool valid = c[r] != 0u && (((r & 1) << 6) | lane) < BV_NQ_VALID;
                        nbv += (uint32_t)__popcll(__ballot(valid));
                    }
#pragma unroll
                    for (int b = 0; b < 4; ++b) dpart[b] = c[2 * b] + c[2 * b + 1];
                    uint32_t gd[4];
                    {
                        const uint32_t v8[8] = {dpart[0], dpart[1], dpart[2], dpart[3], 0u, 0u, 0u, 0u};
                        uint32_t t8[8];
                        bv_wave_sum8_u32(v8, t8, lane);
                        gd[0] = t8[0]; gd[1] = t8[1]; gd[2] = t8[2]; gd[3] = t8[3];
                    }
                    const uint32_t gt = gd[0] + gd[1] + gd[2] + gd[3];
                    const int seen = (gd[0] != 0) + (gd[1] != 0) + (gd[2] != 0) + (gd[3] != 0);
                    const bool shal = gt <= (uint32_t)BV_ORD_MAX && seen >= 2;
                    const bool big = __ballot(cm > 0xFFFFu) != 0ull;
                    const bool four = q0 == 0u && nbv <= (uint32_t)BV_G16_MAX_BINS && !big && a.min_af > 0.0 && !(a.flags & BV_FLAG_WAVE_SOLVER);
                    uint32_t *dst = a.gitems + ((size_t)v * a.n_groups + g) * BV_P2G_ITEM_WORDS;
                    if (gt != 0u) {
                        uint32_t pos0 = 0;
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            if ((r & 1) && lo_only) continue;
                            const uint32_t code = ((uint32_t)(r >> 1) << 7) | (uint32_t)(((r & 1) << 6) | lane);
                            const bool valid = c[r] != 0u && (code & 127u) < (uint32_t)BV_NQ_VALID;
                            const unsigned long long m = __ballot(valid);
                            const uint32_t pos = pos0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                            if (valid) dst[8u + pos] = four ? ((code << 16) | c[r]) : ((code << 23) | c[r]);
                            pos0 += (uint32_t)__popcll(m);
                        }
                    }
                    const uint32_t hdr = gt == 0u ? 0u : (nbv | (four ? BV_P2G_PENDING : BV_P2G_HARD) | (shal ? BV_P2G_SHALLOW : 0u));
                    uint32_t w = 0u;  // (word 0, the header, follows when the site's kind is known: below)
#pragma unroll
                    for (int b = 0; b < 4; ++b) w = (lane == 1 + b) ? gd[b] : w;
                    w = (lane == 5) ? q0 : w;
                    if (lane >= 1 && lane < 6) dst[lane] = w;
                    if (lane == 0) sh.ghdr[g] = hdr;
                    continue;
                }
                uint32_t nb = 0, gdepth[4], gtotal = 0, q0_mask = 0;
                uint32_t cmax = 0;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int b = r >> 1;
                    const int q = ((r & 1) << 6) | lane;
                    uint32_t c = h[(b << 7) | q];
                    cmax = max(cmax, c);
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
                const int n_seen = (gdepth[0] != 0) + (gdepth[1] != 0) + (gdepth[2] != 0) + (gdepth[3] != 0);
                const bool shallow = gtotal <= (uint32_t)BV_ORD_MAX && n_seen >= 2;
                // The LRT of an ordinary group is a small problem (<= 128 bins) that a quarter wave solves as fast as a
                // whole one: hand it to bv_p2g_solve16_kernel (four groups per wave) and go on streaming.  Kept here: shallow
                // groups (ordered replay), phred-0 calls and min_af <= 0 (literal 0/0 arithmetic), > 128 bins, counts past 16 bits.
                const size_t item = (size_t)v * a.n_groups + g;
                if (a.gitems != nullptr && item < (size_t)a.gitem_cap) {
                    uint32_t *dst = a.gitems + item * BV_P2G_ITEM_WORDS;
                    const bool big = __ballot(cmax > 0xFFFFu) != 0ull;
                    const bool hand_over = gtotal > 0 && !shallow && q0_mask == 0u && nb <= (uint32_t)BV_G16_MAX_BINS && !big && a.min_af > 0.0 &&
                                           !(a.flags & BV_FLAG_WAVE_SOLVER);
                    if (lane == 0) {
                        dst[0] = hand_over ? (nb | BV_P2G_PENDING) : 0u;
                        dst[1] = gdepth[0]; dst[2] = gdepth[1]; dst[3] = gdepth[2]; dst[4] = gdepth[3]; dst[5] = q0_mask;
                    }
                    if (hand_over) {
                        for (uint32_t i = (uint32_t)lane; i < nb; i += BV_WAVE) dst[8u + i] = (sh.bin_code[wave][i] << 16) | sh.bin_cnt[wave][i];
                        bv_lrt_sync<0>();
                        continue;
                    }
                }
                BvLrtOut L;
                L.n_alt = 0; L.alt_packed = 0; L.af[0] = L.af[1] = L.af[2] = L.af[3] = 0.;
                if (gtotal > 0) {
                    BvBins B;
                    B.code = sh.bin_code[wave]; B.cnt = sh.bin_cnt[wave]; B.skip_mask = 0u;
                    B.hit = sh.tab_hit; B.miss = sh.tab_miss; B.nb = (int)nb;
                    B.loghit = a.tables->loghit; B.logmiss = a.tables->logmiss; B.ord = nullptr; B.n_ord = 0;
                    if (shallow) {
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
            if (!INLINE) {
                // the site's items are written: which small-group solver takes them (BV_P2G_L4 / BV_P2G_L8, bv_kernels.h) is decided
                // for the site -- lane g of wave 0 classifies group g and writes its header
                __syncthreads();
                if (wave == 0) {
                    const uint32_t hdr = (uint32_t)lane < a.n_groups ? sh.ghdr[lane] : 0u;
                    const bool pend = (hdr & BV_P2G_PENDING) != 0u;
                    const uint32_t nbv = hdr & 0xFFFFu;
                    // (Measured, 100 k sites x 10 k samples, 8 / 16 / 32 / 64 groups: this rule 115 / 93 / 70 / 45 M sites/s; "at most 32
                    // bins -> 4 lanes" per item 103 / 80 / 73 / 46, per site 109 / 80 / 73 / 46: a job whose items need eight slots per
                    // lane takes three times as long as one that needs four, so 4 lanes pay only where the items fit 16 bins.)
                    const uint32_t n_pend = (uint32_t)__popcll(__ballot(pend)), n_16 = (uint32_t)__popcll(__ballot(pend && nbv <= 16u));
                    const bool site_l4 = 4u * n_16 >= 3u * n_pend;
                    uint32_t kind = 0u;
                    if (pend && site_l4 && nbv <= 32u) kind = BV_P2G_L4;
                    else if (pend && nbv <= 64u) kind = BV_P2G_L8;
                    if ((uint32_t)lane < a.n_groups) a.gitems[((size_t)v * a.n_groups + (uint32_t)lane) * BV_P2G_ITEM_WORDS] = hdr | kind;
                }
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
        cx.lut = lut; cx.win_lo = 0; cx.n_groups = 0; cx.maxr = 0; cx.half = false; cx.rmask = bv_rpr_rank_mask(a.rpr_tag);
        BvPass2Args as = a;  // the sweeps index the planes with the site number themselves
        if (a.ch != nullptr) {  // a chained launch (short rows: ref_base / out are contiguous): the segment's biased planes
            const BvChainC ch = bv_chain_const(a.ch);
            const uint32_t sg = bv_chain_seg(ch, (uint32_t)__builtin_amdgcn_readfirstlane((int)site));
            as.bs = ch->bs[sg]; as.mapq = ch->mapq[sg]; as.rpr = ch->rpr[sg];
        }
        bv_p2_sweep<BV_WAVE, true, true, false, BV_P2S_RW>(cx, as, site, lane);
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
                bv_p2_sweep<BV_WAVE, true, false, false, BV_P2S_RW>(cx, as, site, lane);
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

// ------------------------------------------------------------------------------ short rows, streamed through LDS
// The same work as bv_pass2_short_kernel for rows of 2,049 .. 49,152 samples, built like the short-row pass 1
// (bv_pass1_short.hip): every wave walks its share of the variant list as ONE sequence of slots -- 1 KiB of calls, 1 KiB of
// mapq, 2 KiB of read-position ranks -- through a private ring in LDS filled by LDS-DMA (no VGPR staging, counted waits, the
// next row already arriving while a row's rank sums are formed), and the per-cell work is cut to the tally's form:
//   * one v_perm_b32 per FOUR cells turns the call bytes into class bytes: the call byte itself is the selector into a
//     4-byte table (0x80 REF, 0x81 ALT, 0xFF neither; strand bit folded by giving both sources the same table); N / + / -
//     (8..10) select a sign-replication of table bytes whose top bit is always set -> 0xFF, garbage -> 0x00 / 0xFF: neither;
//   * after an XOR with 0x80808080, X = class << 8 | mapq and X' = rank_hi << 16 | class << 8 | rank_lo are glued by one
//     v_perm each; "X < 0x200" is the whole predicate (REF/ALT class, and rank < 256) and X the histogram word: 3 VALU + one
//     EXEC-predicated ds_add_u32 per cell and plane (bv_lds_add16), against ~20 VALU per cell in the branchy form.
// A row that holds a rank >= 256 anywhere (long reads; one OR per slot finds it) is re-done by the window sweeps of
// bv_pass2_short_kernel's code.  The per-site facts pass 1 left (depths, alt set, reference base) and the site indices are
// fetched for 64 sites at a time, so the ring is drained once per 64 sites, not per site.
#define BV_P2D_WAVES 4
#define BV_P2D_K 3
#define BV_P2D_SLOT_WORDS 1024  /* 1 KiB calls, 1 KiB mapq, 2 KiB ranks */
struct __attribute__((aligned(16))) BvPass2DmaShared {
    uint32_t h[BV_P2D_WAVES][4 * 256];                         // per wave: hm[2][256] then hr[2][256]
    uint32_t ring[BV_P2D_WAVES][BV_P2D_K][BV_P2D_SLOT_WORDS];
};

// TAG (BV_SLAB_RPR_TAGGED): a slot is three pieces -- 1 KiB of mapq, 2 KiB of ranks --, the class of a cell comes from the tag in
// its rank word (bv_p2t_class4) and the call plane is not read.
template <bool TAG>
__global__ __launch_bounds__(BV_WAVE *BV_P2D_WAVES) void bv_pass2_dma_kernel(BvPass2Args a) {
    constexpr int PIECES = TAG ? 3 : 4;  // LDS-DMA loads per slot: what the counted waits count
    __shared__ BvPass2DmaShared sh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *h = sh.h[wave];
    const uint32_t *ring = sh.ring[wave][0];
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(bv_lds_u32 *)sh.ring[wave][0]);
    const uint32_t n_var = a.counters[BV_CTR_VARIANTS];
    const uint32_t n_waves = gridDim.x * BV_P2D_WAVES, gw = blockIdx.x * BV_P2D_WAVES + (uint32_t)wave;
    if (gw >= n_var) return;
    const uint32_t mine = (n_var - gw + n_waves - 1u) / n_waves;  // variant sites of this wave: gw, gw + n_waves, ...
    const uint32_t n_chunks = (a.n_samples + 15u) >> 4, n_slots = (n_chunks + 63u) >> 6;
    const int tail = (int)(a.n_samples & 15u);
    const uint32_t last_valid = n_chunks - (n_slots - 1u) * 64u;
    const uint32_t voff = (uint32_t)lane * 16u;
    uint32_t one;
    asm volatile("v_mov_b32 %0, 1" : "=v"(one));

    // facts of 64 sites at a time (lane i: the wave's site number blk0 + i)
    uint32_t siteA = 0, siteB = 0;  // site indices of the current and of the next block of 64
    uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0, altw0 = 0, altw1 = 0, refv = 4;
    auto load_sites = [&](uint32_t blk) -> uint32_t {
        const uint32_t k = blk * 64u + (uint32_t)lane;
        return k < mine ? a.var_list[gw + k * n_waves] : 0u;
    };
    auto site_of = [&](uint32_t k, uint32_t blk0) -> uint32_t {  // k in [blk0, blk0 + 128)
        const uint32_t i = k - blk0;
        return (uint32_t)(i < 64u ? __builtin_amdgcn_readlane((int)siteA, (int)i) : __builtin_amdgcn_readlane((int)siteB, (int)(i - 64u)));
    };
    // prefetch cursor
    uint32_t p_k = 0, p_j = 0, ring_w = 0, inflight = 0, blk0 = 0;
    // a chained launch (a.ch): the planes of the segment that holds the prefetch cursor's site (biased: indexed with the
    // global site number); ref_base / out are contiguous either way
    const uint8_t *seg_bs = a.bs, *seg_mq = a.mapq, *seg_rp = reinterpret_cast<const uint8_t *>(a.rpr);
    auto issue = [&]() {
        if (p_k < mine) {
            const uint32_t site = site_of(p_k, blk0);
            if (a.ch != nullptr && p_j == 0u) {
                const BvChainC ch = bv_chain_const(a.ch);
                const uint32_t sg = bv_chain_seg(ch, site);
                seg_bs = ch->bs[sg]; seg_mq = ch->mapq[sg]; seg_rp = reinterpret_cast<const uint8_t *>(ch->rpr[sg]);
            }
            const size_t row = (size_t)site * a.pitch + (size_t)p_j * 1024u;
            const uint8_t *pb = bv_uniform_ptr(seg_bs + row), *pm = bv_uniform_ptr(seg_mq + row);
            const uint8_t *pr = bv_uniform_ptr(seg_rp + 2u * row);
            const uint32_t dst = ring_lds + ring_w * (BV_P2D_SLOT_WORDS * 4u);
            if (p_j + 1u < n_slots || (uint32_t)lane < last_valid) {  // lanes past the row's end load nothing (lane 0 always loads)
                if (!TAG) bv_glds16(dst, pb, voff);
                bv_glds16(dst + 1024u, pm, voff);
                bv_glds16(dst + 2048u, pr, voff * 2u);         // 32 bytes of ranks per lane: two 16-byte halves
                bv_glds16(dst + 3072u, pr + 16, voff * 2u);
            }
            ring_w = (ring_w + 1u == (uint32_t)BV_P2D_K) ? 0u : ring_w + 1u;
            ++inflight;
            if (++p_j == n_slots) { p_j = 0; ++p_k; }
        }
    };
    siteA = load_sites(0);
    siteB = load_sites(1);
    asm volatile("" : "+v"(siteA), "+v"(siteB)::"memory");
    uint32_t ring_r = 0;
    unsigned long long tw_m = 0, tw_r = 0, fast_mask = 0;  // lane i: twice the REF rank sums of the block's i-th site; bit i: pending
#pragma unroll 1
    for (uint32_t k = 0; k < mine; ++k) {
        if (k == blk0 + 64u) {  // next block of 64 sites (the prefetch cursor is at most one site ahead: n_slots >= K)
            blk0 += 64u;
            siteA = siteB;
            siteB = load_sites(blk0 / 64u + 1u);
            asm volatile("" : "+v"(siteA), "+v"(siteB)::"memory");
        }
        if (k == blk0) {
            // what pass 1 decided for these 64 sites (one drain of the ring per 64 sites)
            const uint32_t kk = k + (uint32_t)lane;
            const uint32_t s_ = kk < mine ? siteA : 0xFFFFFFFFu;
            if (s_ != 0xFFFFFFFFu) {
                const uint4 dd = *reinterpret_cast<const uint4 *>(&a.out[s_].depth[0]);
                const uint2 aw = *reinterpret_cast<const uint2 *>(&a.out[s_].n_alt);
                d0 = dd.x; d1 = dd.y; d2 = dd.z; d3 = dd.w; altw0 = aw.x; altw1 = aw.y;
                refv = a.ref_base[s_];
            }
            asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(altw0), "+v"(altw1), "+v"(refv)::"memory");
            if (k == 0) {
#pragma unroll 1
                for (int t = 0; t < BV_P2D_K; ++t) issue();
            }
        }
        const int li = (int)(k - blk0);
        const uint32_t site = (uint32_t)__builtin_amdgcn_readlane((int)siteA, li);
        int ref = __builtin_amdgcn_readlane((int)refv, li);
        if (ref > 4) ref = 4;
        const uint32_t depth[4] = {(uint32_t)__builtin_amdgcn_readlane((int)d0, li), (uint32_t)__builtin_amdgcn_readlane((int)d1, li),
                                   (uint32_t)__builtin_amdgcn_readlane((int)d2, li), (uint32_t)__builtin_amdgcn_readlane((int)d3, li)};
        const uint32_t aw0 = (uint32_t)__builtin_amdgcn_readlane((int)altw0, li), aw1 = (uint32_t)__builtin_amdgcn_readlane((int)altw1, li);
        const int n_alt = (int)(aw0 & 0xFFu);  // bytes at bv_site_result.n_alt: n_alt, alt[0..3]
        // class table: byte b = class of base b (0x80 REF, 0x81 ALT, 0xFF neither) and the plain 2-bit lut of the sweep code
        uint32_t L = 0xFFFFFFFFu, lut = 0xAAu;
        unsigned long long n1 = 0, n2 = 0;
        if (ref < 4) { L = (L & ~(0xFFu << (8 * ref))) | (0x80u << (8 * ref)); lut &= ~(3u << (2 * ref)); n1 = bv_sel4u(depth, ref); }
#pragma unroll
        for (int t = 0; t < BV_MAX_ALT; ++t) {
            if (t < n_alt) {
                const int b = (int)((t < 3 ? (aw0 >> (8 * (t + 1))) : aw1) & 3u);
                L = (L & ~(0xFFu << (8 * b))) | (0x81u << (8 * b));
                lut = (lut & ~(3u << (2 * b))) | (1u << (2 * b));
                n2 += bv_sel4u(depth, b);
            }
        }
        {
            uint4 *z = reinterpret_cast<uint4 *>(h);
#pragma unroll
            for (int i = 0; i < 4 * 256 / 4 / BV_WAVE; ++i) z[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
        }
        uint32_t hi_acc = 0, dom = BV_DOM_NONE;
        const bool deep = (n1 + n2) * 8ull >= (unsigned long long)a.n_samples && !(a.flags & BV_FLAG_NO_DOM);
#pragma unroll 1
        for (uint32_t j = 0; j < n_slots; ++j) {
            if (inflight == (uint32_t)BV_P2D_K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (BV_P2D_K - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t *sl = ring + ring_r * BV_P2D_SLOT_WORDS;
            bv_u32x4 vb = bv_u32x4{0u, 0u, 0u, 0u};
            if (!TAG) vb = *reinterpret_cast<const bv_u32x4 *>(sl + lane * 4);
            const bv_u32x4 vm = *reinterpret_cast<const bv_u32x4 *>(sl + 256 + lane * 4);
            bv_u32x4 r0 = *reinterpret_cast<const bv_u32x4 *>(sl + 512 + lane * 4);   // ranks 0-7
            bv_u32x4 r1 = *reinterpret_cast<const bv_u32x4 *>(sl + 768 + lane * 4);   // ranks 8-15
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ring_r = (ring_r + 1u == (uint32_t)BV_P2D_K) ? 0u : ring_r + 1u;
            --inflight;
            issue();
            if (j + 1u == n_slots) {
                const uint32_t chunk = j * 64u + (uint32_t)lane;
                if (chunk >= n_chunks) {  // not loaded: stale bytes
                    vb = bv_u32x4{0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u};
                    r0 = TAG ? bv_u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u} : bv_u32x4{0u, 0u, 0u, 0u};
                    r1 = r0;
                } else if (tail && chunk == n_chunks - 1) {
                    if (TAG) {
                        r0.x = bv_p2t_mask_tail(r0.x, tail); r0.y = bv_p2t_mask_tail(r0.y, tail - 2);
                        r0.z = bv_p2t_mask_tail(r0.z, tail - 4); r0.w = bv_p2t_mask_tail(r0.w, tail - 6);
                        r1.x = bv_p2t_mask_tail(r1.x, tail - 8); r1.y = bv_p2t_mask_tail(r1.y, tail - 10);
                        r1.z = bv_p2t_mask_tail(r1.z, tail - 12); r1.w = bv_p2t_mask_tail(r1.w, tail - 14);
                    } else {
                        vb.x = bv_mask_tail_dword(vb.x, tail); vb.y = bv_mask_tail_dword(vb.y, tail - 4);
                        vb.z = bv_mask_tail_dword(vb.z, tail - 8); vb.w = bv_mask_tail_dword(vb.w, tail - 12);
                    }
                }
            }
            // class bytes of the 16 cells (REF 0x00, ALT 0x01, neither >= 0x7F)
            uint32_t c0, c1, c2, c3;
            if (TAG) {
                c0 = bv_p2t_class4(L, r0.x, r0.y); c1 = bv_p2t_class4(L, r0.z, r0.w);
                c2 = bv_p2t_class4(L, r1.x, r1.y); c3 = bv_p2t_class4(L, r1.z, r1.w);
            } else {
                c0 = __builtin_amdgcn_perm(L, L, vb.x) ^ 0x80808080u; c1 = __builtin_amdgcn_perm(L, L, vb.y) ^ 0x80808080u;
                c2 = __builtin_amdgcn_perm(L, L, vb.z) ^ 0x80808080u; c3 = __builtin_amdgcn_perm(L, L, vb.w) ^ 0x80808080u;
            }
            // ranks that do not fit the 256-rank window: remembered, the row is then re-done by sweeps (valid data has rank 0
            // in uncovered cells)
            hi_acc |= (r0.x | r0.y | r0.z | r0.w | r1.x | r1.y | r1.z | r1.w) & (TAG ? 0x1F001F00u : 0xFF00FF00u);
            uint32_t x[16];
            x[0] = bv_p2d_xm<0>(c0, vm.x); x[1] = bv_p2d_xm<1>(c0, vm.x); x[2] = bv_p2d_xm<2>(c0, vm.x); x[3] = bv_p2d_xm<3>(c0, vm.x);
            x[4] = bv_p2d_xm<0>(c1, vm.y); x[5] = bv_p2d_xm<1>(c1, vm.y); x[6] = bv_p2d_xm<2>(c1, vm.y); x[7] = bv_p2d_xm<3>(c1, vm.y);
            x[8] = bv_p2d_xm<0>(c2, vm.z); x[9] = bv_p2d_xm<1>(c2, vm.z); x[10] = bv_p2d_xm<2>(c2, vm.z); x[11] = bv_p2d_xm<3>(c2, vm.z);
            x[12] = bv_p2d_xm<0>(c3, vm.w); x[13] = bv_p2d_xm<1>(c3, vm.w); x[14] = bv_p2d_xm<2>(c3, vm.w); x[15] = bv_p2d_xm<3>(c3, vm.w);
            uint32_t y[16];
            y[0] = bv_p2d_xr<0, 0>(c0, r0.x); y[1] = bv_p2d_xr<1, 1>(c0, r0.x); y[2] = bv_p2d_xr<2, 0>(c0, r0.y); y[3] = bv_p2d_xr<3, 1>(c0, r0.y);
            y[4] = bv_p2d_xr<0, 0>(c1, r0.z); y[5] = bv_p2d_xr<1, 1>(c1, r0.z); y[6] = bv_p2d_xr<2, 0>(c1, r0.w); y[7] = bv_p2d_xr<3, 1>(c1, r0.w);
            y[8] = bv_p2d_xr<0, 0>(c2, r1.x); y[9] = bv_p2d_xr<1, 1>(c2, r1.x); y[10] = bv_p2d_xr<2, 0>(c2, r1.y); y[11] = bv_p2d_xr<3, 1>(c2, r1.y);
            y[12] = bv_p2d_xr<0, 0>(c3, r1.z); y[13] = bv_p2d_xr<1, 1>(c3, r1.z); y[14] = bv_p2d_xr<2, 0>(c3, r1.w); y[15] = bv_p2d_xr<3, 1>(c3, r1.w);
            if (deep) {  // (a deep row: the dominant mapq's lanes are counted, not added one by one)
                bv_lds_add16_dom<2>(x, h, one, 0x200u, dom);
                bv_lds_add16<2>(y, h + 512, one, 0x200u);
            } else bv_lds_add16x2<2>(x, y, h, h + 512, one, 0x200u);  // (both histograms under one predicate: the class byte is the same in x and y)
        }
        bv_lrt_sync<0>();
        uint32_t *hm = h, *hr = h + 512;
        if (__ballot(hi_acc != 0u) != 0ull) {
            // a rank >= 256 somewhere in the row: the exact window sweeps (plain loads; rare -- long reads)
            {
                uint4 *z = reinterpret_cast<uint4 *>(h);
#pragma unroll
                for (int i = 0; i < 4 * 256 / 4 / BV_WAVE; ++i) z[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
            }
            bv_lrt_sync<0>();
            BvP2Ctx cx;
            cx.hm = hm; cx.hr = hr; cx.hg = nullptr; cx.lut = lut; cx.win_lo = 0; cx.n_groups = 0; cx.maxr = 0; cx.half = false;
            cx.rmask = bv_rpr_rank_mask(a.rpr_tag);
            BvPass2Args as = a;  // the sweeps index the planes with the site number themselves
            if (a.ch != nullptr) {
                const BvChainC ch = bv_chain_const(a.ch);
                const uint32_t sg = bv_chain_seg(ch, site);
                as.bs = ch->bs[sg]; as.mapq = ch->mapq[sg]; as.rpr = ch->rpr[sg];
            }
            bv_p2_sweep<BV_WAVE, true, true, false, 256>(cx, as, site, lane);
            const uint32_t maxr = (uint32_t)bv_wave_max_i32((int)cx.maxr);
            bv_lrt_sync<0>();
            {
                unsigned long long below = 0, twoR = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) twoR += bv_ranksum_window(hm[w * 64 + lane], hm[256 + w * 64 + lane], n1 + n2, below, lane);
                const double ph = bv_ranksum_phred(twoR, n1, n2);
                if (lane == 0) a.out[site].mq_ranksum = ph;
            }
            unsigned long long below = 0, twoR = 0;
            for (uint32_t win_lo = 0;; win_lo += 256u) {
                const int nblk = (maxr < win_lo + 256u) ? (int)((maxr - win_lo) >> 6) + 1 : 4;
                for (int w = 0; w < nblk; ++w) twoR += bv_ranksum_window(hr[w * 64 + lane], hr[256 + w * 64 + lane], n1 + n2, below, lane);
                if (maxr < win_lo + 256u) break;
                bv_lrt_sync<0>();
                {
                    uint4 *z = reinterpret_cast<uint4 *>(hr);
#pragma unroll
                    for (int i = 0; i < 2 * 256 / 4 / BV_WAVE; ++i) z[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
                }
                bv_lrt_sync<0>();
                cx.win_lo = win_lo + 256u;
                bv_p2_sweep<BV_WAVE, true, false, false, 256>(cx, as, site, lane);
                bv_lrt_sync<0>();
            }
            const double ph = bv_ranksum_phred(twoR, n1, n2);
            if (lane == 0) {
                a.out[site].rpr_ranksum = ph;
                atomicOr(&a.out[site].status, BV_SITE_RANKSUM);
            }
        } else {
            // the two rank sums as exact integers; their phred values (erfc, log10: scalar work) are formed for 64 sites at a
            // time, one site per lane, when the block of sites ends
            unsigned long long below = 0, twoR = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) twoR += bv_ranksum_window(hm[w * 64 + lane], hm[256 + w * 64 + lane], n1 + n2, below, lane);
            tw_m = (lane == li) ? twoR : tw_m;
            below = 0; twoR = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) twoR += bv_ranksum_window(hr[w * 64 + lane], hr[256 + w * 64 + lane], n1 + n2, below, lane);
            tw_r = (lane == li) ? twoR : tw_r;
            fast_mask |= 1ull << li;
        }
        if (li == 63 || k + 1u == mine) {
            if ((fast_mask >> lane) & 1ull) {
                // this lane's site: n1 / n2 from its own facts (the same arithmetic as above, per lane)
                int r_ = (int)refv;
                if (r_ > 4) r_ = 4;
                const uint32_t dl[4] = {d0, d1, d2, d3};
                unsigned long long m1 = (r_ < 4) ? bv_sel4u(dl, r_) : 0ull, m2 = 0;
                const int na = (int)(altw0 & 0xFFu);
#pragma unroll
                for (int t = 0; t < BV_MAX_ALT; ++t)
                    if (t < na) m2 += bv_sel4u(dl, (int)((t < 3 ? (altw0 >> (8 * (t + 1))) : altw1) & 3u));
                const double ph_m = bv_ranksum_phred(tw_m, m1, m2), ph_r = bv_ranksum_phred(tw_r, m1, m2);
                a.out[siteA].mq_ranksum = ph_m;
                a.out[siteA].rpr_ranksum = ph_r;
                atomicOr(&a.out[siteA].status, BV_SITE_RANKSUM);
            }
            fast_mask = 0ull;
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
    const bool items = bv_p2g_all_items(a);
    // many groups on rows whose counts fit 16 bits: half the histogram, more workgroups per CU -- from 14 groups on, where 32-bit
    // counters leave room for three workgroups (13 KiB + 2 KiB per group of 160); below that the three extra VALU per cell cost
    // more than the occupancy brings (100 k sites x 10 k samples: 8 groups -3 %, 12 +-0, 14 +4 %, 16 +4.5 %, 32 +9 %)
    if (NT == 256 && groups && items && a.n_groups >= 14u && a.n_samples <= 65535u) {
        dyn /= 2;
        if (ranks) hipLaunchKernelGGL((bv_pass2_kernel<256, true, true, false, true>), dim3(grid), dim3(256), dyn, stream, a);
        else hipLaunchKernelGGL((bv_pass2_kernel<256, false, true, false, true>), dim3(grid), dim3(256), dyn, stream, a);
        return;
    }
    if (ranks && groups && items)
        hipLaunchKernelGGL((bv_pass2_kernel<NT, true, true, false>), dim3(grid), dim3(NT), dyn, stream, a);
    else if (ranks && groups)
        hipLaunchKernelGGL((bv_pass2_kernel<NT, true, true, true>), dim3(grid), dim3(NT), dyn, stream, a);
    else if (ranks && a.rpr_tag)
        hipLaunchKernelGGL((bv_pass2_kernel<NT, true, false, true, false, true>), dim3(grid), dim3(NT), dyn, stream, a);
    else if (ranks)
        hipLaunchKernelGGL((bv_pass2_kernel<NT, true, false>), dim3(grid), dim3(NT), dyn, stream, a);
    else if (items)
        hipLaunchKernelGGL((bv_pass2_kernel<NT, false, true, false>), dim3(grid), dim3(NT), dyn, stream, a);
    else
        hipLaunchKernelGGL((bv_pass2_kernel<NT, false, true, true>), dim3(grid), dim3(NT), dyn, stream, a);
}

void bv_launch_pass2(const BvPass2Args &a_in, hipStream_t stream) {
    BvPass2Args a = a_in;
    const bool ranks = a.mapq != nullptr && a.rpr != nullptr;
    bool groups = a.n_groups > 0 && a.group_id != nullptr && a.gout != nullptr;
    if (bv_p2g_streams(a)) {
        // short rows with pop-groups: the group tallies stream on their own (2 B per cell), the rank sums below as if
        // there were no groups (4 B per cell); the items are solved by bv_launch_p2g_solve16
        bv_launch_p2g_stream(a, stream);
        groups = false;
        a.n_groups = 0; a.group_id = nullptr; a.gout = nullptr;
        if (!ranks) return;
    }
    if (a.n_samples > 2048u && a.n_samples <= BV_SHORT_ROW_MAX && ranks && !groups && !(a.flags & BV_FLAG_PASS2_SWEEP)) {
        uint32_t grid = (a.n_cu ? a.n_cu : 256u) * 2u;  // 64 KiB of LDS per workgroup: 2 per CU, 8 waves
        const uint32_t need = (a.n_sites + BV_P2D_WAVES - 1) / BV_P2D_WAVES;
        if (grid > need) grid = need;
        const uint32_t cap = (a.flags >> 16) & 0xFFu;  // BV_FLAG_GRID_LIMIT
        if (cap && grid > cap) grid = cap;
        if (a.rpr_tag) hipLaunchKernelGGL(bv_pass2_dma_kernel<true>, dim3(grid), dim3(BV_WAVE * BV_P2D_WAVES), 0, stream, a);
        else hipLaunchKernelGGL(bv_pass2_dma_kernel<false>, dim3(grid), dim3(BV_WAVE * BV_P2D_WAVES), 0, stream, a);
        return;
    }
    if (a.n_samples <= 16384u && ranks && !groups) {
        uint32_t grid = (a.n_cu ? a.n_cu : 256u) * 4u;  // 16 KiB of LDS and <= 128 VGPRs: 4 workgroups per CU
        const uint32_t need = (a.n_sites + BV_P2S_WAVES - 1) / BV_P2S_WAVES;
        if (grid > need) grid = need;
        hipLaunchKernelGGL(bv_pass2_short_kernel, dim3(grid), dim3(BV_WAVE * BV_P2S_WAVES), 0, stream, a);
        return;
    }
    if (a.n_samples <= 16384u && !(groups && a.n_groups >= BV_P2_WIDE_GROUPS)) bv_launch_pass2_nt<64>(a, stream);
    // many pop-groups: 2 KiB of histogram per group leave room for two workgroups per CU only -- eight waves each then
    // (sixteen waves were measured too: 10 % slower than eight at 24-32 groups)
    else if (groups && a.n_groups >= BV_P2_BIG_GROUPS && bv_p2g_all_items(a) && a.n_samples > 16384u) bv_launch_pass2_nt<512>(a, stream);
    else bv_launch_pass2_nt<256>(a, stream);
}
