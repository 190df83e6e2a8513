// bv_pass2_groups.hip -- the pop-group calls of pass 2 (__gb(): BaseType(subset) + lrt([REF] + alts),
// src/basetype_caller.cpp:756-759, 767-797) as kernels of their own:
//   bv_p2g_stream_kernel   short rows: the per-group (base, phred) histograms of every variant row through an LDS-DMA ring
//   bv_p2g_solve16_kernel  the group LRTs, four per wave (bv_solver16.h), for the items the tally kernels hand over
//   bv_p2g_hard_kernel     the items that need the one-wave solver: phred-0 calls, more than 128 bins, min_af <= 0, and the
//                          shallow groups (<= 64 covered samples) in which the four-per-wave solver met a tie: replayed in the
//                          reference's per-sample order
// The workgroup-per-row tally of long rows lives in bv_pass2.hip (bv_pass2_kernel<.., GROUPS>); it hands its groups over
// in the same item format (BvPass2Args::gitems, bv_kernels.h).
#include "bv_kernels.h"
#include "bv_solver16.h"
#include "bv_tally.h"

extern __shared__ __attribute__((aligned(16))) uint32_t bv_dyn_lds[];  // bv_p2g_stream_kernel: per wave [group][base][128]

// ------------------------------------------------------------------------------ short rows: pop-group tallies by LDS-DMA
// The (base, phred) histogram of every pop-group of every variant site, rows of 2049 ... 49152 samples and up to
// BV_P2GS_MAX_GROUPS groups.  Same structure as bv_pass2_dma_kernel: a persistent grid, every wave walks its share of the
// variant list with a ring of K slots in flight across rows; a slot is 1 KiB of calls, 1 KiB of phreds and the matching 1 KiB of
// the prepared group plane (g << 2, or 0x80 for samples in no group; the same 10-50 KB for every row: L2 hits).
// Per cell: X = (group << 2 | base) << 8 | phred << 1, twice the word index of the wave's [group][base][128] histogram, with
// bit 15 set for no-call cells and samples without a group -- one v_perm per cell on top of three per-dword operations, then the
// predicated ds_add batch of bv_tally.h.  After a row, every group's bins leave as an item for the solve kernels (the LRTs are
// not done here: the wave goes on streaming).
#define BV_P2GS_WAVES 4
#define BV_P2GS_SLOT_WORDS 768
#define BV_P2GS_MAX_GROUPS 7   /* beyond: too few waves fit the LDS; the workgroup-per-row kernel shares one histogram set between four waves */
template <int K>
struct __attribute__((aligned(16))) BvP2gsShared {
    uint32_t ring[BV_P2GS_WAVES][K][BV_P2GS_SLOT_WORDS];
};
template <int K>
__global__ __launch_bounds__(BV_WAVE *BV_P2GS_WAVES) void bv_p2g_stream_kernel(BvPass2Args a) {
    __shared__ BvP2gsShared<K> sh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t G = a.n_groups;
    uint32_t *hist = bv_dyn_lds + (size_t)wave * G * 512u;  // [group][base][phred < 128]
    const uint32_t *ring = sh.ring[wave][0];
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(bv_lds_u32 *)sh.ring[wave][0]);
    const uint32_t n_var = a.counters[BV_CTR_VARIANTS];
    const uint32_t n_waves = gridDim.x * BV_P2GS_WAVES, gw = blockIdx.x * BV_P2GS_WAVES + (uint32_t)wave;
    if (gw >= n_var) return;
    const uint32_t mine = (n_var - gw + n_waves - 1u) / n_waves;  // variant sites of this wave: gw, gw + n_waves, ...
    const uint32_t n_chunks = (a.n_samples + 15u) >> 4, n_slots = (n_chunks + 63u) >> 6;
    const int tail = (int)(a.n_samples & 15u);
    const uint32_t last_valid = n_chunks - (n_slots - 1u) * 64u;
    const uint32_t voff = (uint32_t)lane * 16u;
    uint32_t one;
    asm volatile("v_mov_b32 %0, 1" : "=v"(one));
    {
        uint4 *z = reinterpret_cast<uint4 *>(hist);
        for (uint32_t i = 0; i < 2u * G; ++i) z[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
    }

    uint32_t siteA = 0, siteB = 0;  // lane i: site of this wave's variant number blk0 + i, blk0 + 64 + i
    auto load_sites = [&](uint32_t blk) -> uint32_t {
        const uint32_t k = blk * 64u + (uint32_t)lane;
        return k < mine ? a.var_list[gw + k * n_waves] : 0u;
    };
    auto site_of = [&](uint32_t k, uint32_t blk0) -> uint32_t {  // k in [blk0, blk0 + 128)
        const uint32_t i = k - blk0;
        return (uint32_t)(i < 64u ? __builtin_amdgcn_readlane((int)siteA, (int)i) : __builtin_amdgcn_readlane((int)siteB, (int)(i - 64u)));
    };
    uint32_t p_k = 0, p_j = 0, ring_w = 0, inflight = 0, blk0 = 0;  // prefetch cursor
    const uint8_t *seg_bs = a.bs, *seg_q = a.q;  // a chained launch: the (biased) planes of the segment that holds the cursor's site
    auto issue = [&]() {
        if (p_k < mine) {
            const uint32_t site = site_of(p_k, blk0);
            if (a.ch != nullptr && p_j == 0u) {
                const BvChainC ch = bv_chain_const(a.ch);
                const uint32_t sg = bv_chain_seg(ch, site);
                seg_bs = ch->bs[sg]; seg_q = ch->q[sg];
            }
            const size_t row = (size_t)site * a.pitch + (size_t)p_j * 1024u;
            const uint8_t *pb = bv_uniform_ptr(seg_bs + row), *pq = bv_uniform_ptr(seg_q + row);
            const uint8_t *pg = bv_uniform_ptr(a.gidp + (size_t)p_j * 1024u);
            const uint32_t dst = ring_lds + ring_w * (BV_P2GS_SLOT_WORDS * 4u);
            if (p_j + 1u < n_slots || (uint32_t)lane < last_valid) {  // lanes past the row's end load nothing (lane 0 always loads)
                bv_glds16(dst, pb, voff);
                bv_glds16(dst + 1024u, pq, voff);
                bv_glds16(dst + 2048u, pg, voff);
            }
            ring_w = (ring_w + 1u == (uint32_t)K) ? 0u : ring_w + 1u;
            ++inflight;
            if (++p_j == n_slots) { p_j = 0; ++p_k; }
        }
    };
    siteA = load_sites(0);
    siteB = load_sites(1);
    asm volatile("" : "+v"(siteA), "+v"(siteB)::"memory");
    uint32_t ring_r = 0;
#pragma unroll 1
    for (uint32_t k = 0; k < mine; ++k) {
        if (k == blk0 + 64u) {  // next block of 64 sites (the prefetch cursor is at most one site ahead: n_slots >= K)
            blk0 += 64u;
            siteA = siteB;
            siteB = load_sites(blk0 / 64u + 1u);
            asm volatile("" : "+v"(siteA), "+v"(siteB)::"memory");
        }
        if (k == 0) {
#pragma unroll 1
            for (int t = 0; t < K; ++t) issue();
        }
#pragma unroll 1
        for (uint32_t j = 0; j < n_slots; ++j) {
            if (inflight == (uint32_t)K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (K - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const uint32_t *sl = ring + ring_r * BV_P2GS_SLOT_WORDS;
            bv_u32x4 vb = *reinterpret_cast<const bv_u32x4 *>(sl + lane * 4);
            bv_u32x4 vq = *reinterpret_cast<const bv_u32x4 *>(sl + 256 + lane * 4);
            const bv_u32x4 vg = *reinterpret_cast<const bv_u32x4 *>(sl + 512 + lane * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ring_r = (ring_r + 1u == (uint32_t)K) ? 0u : ring_r + 1u;
            --inflight;
            issue();
            if (j + 1u == n_slots) {
                const uint32_t chunk = j * 64u + (uint32_t)lane;
                if (chunk >= n_chunks) {  // not loaded: stale bytes
                    vb = bv_u32x4{0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u};
                    vq = bv_u32x4{0u, 0u, 0u, 0u};
                } else if (tail && chunk == n_chunks - 1) {
                    vb.x = bv_mask_tail_dword(vb.x, tail); vb.y = bv_mask_tail_dword(vb.y, tail - 4);
                    vb.z = bv_mask_tail_dword(vb.z, tail - 8); vb.w = bv_mask_tail_dword(vb.w, tail - 12);
                }
            }
            // call bytes above 15 and phred bytes above 127 (invalid input) do not fit the packed index: cell by cell then
            const uint32_t odd = ((vb.x | vb.y | vb.z | vb.w) & 0xF0F0F0F0u) | ((vq.x | vq.y | vq.z | vq.w) & 0x80808080u);
            if (__builtin_expect(__ballot(odd != 0u) != 0ull, 0)) {
                const uint32_t wb[4] = {vb.x, vb.y, vb.z, vb.w}, wq[4] = {vq.x, vq.y, vq.z, vq.w}, wg[4] = {vg.x, vg.y, vg.z, vg.w};
#pragma unroll 1
                for (int t = 0; t < 16; ++t) {
                    const uint32_t c = (wb[t >> 2] >> (8 * (t & 3))) & 0xFFu, p = (wq[t >> 2] >> (8 * (t & 3))) & 0xFFu;
                    const uint32_t gp = (wg[t >> 2] >> (8 * (t & 3))) & 0xFFu;
                    if (c < 8u && !(gp & 0x80u)) atomicAdd(&hist[((gp | (c & 3u)) << 7) | min(p, 127u)], 1u);
                }
                continue;
            }
            // byte = group << 2 | base, bit 7 for "no call" (call bit 3) or "no group" (0x80 in the prepared plane)
            const uint32_t y0 = (((vb.x & 0x08080808u) << 4) | (vb.x & 0x03030303u)) | vg.x, y1 = (((vb.y & 0x08080808u) << 4) | (vb.y & 0x03030303u)) | vg.y;
            const uint32_t y2 = (((vb.z & 0x08080808u) << 4) | (vb.z & 0x03030303u)) | vg.z, y3 = (((vb.w & 0x08080808u) << 4) | (vb.w & 0x03030303u)) | vg.w;
            vq.x <<= 1; vq.y <<= 1; vq.z <<= 1; vq.w <<= 1;
            uint32_t x[16];
            x[0] = bv_cell_index<0>(y0, vq.x); x[1] = bv_cell_index<1>(y0, vq.x); x[2] = bv_cell_index<2>(y0, vq.x); x[3] = bv_cell_index<3>(y0, vq.x);
            x[4] = bv_cell_index<0>(y1, vq.y); x[5] = bv_cell_index<1>(y1, vq.y); x[6] = bv_cell_index<2>(y1, vq.y); x[7] = bv_cell_index<3>(y1, vq.y);
            x[8] = bv_cell_index<0>(y2, vq.z); x[9] = bv_cell_index<1>(y2, vq.z); x[10] = bv_cell_index<2>(y2, vq.z); x[11] = bv_cell_index<3>(y2, vq.z);
            x[12] = bv_cell_index<0>(y3, vq.w); x[13] = bv_cell_index<1>(y3, vq.w); x[14] = bv_cell_index<2>(y3, vq.w); x[15] = bv_cell_index<3>(y3, vq.w);
            bv_lds_add16<1>(x, hist, one, 0x8000u);
        }
        bv_lrt_sync<0>();

        // ---- every group's bins -> its item
#pragma unroll 1
        for (uint32_t g = 0; g < G; ++g) {
            uint32_t *h = hist + g * 512u;
            uint32_t c[4][2], dpart[4], q0_mask = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                c[b][0] = h[(b << 7) | lane];
                c[b][1] = h[(b << 7) | 64 | lane];
                dpart[b] = c[b][0] + c[b][1];
                if (__builtin_amdgcn_readfirstlane((int)c[b][0]) != 0) q0_mask |= 1u << b;  // phred-0 calls of base b
            }
            {
                uint4 *z = reinterpret_cast<uint4 *>(h);
                z[lane] = make_uint4(0, 0, 0, 0);
                z[BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
            }
            uint32_t gdepth[4];
            {
                const uint32_t v8[8] = {dpart[0], dpart[1], dpart[2], dpart[3], 0u, 0u, 0u, 0u};
                uint32_t t8[8];
                bv_wave_sum8_u32(v8, t8, lane);
                gdepth[0] = t8[0]; gdepth[1] = t8[1]; gdepth[2] = t8[2]; gdepth[3] = t8[3];
            }
            const uint32_t gtotal = gdepth[0] + gdepth[1] + gdepth[2] + gdepth[3];
            const size_t item = ((size_t)gw + (size_t)k * n_waves) * G + g;
            uint32_t *dst = a.gitems + item * BV_P2G_ITEM_WORDS;
            uint32_t nb = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int qr = 0; qr < 2; ++qr) nb += (uint32_t)__popcll(__ballot(c[b][qr] != 0u && ((qr << 6) | lane) < BV_NQ_VALID));
            }
            const int n_seen = (gdepth[0] != 0) + (gdepth[1] != 0) + (gdepth[2] != 0) + (gdepth[3] != 0);
            const bool shallow = gtotal <= (uint32_t)BV_ORD_MAX && n_seen >= 2;
            // (a shallow group goes to the four-per-wave solver first; it comes back as a HARD item when that meets a tie)
            const bool four = q0_mask == 0u && nb <= (uint32_t)BV_G16_MAX_BINS && a.min_af > 0.0 && !(a.flags & BV_FLAG_WAVE_SOLVER);
            if (gtotal != 0u) {
                uint32_t pos0 = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int qr = 0; qr < 2; ++qr) {
                        const int q = (qr << 6) | lane;
                        const bool valid = c[b][qr] != 0u && q < BV_NQ_VALID;
                        const unsigned long long m = __ballot(valid);
                        const uint32_t pos = pos0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                        const uint32_t code = ((uint32_t)b << 7) | (uint32_t)q;
                        if (valid) dst[8u + pos] = four ? ((code << 16) | c[b][qr]) : ((code << 23) | c[b][qr]);
                        pos0 += (uint32_t)__popcll(m);
                    }
                }
            }
            const uint32_t hdr = gtotal == 0u ? 0u : (nb | (four ? BV_P2G_PENDING : BV_P2G_HARD) | (shallow ? BV_P2G_SHALLOW : 0u));
            uint32_t w = hdr;
#pragma unroll
            for (int b = 0; b < 4; ++b) w = (lane == 1 + b) ? gdepth[b] : w;
            w = (lane == 5) ? q0_mask : w;
            if (lane < 6) dst[lane] = w;
        }
        bv_lrt_sync<0>();
    }
}

// ---- the items the 16-lane solver cannot take: one wave per item, the solver of bv_pass2_kernel
#define BV_P2GH_WAVES 4
struct __attribute__((aligned(16))) BvP2ghShared {
    uint32_t bin_code[BV_P2GH_WAVES][BV_SLOTS * BV_WAVE];
    uint32_t bin_cnt[BV_P2GH_WAVES][BV_SLOTS * BV_WAVE];
    BvLrtShared lrt[BV_P2GH_WAVES];
    double tab_hit[BV_QBINS], tab_miss[BV_QBINS];
    alignas(8) uint16_t ord[BV_P2GH_WAVES][BV_ORD_ALLOC];
};
__global__ __launch_bounds__(BV_WAVE *BV_P2GH_WAVES) void bv_p2g_hard_kernel(BvPass2Args a) {
    __shared__ BvP2ghShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < BV_QBINS; i += BV_WAVE * BV_P2GH_WAVES) {
        sh.tab_hit[i] = a.tables->hit[i];
        sh.tab_miss[i] = a.tables->miss[i];
    }
    __syncthreads();
    const uint32_t n_var = a.counters[BV_CTR_VARIANTS];
    const uint64_t all = (uint64_t)n_var * a.n_groups;
    const uint32_t n_items = all < (uint64_t)a.gitem_cap ? (uint32_t)all : a.gitem_cap;
    const uint32_t n_waves = gridDim.x * BV_P2GH_WAVES, gw = blockIdx.x * BV_P2GH_WAVES + (uint32_t)wave;
    // this wave's items: gw, gw + n_waves, ... (interleaved, so that runs of hard items spread over the grid); their headers
    // are read 64 at a time
    for (uint32_t t = 0; (uint64_t)gw + (uint64_t)t * 64u * n_waves < (uint64_t)n_items; ++t) {
        const uint32_t mine = gw + (t * 64u + (uint32_t)lane) * n_waves;  // (items number at most a few million: no overflow)
        const uint32_t hdr_l = mine < n_items ? a.gitems[(size_t)mine * BV_P2G_ITEM_WORDS] : 0u;
        unsigned long long todo = __ballot((hdr_l & BV_P2G_HARD) != 0u);
        while (todo) {
            const int li = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t idx = gw + (t * 64u + (uint32_t)li) * n_waves;
            const uint32_t hdr = (uint32_t)__builtin_amdgcn_readlane((int)hdr_l, li);
            const uint32_t *it = a.gitems + (size_t)idx * BV_P2G_ITEM_WORDS;
            const uint32_t nb = hdr & 0xFFFFu;
            const uint32_t v = idx / a.n_groups, g = idx - v * a.n_groups;
            const uint32_t site = a.var_list[v];
            // a chained launch: the segment's records, reference bases, planes (the wave works on one item: uniform)
            const bv_site_result *outp = a.out;
            const uint8_t *refp = a.ref_base, *bsp = a.bs, *qp = a.q;
            bv_group_result *goutp = a.gout;
            if (a.ch != nullptr) {
                const BvChainC ch = bv_chain_const(a.ch);
                const uint32_t sg = bv_chain_seg(ch, (uint32_t)__builtin_amdgcn_readfirstlane((int)site));
                if (!a.ch_cat) { outp = ch->out[sg]; refp = ch->ref_base[sg]; }
                bsp = ch->bs[sg]; qp = ch->q[sg]; goutp = ch->gout[sg];
            }
            const bv_site_result *res = &outp[site];
            int ref = refp[site];
            if (ref > 4) ref = 4;
            const int n_alt = res->n_alt;
            int comb = ref, nc = 1;  // caller.cpp:750-753: [toupper(REF)] + alts, 3 bits per entry
#pragma unroll
            for (int k = 0; k < BV_MAX_ALT; ++k) {
                if (k < n_alt) {
                    comb |= (res->alt[k] & 3) << (3 * nc);
                    ++nc;
                }
            }
            uint32_t gdepth[4] = {it[1], it[2], it[3], it[4]};
            const uint32_t gtotal = gdepth[0] + gdepth[1] + gdepth[2] + gdepth[3], q0_mask = it[5];
            for (uint32_t i = (uint32_t)lane; i < nb; i += BV_WAVE) {
                const uint32_t w = it[8u + i];  // code << 23 | count
                sh.bin_code[wave][i] = w >> 23;
                sh.bin_cnt[wave][i] = w & 0x7FFFFFu;
            }
            bv_lrt_sync<0>();
            BvBins B;
            B.code = sh.bin_code[wave]; B.cnt = sh.bin_cnt[wave]; B.skip_mask = 0u;
            B.hit = sh.tab_hit; B.miss = sh.tab_miss; B.nb = (int)nb;
            B.loghit = a.tables->loghit; B.logmiss = a.tables->logmiss; B.ord = nullptr; B.n_ord = 0;
            if (hdr & BV_P2G_SHALLOW) {
                const uint32_t got = bv_gather_ordered(bsp + (size_t)site * a.pitch, qp + (size_t)site * a.pitch, a.n_samples,
                                                       sh.ord[wave], lane, a.group_id, g);
                bv_lrt_sync<0>();
                if (got == gtotal) { B.ord = sh.ord[wave]; B.n_ord = (int)gtotal; }
            }
            BvLrtOut L;
            bv_lrt<0>(B, gdepth, gtotal, comb, nc, ref, a.min_af, &sh.lrt[wave], wave, lane, L, q0_mask);
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
                goutp[(size_t)site * a.n_groups + g] = gr;
            }
            bv_lrt_sync<0>();
        }
    }
}

__global__ void bv_gid_prepare_kernel(const uint8_t *gid, uint8_t *gidp, uint32_t n, uint32_t n_groups) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t g = gid[i];
        gidp[i] = g < n_groups ? (uint8_t)(g << 2) : (uint8_t)0x80u;
    }
}
__global__ void bv_gid_round_kernel(const uint8_t *gid, uint8_t *out, uint32_t n, uint32_t lo, uint32_t cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint32_t g = (uint32_t)gid[i] - lo;  // (wraps for groups below lo)
        out[i] = g < cnt ? (uint8_t)g : (uint8_t)BV_NO_GROUP;
    }
}
void bv_launch_gid_round(const uint8_t *gid, uint8_t *out, uint32_t n_bytes, uint32_t lo, uint32_t n, hipStream_t stream) {
    hipLaunchKernelGGL(bv_gid_round_kernel, dim3((n_bytes + 255u) / 256u), dim3(256), 0, stream, gid, out, n_bytes, lo, n);
}
void bv_launch_gid_prepare(const uint8_t *gid, uint8_t *gidp, uint32_t n_bytes, uint32_t n_groups, hipStream_t stream) {
    hipLaunchKernelGGL(bv_gid_prepare_kernel, dim3((n_bytes + 255u) / 256u), dim3(256), 0, stream, gid, gidp, n_bytes, n_groups);
}

// ------------------------------------------------------------------------------ pop-group calls, four per wave
// __gb(): BaseType(subset) + lrt([REF] + alts), caller.cpp:756-759, 767-797, for the (variant site, group) items the tally
// kernels handed over (BvPass2Args::gitems): one item per group of 16 lanes, bins in registers (bv_solver16.h).
#define BV_P2G_NW 4
#define BV_P2G_OCC 3  /* 160 VGPRs, no spills; at 4 waves per SIMD (128 VGPRs) 28 registers spilled: equal at 1-2 groups, 19 % slower at 8 */
struct __attribute__((aligned(16))) BvP2gShared {
    double tab_hit[BV_QBINS], tab_miss[BV_QBINS];
    double tab_loghit[BV_QBINS], tab_logmiss[BV_QBINS];  // (from device memory they are a trip per slot inside a dependent chain)
    uint32_t grp[BV_P2G_NW][4][BV_G16_GRP_WORDS];  // per group of 16 lanes: previous marginals, [slot][lane of the group]
};
__global__ __launch_bounds__(BV_WAVE *BV_P2G_NW, BV_P2G_OCC) void bv_p2g_solve16_kernel(BvPass2Args a) {
    __shared__ BvP2gShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < BV_QBINS; i += BV_WAVE * BV_P2G_NW) {
        sh.tab_hit[i] = a.tables->hit[i];
        sh.tab_miss[i] = a.tables->miss[i];
        sh.tab_loghit[i] = a.tables->loghit[i];
        sh.tab_logmiss[i] = a.tables->logmiss[i];
    }
    __syncthreads();
    const uint32_t n_var = a.counters[BV_CTR_VARIANTS];
    const uint64_t all = (uint64_t)n_var * a.n_groups;
    const uint32_t n_items = all < (uint64_t)a.gitem_cap ? (uint32_t)all : a.gitem_cap;
    const uint32_t n_waves = gridDim.x * BV_P2G_NW, gw = blockIdx.x * BV_P2G_NW + (uint32_t)wave;
    const int grp = lane >> 4, gl = lane & 15;
    // (Tried, round 6: every group of 16 lanes walking a stream of items of its own and skipping what is not pending sixteen
    // headers at a time, so that no wave carries an idle quarter -- the probes' load latency in front of every job cost more than
    // the idle quarters: 60 -> 52 M sites/s at 32 groups.)
    for (uint32_t t = gw; (uint64_t)t * 4u < n_items; t += n_waves) {
        const uint32_t idx = t * 4u + (uint32_t)grp;
        if (idx >= n_items) continue;
        const uint32_t *it = a.gitems + (size_t)idx * BV_P2G_ITEM_WORDS;
        const uint32_t hdr = it[0];
        if (!(hdr & BV_P2G_PENDING)) continue;
        const uint32_t nb = hdr & 0xFFFFu;
        if (hdr & (BV_P2G_L4 | BV_P2G_L8)) continue;  // bv_p2g_solve_small_kernel's: sixteen or eight such items per wave
        const uint32_t v = idx / a.n_groups, g = idx - v * a.n_groups;
        const uint32_t site = a.var_list[v];
        // a chained launch: the segment's records and reference bases, looked up per group of 16 lanes (its site is its own)
        const bv_site_result *outp = a.out;
        const uint8_t *refp = a.ref_base;
        bv_group_result *goutp = a.gout;
        if (a.ch != nullptr) {
            const BvChainC ch = bv_chain_const(a.ch);
            const uint32_t sg = bv_chain_seg(ch, site);
            if (!a.ch_cat) { outp = ch->out[sg]; refp = ch->ref_base[sg]; }
            goutp = ch->gout[sg];
        }
        const bv_site_result *res = &outp[site];
        int ref = refp[site];
        if (ref > 4) ref = 4;
        const int n_alt = res->n_alt;
        int comb = ref, nc = 1;  // caller.cpp:750-753: [toupper(REF)] + alts, 3 bits per entry
        // (Tried, round 6: site and [REF] + alts riding in the item, so that a job starts from one load instead of the chain
        // variant list -> record -> reference base: no change, 60.4 against 60.2 M sites/s at 32 groups -- the kernel is bound by
        // its FP64 instructions, not by these loads.)
#pragma unroll
        for (int k = 0; k < BV_MAX_ALT; ++k) {
            if (k < n_alt) {
                comb |= (res->alt[k] & 3) << (3 * nc);
                ++nc;
            }
        }
        uint32_t gdepth[4] = {it[1], it[2], it[3], it[4]};
        const uint32_t gtotal = gdepth[0] + gdepth[1] + gdepth[2] + gdepth[3];
        BvG16Bins B;
        B.hit = sh.tab_hit; B.miss = sh.tab_miss; B.loghit = sh.tab_loghit; B.logmiss = sh.tab_logmiss;
        B.pm = reinterpret_cast<double *>(sh.grp[wave][grp]) + gl;
#pragma unroll
        for (int s = 0; s < BV_G16_SLOTS; ++s) {
            const uint32_t i = (uint32_t)(s * 16 + gl);
            B.w[s] = i < nb ? it[8u + i] : 0u;
        }
        BvLrtOut L;
        // (a job whose four groups all have at most 32 bins -- every job of a run with many groups -- runs the two-slot
        // instance: a third of the eight-slot one's instructions are the tests of empty slots)
        if (__ballot(nb > 48u) == 0ull) bv_lrt_g16<true, 3, true>(B, gdepth, gtotal, ref, a.min_af, L, comb, nc);
        else bv_lrt_g16<true, BV_G16_SLOTS, true>(B, gdepth, gtotal, ref, a.min_af, L, comb, nc);
        if ((hdr & BV_P2G_SHALLOW) && L.tie_risk) {
            // a tie (or what rounding makes of one) in a group of at most BV_ORD_MAX covered samples: the reference's per-sample
            // order decides it -- the item goes on to bv_p2g_hard_kernel, which runs behind this kernel, in that kernel's format
            uint32_t *itw = a.gitems + (size_t)idx * BV_P2G_ITEM_WORDS;
#pragma unroll
            for (int s = 0; s < BV_G16_SLOTS; ++s) {
                const uint32_t i = (uint32_t)(s * 16 + gl);
                if (i < nb) itw[8u + i] = ((B.w[s] >> 16) << 23) | (B.w[s] & 0xFFFFu);
            }
            if (gl == 0) itw[0] = nb | BV_P2G_HARD | BV_P2G_SHALLOW;
            continue;
        }
        if (gl == 0) {
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
            goutp[(size_t)site * a.n_groups + g] = gr;
        }
    }
}

// ------------------------------------------------------------------------------ small pop-groups, eight or sixteen per wave
// With 32 or 64 pop-groups a group sees a dozen or two covered cells per site: the LRT of such an item is a few hundred FP64
// instructions of which most do not depend on the number of bins -- divisions, two logarithms' worth of polynomial, the LRT's
// control flow, DPP reductions -- and on 16 lanes per item a wave spends them four items at a time (bv_p2g_solve16_kernel:
// 2,370 VALU instructions per job of four, the kernel FP64-issue-bound at 0.72 ms for the 640 k items of 100 k sites x 32 groups).
// Here an item owns L = 4 or 8 lanes -- an aligned part of a DPP row, four register slots per lane: up to 16 / 32 bins -- so the
// same instructions serve 16 or 8 items (eight slots when an item of the job has more); the row reductions are two or three DPP
// steps instead of four.  Which kernel takes an item is decided per SITE by the tally kernel (BV_P2G_L4 / BV_P2G_L8 in the item's
// header, bv_kernels.h): a wave's consecutive items belong to one site and are of one kind, and a site's records do not depend
// on what else is in the launch; every kernel walks all items and skips the other kinds (a skipped item costs its header's load).
// Same arithmetic per term as bv_lrt_g16 on 16 lanes; only the order of the sums over bins differs (~1e-16 relative).
template <int L>
__global__ __launch_bounds__(BV_WAVE *BV_P2G_NW, BV_P2G_OCC) void bv_p2g_solve_small_kernel(BvPass2Args a) {
    constexpr uint32_t IPW = BV_WAVE / L, KIND = (L == 4) ? BV_P2G_L4 : BV_P2G_L8;  // items per wave; the header bit of this kernel's items
    __shared__ BvP2gShared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < BV_QBINS; i += BV_WAVE * BV_P2G_NW) {
        sh.tab_hit[i] = a.tables->hit[i];
        sh.tab_miss[i] = a.tables->miss[i];
        sh.tab_loghit[i] = a.tables->loghit[i];
        sh.tab_logmiss[i] = a.tables->logmiss[i];
    }
    __syncthreads();
    const uint32_t n_var = a.counters[BV_CTR_VARIANTS];
    const uint64_t all = (uint64_t)n_var * a.n_groups;
    const uint32_t n_items = all < (uint64_t)a.gitem_cap ? (uint32_t)all : a.gitem_cap;
    const uint32_t n_waves = gridDim.x * BV_P2G_NW, gw = blockIdx.x * BV_P2G_NW + (uint32_t)wave;
    const uint32_t sub = (uint32_t)lane / L, gl = (uint32_t)lane % L;
    const int row = lane >> 4, lr = lane & 15;
    for (uint32_t t = gw; (uint64_t)t * IPW < n_items; t += n_waves) {
        const uint32_t idx = t * IPW + sub;
        if (idx >= n_items) continue;
        const uint32_t *it = a.gitems + (size_t)idx * BV_P2G_ITEM_WORDS;
        const uint32_t hdr = it[0];
        if (!(hdr & BV_P2G_PENDING) || !(hdr & KIND)) continue;
        const uint32_t nb = hdr & 0xFFFFu;  // <= 8 * L (the tally kernel's rule)
        const uint32_t v = idx / a.n_groups, g = idx - v * a.n_groups;
        const uint32_t site = a.var_list[v];
        const bv_site_result *outp = a.out;
        const uint8_t *refp = a.ref_base;
        bv_group_result *goutp = a.gout;
        if (a.ch != nullptr) {  // a chained launch: the segment's records and reference bases (the item's site is its own)
            const BvChainC ch = bv_chain_const(a.ch);
            const uint32_t sg = bv_chain_seg(ch, site);
            if (!a.ch_cat) { outp = ch->out[sg]; refp = ch->ref_base[sg]; }
            goutp = ch->gout[sg];
        }
        const bv_site_result *res = &outp[site];
        int ref = refp[site];
        if (ref > 4) ref = 4;
        const int n_alt = res->n_alt;
        int comb = ref, nc = 1;  // caller.cpp:750-753: [toupper(REF)] + alts, 3 bits per entry
#pragma unroll
        for (int k = 0; k < BV_MAX_ALT; ++k) {
            if (k < n_alt) {
                comb |= (res->alt[k] & 3) << (3 * nc);
                ++nc;
            }
        }
        uint32_t gdepth[4] = {it[1], it[2], it[3], it[4]};
        const uint32_t gtotal = gdepth[0] + gdepth[1] + gdepth[2] + gdepth[3];
        BvG16Bins B;
        B.hit = sh.tab_hit; B.miss = sh.tab_miss; B.loghit = sh.tab_loghit; B.logmiss = sh.tab_logmiss;
        B.pm = reinterpret_cast<double *>(sh.grp[wave][row]) + lr;  // (the row's scratch: every lane of the row has a column of its own)
#pragma unroll
        for (int s = 0; s < BV_G16_SLOTS; ++s) {
            const uint32_t i = (uint32_t)s * L + gl;
            B.w[s] = i < nb ? it[8u + i] : 0u;
        }
        BvLrtOut Lo;
        // (four slots when no item of the job needs more: the loops over the slots stop there)
        if (__ballot(nb > 4u * L) == 0ull) bv_lrt_g16<true, 4, true, L>(B, gdepth, gtotal, ref, a.min_af, Lo, comb, nc);
        else bv_lrt_g16<true, BV_G16_SLOTS, true, L>(B, gdepth, gtotal, ref, a.min_af, Lo, comb, nc);
        if ((hdr & BV_P2G_SHALLOW) && Lo.tie_risk) {
            // a tie in a group of at most BV_ORD_MAX covered samples: on to bv_p2g_hard_kernel, in that kernel's format (as bv_p2g_solve16_kernel)
            uint32_t *itw = a.gitems + (size_t)idx * BV_P2G_ITEM_WORDS;
#pragma unroll
            for (int s = 0; s < BV_G16_SLOTS; ++s) {
                const uint32_t i = (uint32_t)s * L + gl;
                if (i < nb) itw[8u + i] = ((B.w[s] >> 16) << 23) | (B.w[s] & 0xFFFFu);
            }
            if (gl == 0) itw[0] = nb | BV_P2G_HARD | BV_P2G_SHALLOW;
            continue;
        }
        if (gl == 0) {
            bv_group_result gr;
            gr.n_alt = (uint8_t)Lo.n_alt;
            gr.reserved[0] = gr.reserved[1] = gr.reserved[2] = 0;
            gr.total_depth = gtotal;
            gr.reserved2 = 0;
#pragma unroll
            for (int k = 0; k < BV_MAX_ALT; ++k) {
                gr.alt[k] = (k < Lo.n_alt) ? (uint8_t)bv_alt_at(Lo, k) : 0;
                gr.af[k] = (k < Lo.n_alt) ? Lo.af[k] : 0.0;
            }
            goutp[(size_t)site * a.n_groups + g] = gr;
        }
    }
}

void bv_launch_p2g_solve16(const BvPass2Args &a, hipStream_t stream) {
    const bool groups = a.n_groups > 0 && a.group_id != nullptr && a.gout != nullptr;
    if (!groups || a.gitems == nullptr || a.gitem_cap == 0u) return;
    uint32_t grid = (a.n_cu ? a.n_cu : 256u) * (uint32_t)BV_P2G_OCC;
    const uint64_t items = (uint64_t)a.n_sites * a.n_groups;
    const uint64_t need = (items + 4u * BV_P2G_NW - 1u) / (4u * BV_P2G_NW);
    if ((uint64_t)grid > need) grid = need > 0 ? (uint32_t)need : 1u;
    const uint32_t cap = (a.flags >> 16) & 0xFFu;  // BV_FLAG_GRID_LIMIT
    if (cap && grid > cap) grid = cap;
    // the small classes first (sixteen / eight items per wave), then everything larger four per wave
    {
        auto grid_for = [&](uint32_t ipw) {
            uint32_t gsm = (a.n_cu ? a.n_cu : 256u) * (uint32_t)BV_P2G_OCC;
            const uint64_t nd = (items + (uint64_t)ipw * BV_P2G_NW - 1u) / ((uint64_t)ipw * BV_P2G_NW);
            if ((uint64_t)gsm > nd) gsm = nd > 0 ? (uint32_t)nd : 1u;
            if (cap && gsm > cap) gsm = cap;
            return gsm;
        };
        hipLaunchKernelGGL(bv_p2g_solve_small_kernel<4>, dim3(grid_for(16u)), dim3(BV_WAVE * BV_P2G_NW), 0, stream, a);
        hipLaunchKernelGGL(bv_p2g_solve_small_kernel<8>, dim3(grid_for(8u)), dim3(BV_WAVE * BV_P2G_NW), 0, stream, a);
    }
    hipLaunchKernelGGL(bv_p2g_solve16_kernel, dim3(grid), dim3(BV_WAVE * BV_P2G_NW), 0, stream, a);
    if (bv_p2g_all_items(a)) {  // (when some items do not fit the scratch, the workgroup-per-row kernel solves the rest of them itself)
        uint32_t gridh = (a.n_cu ? a.n_cu : 256u) * 4u;  // 127 VGPRs: four waves per SIMD
        const uint64_t needh = (items + BV_P2GH_WAVES - 1u) / BV_P2GH_WAVES;
        if ((uint64_t)gridh > needh) gridh = needh > 0 ? (uint32_t)needh : 1u;
        if (cap && gridh > cap) gridh = cap;
        hipLaunchKernelGGL(bv_p2g_hard_kernel, dim3(gridh), dim3(BV_WAVE * BV_P2GH_WAVES), 0, stream, a);
    }
}

// every (variant site, group) has an item: no tally kernel needs the solver
bool bv_p2g_all_items(const BvPass2Args &a) {
    const bool groups = a.n_groups > 0 && a.group_id != nullptr && a.gout != nullptr;
    return groups && a.gitems != nullptr && (uint64_t)a.n_sites * a.n_groups <= (uint64_t)a.gitem_cap;
}

bool bv_p2g_streams(const BvPass2Args &a) {
    const bool groups = a.n_groups > 0 && a.group_id != nullptr && a.gout != nullptr;
    return groups && a.n_groups <= BV_P2GS_MAX_GROUPS && a.n_samples > 2048u && a.n_samples <= BV_SHORT_ROW_MAX && a.gitems != nullptr &&
           a.gidp != nullptr && (uint64_t)a.n_sites * a.n_groups <= (uint64_t)a.gitem_cap && !(a.flags & BV_FLAG_PASS2_SWEEP);
}

template <int K>
static void bv_launch_p2g_stream_k(const BvPass2Args &a, hipStream_t stream) {
    const size_t dyn = (size_t)BV_P2GS_WAVES * a.n_groups * 512u * sizeof(uint32_t);
    const size_t per_wg = sizeof(BvP2gsShared<K>) + dyn;
    uint32_t wg_per_cu = (uint32_t)((160u * 1024u) / per_wg);
    if (wg_per_cu > 4u) wg_per_cu = 4u;
    if (wg_per_cu < 1u) wg_per_cu = 1u;
    uint32_t grid = (a.n_cu ? a.n_cu : 256u) * wg_per_cu;
    const uint32_t need = (a.n_sites + BV_P2GS_WAVES - 1) / BV_P2GS_WAVES;
    if (grid > need) grid = need;
    const uint32_t cap = (a.flags >> 16) & 0xFFu;  // BV_FLAG_GRID_LIMIT
    if (cap && grid > cap) grid = cap;
    hipLaunchKernelGGL((bv_p2g_stream_kernel<K>), dim3(grid), dim3(BV_WAVE * BV_P2GS_WAVES), dyn, stream, a);
}
void bv_launch_p2g_stream(const BvPass2Args &a, hipStream_t stream) {
    // three slots per wave while two workgroups (8 waves) still fit a CU's 160 KiB next to the histograms (2 KiB per wave
    // and group), else two: 6 and 7 groups
    if (sizeof(BvP2gsShared<3>) + (size_t)BV_P2GS_WAVES * a.n_groups * 2048u <= 80u * 1024u) bv_launch_p2g_stream_k<3>(a, stream);
    else bv_launch_p2g_stream_k<2>(a, stream);
}

