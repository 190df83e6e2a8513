// bv_pass1.hip -- pass 1 of the per-site basetype path: tally + solve, every site.
//
// One workgroup owns one site (row of the slab).  Its waves stream the row's two byte
// planes with 16-byte coalesced loads along the sample axis (2 B / cell, read exactly
// once), tally covered cells into a 4 KiB LDS histogram over (strand, base, phred) with
// LDS atomics, and then run the whole reference solver on that histogram:
//   BaseType ctor counts        src/basetype.cpp:45-71      -> depth[], total_depth
//   strand_bias (CVG flavour)   src/basetype.cpp:244-295    via caller.cpp:1236-1245
//   lrt(): EM / LRT / AF / QUAL src/basetype.cpp:130-199, src/algorithm.h:148-255
//   strand_bias (VCF flavour)   caller.cpp:1164
//   base-quality rank sum       src/basetype.cpp:201-242 via caller.cpp:1157
//   QD, CM_CAF                  caller.cpp:1122, 1160-1161
// The kernel is HBM-bound by design (no inter-site reuse, so no XCD-aware remap is needed:
// nothing is shared between workgroups).  No MFMA: categorical tallies + small FP64 tables.
#include "bv_kernels.h"

struct __attribute__((aligned(16))) BvSiteShared {
    uint32_t hist[BV_HIST_WORDS];        // [(rev<<2)|base][phred]
    uint32_t bin_code[BV_SLOTS * BV_WAVE];  // compacted non-empty (base<<7 | phred) bins
    uint32_t bin_cnt[BV_SLOTS * BV_WAVE];
    uint32_t fwd[4], rev[4];
    uint32_t nb, badq;
    double tab_hit[BV_QBINS], tab_miss[BV_QBINS];  // LDS copy of BvTables
    BvLrtShared lrt;
    bv_site_result res;                  // staged record, stored with one coalesced write
};

// ---- tally of one 16-cell chunk (one lane's 16 B of each plane)
__device__ __forceinline__ void bv_tally_dword(uint32_t w, uint32_t qq, uint32_t *hist) {
    qq &= 0x7F7F7F7Fu;  // keep the histogram index inside its 128-wide row whatever the input
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t c = (w >> (8 * j)) & 0xFFu;
        if (!(c & BV_CELL_NOCALL)) {
            uint32_t idx = ((c & 7u) << 7) | ((qq >> (8 * j)) & 0xFFu);
            atomicAdd(&hist[idx], 1u);  // ds_add_u32, no return
        }
    }
}

// cells at or beyond n_samples in the row's last chunk are forced to 'N'
__device__ __forceinline__ uint32_t bv_mask_tail_dword(uint32_t w, int keep) {
    if (keep >= 4) return w;
    if (keep <= 0) return 0x08080808u;
    uint32_t low = (1u << (8 * keep)) - 1u;
    return (w & low) | (0x08080808u & ~low);
}

template <int NT>
__device__ __forceinline__ void bv_tally_row(const uint8_t *bs_row, const uint8_t *q_row, uint32_t n_samples,
                                             uint32_t *hist, int tid) {
    const bv_u32x4 *b4 = reinterpret_cast<const bv_u32x4 *>(bs_row);
    const bv_u32x4 *q4 = reinterpret_cast<const bv_u32x4 *>(q_row);
    const uint32_t n_chunks = (n_samples + 15u) >> 4;
    const int tail = (int)(n_samples & 15u);
#ifndef BV_TALLY_U
#define BV_TALLY_U 4
#endif
    constexpr int U = BV_TALLY_U;  // 2*U x 16-byte loads in flight per lane
    for (uint32_t base = 0; base < n_chunks; base += NT * U) {
        bv_u32x4 vb[U], vq[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t idx = base + u * NT + tid;
            if (idx < n_chunks) {
                vb[u] = __builtin_nontemporal_load(b4 + idx);
                vq[u] = __builtin_nontemporal_load(q4 + idx);
            } else {
                vb[u] = bv_u32x4{0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u};
                vq[u] = bv_u32x4{0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t idx = base + u * NT + tid;
            if (tail && idx == n_chunks - 1) {
                vb[u].x = bv_mask_tail_dword(vb[u].x, tail);
                vb[u].y = bv_mask_tail_dword(vb[u].y, tail - 4);
                vb[u].z = bv_mask_tail_dword(vb[u].z, tail - 8);
                vb[u].w = bv_mask_tail_dword(vb[u].w, tail - 12);
            }
            bv_tally_dword(vb[u].x, vq[u].x, hist);
            bv_tally_dword(vb[u].y, vq[u].y, hist);
            bv_tally_dword(vb[u].z, vq[u].z, hist);
            bv_tally_dword(vb[u].w, vq[u].w, hist);
        }
    }
}

// ---- wave 0: strand/base row sums and deterministic compaction of non-empty bins
__device__ __forceinline__ void bv_prologue_wave(BvSiteShared *sh, int lane) {
    uint32_t nb = 0, badq = 0;
    uint32_t fwd[4], rev[4];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int b = r >> 1;
        const int q = ((r & 1) << 6) | lane;
        uint32_t f = sh->hist[(b << 7) | q], v = sh->hist[((b | 4) << 7) | q];
        uint32_t fs = bv_wave_sum_u32(f), rs = bv_wave_sum_u32(v);
        if (r & 1) { fwd[b] += fs; rev[b] += rs; } else { fwd[b] = fs; rev[b] = rs; }
        uint32_t c = f + v;
        bool valid = (c != 0) && (q < BV_NQ_VALID);
        bool bad = (c != 0) && (q >= BV_NQ_VALID);
        unsigned long long m = __ballot(valid);
        uint32_t pos = nb + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (valid) {
            sh->bin_code[pos] = ((uint32_t)b << 7) | (uint32_t)q;
            sh->bin_cnt[pos] = c;
        }
        nb += (uint32_t)__popcll(m);
        badq |= (__ballot(bad) != 0ull) ? 1u : 0u;
    }
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < 4; ++b) { sh->fwd[b] = fwd[b]; sh->rev[b] = rev[b]; }
        sh->nb = nb;
        sh->badq = badq;
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void bv_pass1_kernel(BvPass1Args a) {
    __shared__ BvSiteShared sh;
    constexpr int NW = NT / BV_WAVE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int i = tid; i < BV_QBINS; i += NT) {
        sh.tab_hit[i] = a.tables->hit[i];
        sh.tab_miss[i] = a.tables->miss[i];
    }
    // one site per workgroup (no grid-stride loop: a loop around the whole body lets LICM hoist
    // every libm polynomial constant of the solver into registers that then stay live across
    // the streaming phase and cost it its occupancy)
    const uint32_t site = blockIdx.x;
    {
        // ---- clear the histogram and the staged record
        {
            uint4 *h4 = reinterpret_cast<uint4 *>(sh.hist);
            for (int i = tid; i < BV_HIST_WORDS / 4; i += NT) h4[i] = make_uint4(0, 0, 0, 0);
            if (tid < (int)(sizeof(bv_site_result) / 4)) reinterpret_cast<uint32_t *>(&sh.res)[tid] = 0u;
        }
        __syncthreads();

        // ---- tally: the only HBM traffic of this pass, 2 B per cell
        bv_tally_row<NT>(a.bs + (size_t)site * a.pitch, a.q + (size_t)site * a.pitch, a.n_samples, sh.hist, tid);
        __syncthreads();

        if (wave == 0) bv_prologue_wave(&sh, lane);
        __syncthreads();

        uint32_t depth[4], fwd[4], rev[4], total = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            fwd[b] = sh.fwd[b];
            rev[b] = sh.rev[b];
            depth[b] = fwd[b] + rev[b];
            total += depth[b];
        }
        const int nb = (int)sh.nb;
        const uint32_t badq = sh.badq;
        int ref = a.ref_base[site];
        if (ref > 4) ref = 4;

        if (total == 0) {  // nothing to call: caller.cpp:718 / basetype.cpp:132; record stays zero
            if (tid == 0) {
                sh.res.mq_ranksum = sh.res.rpr_ranksum = sh.res.bq_ranksum = __builtin_nan("");
            }
            __syncthreads();
            if (tid < (int)(sizeof(bv_site_result) / 4))
                reinterpret_cast<uint32_t *>(&a.out[site])[tid] = reinterpret_cast<uint32_t *>(&sh.res)[tid];
            return;
        }

        BvBins B;
        B.code = sh.bin_code; B.cnt = sh.bin_cnt; B.hit = sh.tab_hit; B.miss = sh.tab_miss; B.nb = nb;

        uint32_t flags = BV_SITE_COVERED | (badq ? BV_SITE_BAD_QUAL : 0u);

        // ---- CVG strand bias: alt = every non-ref ACGT base (caller.cpp:1236-1245).
        // Runs on wave 1 while wave 0 does the top-level EM.
        if (wave == (NW > 1 ? 1 : 0)) {
            uint32_t rf = 0, rr = 0, af = 0, ar = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b == ref) { rf += fwd[b]; rr += rev[b]; } else { af += fwd[b]; ar += rev[b]; }
            }
            double fs, sor;
            uint32_t fl = 0;
            bv_strand_bias_wave(rf, rr, af, ar, lane, &fs, &sor, &fl);
            if (lane == 0) {
                sh.res.cvg_sb[0] = rf; sh.res.cvg_sb[1] = rr; sh.res.cvg_sb[2] = af; sh.res.cvg_sb[3] = ar;
                sh.res.cvg_fs = fs;
                sh.res.cvg_sor = sor;
                if (fl) atomicOr(&sh.res.status, fl);
            }
        }

        // ---- lrt() over ACGT (basetype.h:115)
        BvLrtOut L;
        bv_lrt<NW>(B, depth, total, /*A,C,G,T*/ 0 | (1 << 3) | (2 << 6) | (3 << 9), 4, ref, a.min_af, &sh.lrt, wave, lane, L);
        if (L.zero_freq) flags |= BV_SITE_ZERO_FREQ;

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
            if (wave == 0 && lane == 0) {
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
                        sh.res.alt[k] = (uint8_t)bv_alt_at(L, k);
                        sh.res.af[k] = L.af[k];
                        sh.res.caf[k] = (double)d / (int)total;
                    }
                }
                double qd = qual / ad_sum;
                if (qd == 0) qd = 0.0;
                sh.res.n_alt = (uint8_t)L.n_alt;
                sh.res.qual = qual;
                sh.res.qd = qd;
            }
            // VCF strand bias w.r.t. the chosen ALTs (caller.cpp:1164)
            if (wave == (1 % NW)) {
                uint32_t rf = 0, rr = 0, af = 0, ar = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b == ref) { rf += fwd[b]; rr += rev[b]; }
                    else if ((alt_mask >> b) & 1u) { af += fwd[b]; ar += rev[b]; }
                }
                double fs, sor;
                uint32_t fl = 0;
                bv_strand_bias_wave(rf, rr, af, ar, lane, &fs, &sor, &fl);
                if (lane == 0) {
                    sh.res.var_sb[0] = rf; sh.res.var_sb[1] = rr; sh.res.var_sb[2] = af; sh.res.var_sb[3] = ar;
                    sh.res.var_fs = fs;
                    sh.res.var_sor = sor;
                    if (fl) atomicOr(&sh.res.status, fl);
                }
            }
            // base-quality rank sum from the histogram this pass already holds (caller.cpp:1157)
            if (wave == (2 % NW)) {
                unsigned long long n1 = (ref < 4) ? bv_sel4u(depth, ref) : 0ull, n2 = ad_sum_u;
                unsigned long long below = 0, twoR = 0;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    int q = w * 64 + lane;
                    uint32_t rv = 0, av = 0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        uint32_t c = sh.hist[(b << 7) | q] + sh.hist[((b | 4) << 7) | q];
                        if (b == ref) rv += c;
                        else if ((alt_mask >> b) & 1u) av += c;
                    }
                    twoR += bv_ranksum_window(rv, av, n1 + n2, below, lane);
                }
                double ph = bv_ranksum_phred(twoR, n1, n2);
                if (lane == 0) sh.res.bq_ranksum = ph;
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) sh.res.depth[b] = depth[b];
            sh.res.total_depth = total;
            atomicOr(&sh.res.status, flags);
            sh.res.chi2 = L.chi2;
            sh.res.em_iters = (uint16_t)L.em_iters;
            sh.res.n_em = (uint8_t)L.n_em;
            sh.res.mq_ranksum = __builtin_nan("");
            sh.res.rpr_ranksum = __builtin_nan("");
            if (L.n_alt == 0) sh.res.bq_ranksum = __builtin_nan("");
            if (L.n_alt > 0) {
                uint32_t slot = atomicAdd(&a.counters[0], 1u);
                a.var_list[slot] = site;
            }
            if (L.zero_freq) atomicAdd(&a.counters[1], 1u);
        }
        __syncthreads();
        if (tid < (int)(sizeof(bv_site_result) / 4))
            reinterpret_cast<uint32_t *>(&a.out[site])[tid] = reinterpret_cast<uint32_t *>(&sh.res)[tid];
    }
}

void bv_launch_pass1(const BvPass1Args &a, hipStream_t stream) {
    // team size by row length: one wave per site for short rows, 4 or 16 waves for long ones
    const uint32_t n = a.n_samples;
    uint32_t grid = a.n_sites;
    if (n <= 16384u) {
        hipLaunchKernelGGL(bv_pass1_kernel<64>, dim3(grid), dim3(64), 0, stream, a);
    } else if (n <= 400000u) {
        hipLaunchKernelGGL(bv_pass1_kernel<256>, dim3(grid), dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(bv_pass1_kernel<1024>, dim3(grid), dim3(1024), 0, stream, a);
    }
}
