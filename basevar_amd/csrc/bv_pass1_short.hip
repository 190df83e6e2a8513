// bv_pass1_short.hip -- pass 1 for short rows (<= 49,152 samples per site): a streaming kernel and a solve kernel.
//
// On short rows the solve, not the stream, used to set the pace of pass 1: one wave tallied a row and then solved it,
// and while it solved nothing of its own was in flight (40-51 % of the HBM peak at 10 k samples).  Here the two
// halves are separate kernels that meet in a small HBM scratch:
//
//   bv_p1s_stream_kernel   every wave owns a contiguous range of sites and streams their rows as ONE sequence of
//                          1 KiB + 1 KiB slots (64 lanes x 16 bytes of the call plane and of the phred plane) through
//                          a private ring in LDS filled by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no wait until
//                          the slot is consumed, K slots in flight per wave ACROSS row boundaries -- the next rows are
//                          already on their way while a row's totals are formed).  Per site it leaves a 48-byte
//                          summary (strand x base totals, flags); only sites that need the EM -- more than one active
//                          base, a non-reference base, a phred-0 call, a deep strand table -- also export their
//                          compacted (base, phred) bins and go on the candidate list.
//   bv_p1s_solve16_kernel  (a) the candidates only the wave solver takes (shallow sites, phred-0 calls, > 128 bins) get a whole
//                          wave and the full solver of bv_solver.h on their bins; (b) the ordinary candidates are solved four
//                          per wave, one per group of 16 lanes (bv_solver16.h); (c) every non-candidate site (hom-ref, or
//                          uncovered) is finished ONE LANE PER SITE: the work there is scalar per site (depths, one small
//                          Fisher test), and a whole wave per site repeated it 64 times over.
//
// Reference functions realised: those of bv_pass1.hip (src/basetype.cpp:22-295, src/algorithm.h:44-255,
// htslib/kfunc.c:39-143,197-313); results are bit-identical to the one-kernel form for candidates (same bins, same
// order, same solver) and equal to ~1e-13 relative for the per-lane Fisher test, which walks the tables exactly as
// kt_fisher_exact does (kfunc.c:291-307, incremental hypergeo_acc with its re-seeding every 11 tables).
//
// HBM-bound by design (2 B per cell, each byte read once); no MFMA (categorical tallies).
#define BV_LNFACT_TABLE_ONLY 1  /* rows of at most 65,535 samples: see bv_lnfact */
#include "bv_kernels.h"

#include "bv_solver.h"
#include "bv_solver16.h"
#include "bv_tally.h"

#include "bv_short.h"


// ------------------------------------------------------------------------------ streaming kernel
#define BV_S_REFCAP 4096                     /* sites of a workgroup's range handled per pass (their reference bases sit in LDS) */
template <int NW, int K, int U>
struct __attribute__((aligned(16))) BvP1sStreamShared {
    uint32_t hist[NW][BV_S_HWORDS + BV_S_OVF + 8];     // per wave: [(rev<<2)|base][phred < 128], then the overflow rows
    uint32_t ring[NW][K][U * BV_S_SLOT_WORDS];         // per wave: K slots of U x (1 KiB calls + 1 KiB phreds)
    uint8_t refl[BV_S_REFCAP];                         // reference bases of the workgroup's current sites
    uint32_t cursor;                                   // sites of the current pass handed out so far
};

// One slot of one row: tally its 64 x 16 cells into the wave's histogram.
template <bool LAST>
__device__ __forceinline__ void bv_p1s_tally_slot(bv_u32x4 vb, bv_u32x4 vq, uint32_t chunk, uint32_t n_chunks, int tail,
                                                  uint32_t *hist, uint32_t one, int lane = 0) {
    if (LAST) {
        // the row's last slot: lanes past the row were not loaded (their LDS bytes are stale), and the last chunk may be partial
        if (chunk >= n_chunks) {
            vb = bv_u32x4{0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u};
            vq = bv_u32x4{0u, 0u, 0u, 0u};
        } else if (tail && chunk == n_chunks - 1) {
            vb.x = bv_mask_tail_dword(vb.x, tail);
            vb.y = bv_mask_tail_dword(vb.y, tail - 4);
            vb.z = bv_mask_tail_dword(vb.z, tail - 8);
            vb.w = bv_mask_tail_dword(vb.w, tail - 12);
        }
    }
    // A phred byte >= 128 (invalid input) would carry into its neighbour under the shift below: such a slot takes the
    // exact cell-by-cell path (wave-uniform branch; never taken on valid data).
    const uint32_t hi = (vq.x | vq.y | vq.z | vq.w) & 0x80808080u;
    if (__builtin_expect(__ballot(hi != 0u) != 0ull, 0)) {
        const uint32_t wb[4] = {vb.x, vb.y, vb.z, vb.w}, wq[4] = {vq.x, vq.y, vq.z, vq.w};
#pragma unroll 1
        for (int j = 0; j < 16; ++j) {
            const uint32_t c = (wb[j >> 2] >> (8 * (j & 3))) & 0xFFu, p = (wq[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            if (c < 8u) atomicAdd(p < 128u ? &hist[(c << 7) | p] : &hist[BV_S_HWORDS + c], 1u);
        }
        return;
    }
    // X = call << 8 | phred << 1 = twice the word index of the 8 x 128 histogram; call < 8 <=> X < 0x800
    vq.x <<= 1; vq.y <<= 1; vq.z <<= 1; vq.w <<= 1;
    bv_tally_chunk<1>(vb, vq, hist, one);
}

#define BV_P1S_NW 4
template <int NW, int K, int U, bool CHAIN = false>
__global__ __launch_bounds__(BV_WAVE *NW) void bv_p1s_stream_kernel(BvP1ShortArgs a) {
    __shared__ BvP1sStreamShared<NW, K, U> sh;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *hist = sh.hist[wave];
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(bv_lds_u32 *)sh.ring[wave][0]);
    const uint32_t *ring = sh.ring[wave][0];
    const uint32_t cursor_lds = (uint32_t)(uintptr_t)(bv_lds_u32 *)&sh.cursor;
    {
        uint4 *h4 = reinterpret_cast<uint4 *>(hist);
#pragma unroll
        for (int i = 0; i < (BV_S_HWORDS + BV_S_OVF + 8) / 4 / BV_WAVE + 1; ++i)
            if (i * BV_WAVE + lane < (BV_S_HWORDS + BV_S_OVF + 8) / 4) h4[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
    }
    // The WORKGROUP owns a contiguous range of sites; its waves draw chunks of C consecutive sites from a cursor in LDS.
    // (Each wave used to own a fixed range of its own.  Measured with per-wave time stamps: of the two waves that share a
    // SIMD the one in wave slot 0 -- the older one, which the issue arbiter prefers -- was done after 303 us, the one in
    // slot 1 after 360 us, and the kernel lasted as long as the slower half.  Drawing the work in small chunks lets the
    // favoured waves take more of it; an LDS atomic per chunk does not touch vmcnt, so the ring keeps its counted waits.)
    const uint32_t B0 = (uint32_t)((uint64_t)a.n_sites * blockIdx.x / gridDim.x), B1 = (uint32_t)((uint64_t)a.n_sites * (blockIdx.x + 1) / gridDim.x);
#ifdef BV_TEAM_DEBUG  /* per-wave end stamps, per-workgroup start stamp and XCD (tools/experiments/r3_team_debug.sh) */
    const uint32_t gw = blockIdx.x * NW + (uint32_t)wave;
    uint32_t *dbg_ = a.counters + BV_CTR_WORDS;
    if (gridDim.x * NW <= 2048u) {
        if (threadIdx.x == 0 && blockIdx.x == 0) dbg_[5150] = 1u;  // whose stamps these are
        if (threadIdx.x == 0) { dbg_[4096 + blockIdx.x] = (uint32_t)__builtin_amdgcn_s_memrealtime(); dbg_[4608 + blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20); }
        if (lane == 0) dbg_[2048 + gw] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID
    }
#endif
    const uint32_t n_chunks = (a.n_samples + 15u) >> 4, n_slots = (n_chunks + 64u * U - 1u) / (64u * U);
    const int tail = (int)(a.n_samples & 15u);
    const uint32_t voff = (uint32_t)lane * 16u;
    const uint32_t last_valid = n_chunks - (n_slots - 1u) * 64u * U;  // chunks of a row's last slot that lie inside the row
    // a chunk is at least K slots long, so the loads in flight never reach past the chunk after the one being tallied
    uint32_t C = 1u;  // = ceil(K / n_slots), without a division (its VALU expansion would drag the ring's scalar state into VGPRs)
    while (C * n_slots < (uint32_t)K) ++C;
    uint32_t one;
    asm volatile("v_mov_b32 %0, 1" : "=v"(one));

    uint32_t ring_w = 0, ring_r = 0, inflight = 0;
    uint32_t cand = 0, n_cand = 0;  // lane k: the k-th wave-solver candidate of this wave since the last flush
    uint32_t easy = 0, n_easy = 0;  // the same for the candidates of the 16-lane solver (<= 2 active bases)
    uint32_t easy3 = 0, n_easy3 = 0;  //                                                  (>= 3 active bases)
#pragma unroll 1
    for (uint32_t e0 = B0; e0 < B1; e0 += (uint32_t)BV_S_REFCAP) {  // one pass unless the workgroup has more than BV_S_REFCAP sites
    const uint32_t e1 = (B1 - e0 > (uint32_t)BV_S_REFCAP) ? e0 + (uint32_t)BV_S_REFCAP : B1;
    __syncthreads();  // (a later pass: every wave is done with refl and the cursor)
    for (uint32_t i = 0; i < e1 - e0; i += BV_WAVE * NW)  // (uniform trip count: a divergent loop here makes the compiler treat the ring state as per-lane)
        if (i + threadIdx.x < e1 - e0) sh.refl[i + threadIdx.x] = a.ref_base[e0 + i + threadIdx.x];
    if (threadIdx.x == 0) sh.cursor = 0u;
    __syncthreads();

    // prefetch cursor: the next slot to request, inside the chunk [p_site, p_end)
    uint32_t p_site = 0, p_end = 0, p_j = 0;
    // the chunk being tallied and the one after it (drawn by the prefetch side, which runs ahead)
    uint32_t c_site = 0, c_end = 0, n_site = 0, n_end = 0;
    // bit 0: no chunks left to draw; bit 1: [c_site, c_end) is valid; bit 2: [n_site, n_end) is valid.  (One word, not three
    // bools: the optimiser merged the bools' stores into one store through a selected pointer, which kept them -- and with
    // them everything the prefetch step computes -- in scratch memory and VGPRs.)
    uint32_t st = 0;
    constexpr uint32_t P_DONE = 1u, C_HAVE = 2u, N_HAVE = 4u;
    const uint8_t *seg_bs = a.bs, *seg_q = a.q;  // CHAIN: the (biased) planes of the segment that holds p_site
    auto issue = [&]() __attribute__((always_inline)) {  // (out of line its captured state would live in scratch memory)
        if (p_site == p_end) {
            if (st & P_DONE) return;
            const uint32_t c = bv_lds_fetch_add_wave(cursor_lds, C);
            if (c >= e1 - e0) { st |= P_DONE; return; }
            p_site = e0 + c; p_end = (e1 - p_site > C) ? p_site + C : e1; p_j = 0;
            if (!(st & C_HAVE)) { c_site = p_site; c_end = p_end; st |= C_HAVE; }
            else { n_site = p_site; n_end = p_end; st |= N_HAVE; }
        }
        {
            if (CHAIN && p_j == 0u) {
                const BvChainC ch = bv_chain_const(a.ch);
                const uint32_t sg = bv_chain_seg(ch, (uint32_t)__builtin_amdgcn_readfirstlane((int)p_site));
                seg_bs = ch->bs[sg]; seg_q = ch->q[sg];
            }
            const size_t off = (size_t)p_site * a.pitch + (size_t)p_j * (1024u * U);
            const uint8_t *pb = bv_uniform_ptr(seg_bs + off), *pq = bv_uniform_ptr(seg_q + off);
            const uint32_t dst = ring_lds + ring_w * (U * BV_S_SLOT_WORDS * 4u);
#pragma unroll
            for (int u = 0; u < U; ++u) {  // slot layout: U KiB of calls, then U KiB of phreds
                // Every slot issues exactly 2 U loads (the counted waits rely on it): a KiB that lies wholly past the row's
                // end is "loaded" by lane 0 alone from the row's first bytes (its cells are masked in the tally)
                const bool any = p_j + 1u < n_slots || 64u * u < last_valid;
                const uint8_t *sb = any ? pb + 1024u * u : bv_uniform_ptr(seg_bs + (size_t)p_site * a.pitch);
                const uint8_t *sq = any ? pq + 1024u * u : bv_uniform_ptr(seg_q + (size_t)p_site * a.pitch);
                if (any ? (p_j + 1u < n_slots || (uint32_t)lane + 64u * u < last_valid) : lane == 0) {  // chunks past the row's end load nothing
                    bv_glds16(dst + 1024u * u, sb, voff);
                    bv_glds16(dst + 1024u * (U + u), sq, voff);
                }
            }
            ring_w = (ring_w + 1u == (uint32_t)K) ? 0u : ring_w + 1u;
            ++inflight;
            if (++p_j == n_slots) { p_j = 0; ++p_site; }
        }
    };
#pragma unroll 1
    for (int k = 0; k < K; ++k) issue();

#pragma unroll 1
    while (st & C_HAVE) {
#pragma unroll 1
    for (uint32_t site = c_site; site < c_end; ++site) {
#pragma unroll 1
        for (uint32_t j = 0; j < n_slots; ++j) {
            // the oldest slot in flight has landed once at most 2 (K - 1) younger loads are outstanding
            if (inflight == (uint32_t)K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * U * (K - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bv_u32x4 vb[U], vq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                vb[u] = *reinterpret_cast<const bv_u32x4 *>(ring + ring_r * (U * BV_S_SLOT_WORDS) + u * 256 + lane * 4);
                vq[u] = *reinterpret_cast<const bv_u32x4 *>(ring + ring_r * (U * BV_S_SLOT_WORDS) + (U + u) * 256 + lane * 4);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // in registers: the slot may be refilled
            ring_r = (ring_r + 1u == (uint32_t)K) ? 0u : ring_r + 1u;
            --inflight;
            issue();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t chunk = (j * U + u) * 64u + (uint32_t)lane;
                if (j + 1u == n_slots) bv_p1s_tally_slot<true>(vb[u], vq[u], chunk, n_chunks, tail, hist, one, lane);
                else bv_p1s_tally_slot<false>(vb[u], vq[u], chunk, n_chunks, tail, hist, one, lane);
            }
        }
        bv_lrt_sync<0>();

        // ---- the row's totals (LDS operations of one wave execute in order: the adds above are done)
        uint32_t c[4][2], facc[4], racc[4];
        bool bad = false;
        uint32_t q0_mask = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int qr = 0; qr < 2; ++qr) {
                const int q = (qr << 6) | lane;
                const uint32_t f = hist[(b << 7) | q], v = hist[((b | 4) << 7) | q];
                c[b][qr] = f + v;
                if (qr == 0) { facc[b] = f; racc[b] = v; } else { facc[b] += f; racc[b] += v; }
                if (qr == 1) bad |= (c[b][qr] != 0u) && (q >= BV_NQ_VALID);
            }
            if (__builtin_amdgcn_readfirstlane((int)c[b][0]) != 0) q0_mask |= 1u << b;
        }
        uint32_t fwd[4], rev[4];
        {
            const uint32_t v[8] = {facc[0], facc[1], facc[2], facc[3], racc[0], racc[1], racc[2], racc[3]};
            uint32_t t[8];
            bv_wave_sum8_u32(v, t, lane);
#pragma unroll
            for (int b = 0; b < 4; ++b) { fwd[b] = t[b]; rev[b] = t[4 + b]; }
        }
        uint32_t badq = (__ballot(bad) != 0ull) ? 1u : 0u;
        {
            uint32_t ovf_any = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t of = hist[BV_S_HWORDS + b], orv = hist[BV_S_HWORDS + 4 + b];
                fwd[b] += of; rev[b] += orv;
                ovf_any |= of | orv;
            }
            if (ovf_any) badq = 1u;
        }
        uint32_t depth[4], total = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) { depth[b] = fwd[b] + rev[b]; total += depth[b]; }

        // ---- candidate or not.  Not a candidate: nothing covered, or exactly one active base (basetype.cpp:135-139), it is
        // the reference base, none of its calls has phred 0, and the all-sites strand table is shallow.
        bool is_cand = false;
        uint32_t n_active = 0;
        if (total != 0u && !(a.flags & BV_FLAG_TALLY_ONLY)) {
            const int bsel = lane & 3;
            const bool act = (double)bv_sel4u(depth, bsel) / (int)total >= a.min_af;  // basetype.cpp:137, one base per lane
            const uint32_t act_mask = (uint32_t)(__ballot(act) & 0xFull);
            n_active = (uint32_t)__popc(act_mask);
            int ref = __builtin_amdgcn_readfirstlane((int)sh.refl[site - e0]);
            if (ref > 4) ref = 4;
            const bool one_ref = act_mask != 0u && (act_mask & (act_mask - 1u)) == 0u && ref < 4 && act_mask == (1u << ref);
            is_cand = !one_ref || (q0_mask & act_mask) != 0u;
            if (!is_cand) {
                // Fisher tables of (ref_fwd, ref_rev, alt_fwd, alt_rev): imax - imin + 1 (kfunc.c:253-257)
                const uint32_t rf = bv_sel4u(fwd, ref), rr = bv_sel4u(rev, ref);
                const uint32_t af = fwd[0] + fwd[1] + fwd[2] + fwd[3] - rf, ar = rev[0] + rev[1] + rev[2] + rev[3] - rr;
                const int n1_ = (int)(rf + rr), n_1 = (int)(rf + af), n = (int)total;
                const int imax = n_1 < n1_ ? n_1 : n1_;
                int imin = n1_ + n_1 - n;
                if (imin < 0) imin = 0;
                (void)ar;
                is_cand = (imax - imin + 1) > BV_S_SIMPLE_MAX_TABLES;
            }
        }
        uint32_t nb = 0;
        if (is_cand) {
            // every non-empty (base, phred < 128) bin, in (base, phred) order -- the order of bv_prologue_wave
            uint32_t *dst = a.bins + (size_t)site * BV_S_BIN_STRIDE;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int qr = 0; qr < 2; ++qr) {
                    const int q = (qr << 6) | lane;
                    const bool valid = c[b][qr] != 0u;
                    const unsigned long long m = __ballot(valid);
                    const uint32_t pos = nb + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (valid) dst[pos] = ((((uint32_t)b << 7) | (uint32_t)q) << 16) | c[b][qr];
                    nb += (uint32_t)__popcll(m);
                }
            }
            // four per wave (bv_solver16.h) unless the site needs what only the wave solver has: the ordered replay of a
            // shallow site, the literal 0/0 arithmetic of phred-0 calls or of min_af <= 0, or more than 128 bins
            const bool is_easy = q0_mask == 0u && total > (uint32_t)BV_ORD_MAX && nb <= (uint32_t)BV_G16_MAX_BINS && a.min_af > 0.0 &&
                                 !(a.flags & BV_FLAG_WAVE_SOLVER);
            if (is_easy && n_active <= 2u) { easy = ((uint32_t)lane == n_easy) ? site : easy; ++n_easy; }
            else if (is_easy) { easy3 = ((uint32_t)lane == n_easy3) ? site : easy3; ++n_easy3; }
            else { cand = ((uint32_t)lane == n_cand) ? site : cand; ++n_cand; }
        }
        {
            // 48-byte summary: 12 lanes, one dword each
            const uint32_t fl = q0_mask | (badq ? BV_SUM_BADQ : 0u) | (is_cand ? BV_SUM_CAND : 0u);
            uint32_t w = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                w = (lane == b) ? fwd[b] : w;
                w = (lane == 4 + b) ? rev[b] : w;
            }
            w = (lane == 8) ? nb : w;
            w = (lane == 9) ? fl : w;
            if (lane < 12) reinterpret_cast<uint32_t *>(&a.summ[site])[lane] = w;
        }
        // ---- hand the histogram back, zeroed
        {
            uint4 *h4 = reinterpret_cast<uint4 *>(hist);
#pragma unroll
            for (int i = 0; i < BV_S_HWORDS / 4 / BV_WAVE; ++i) h4[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
            if (lane < 2) h4[BV_S_HWORDS / 4 + lane] = make_uint4(0, 0, 0, 0);
        }
        bv_lrt_sync<0>();
        if (n_cand == 64u) {
            // flush this wave's candidates: one atomic reserves their places in the list
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&a.counters[BV_CTR_CANDS], n_cand);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if ((uint32_t)lane < n_cand) a.cand_list[base + (uint32_t)lane] = cand;
            n_cand = 0;
        }
        if (n_easy == 64u) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&a.counters[BV_CTR_EASY], n_easy);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if ((uint32_t)lane < n_easy) a.easy_list[base + (uint32_t)lane] = easy;
            n_easy = 0;
        }
        if (n_easy3 == 64u) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&a.counters[BV_CTR_EASY3], n_easy3);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if ((uint32_t)lane < n_easy3) a.easy3_list[base + (uint32_t)lane] = easy3;
            n_easy3 = 0;
        }
    }  // sites of the chunk
        if (st & N_HAVE) { c_site = n_site; c_end = n_end; st &= ~N_HAVE; }
        else st &= ~C_HAVE;
    }  // chunks
    }  // passes
    // what is left of this wave's candidate lists (the wait for an atomic's return drains the ring: empty by now)
    if (n_cand != 0u) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&a.counters[BV_CTR_CANDS], n_cand);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if ((uint32_t)lane < n_cand) a.cand_list[base + (uint32_t)lane] = cand;
    }
    if (n_easy != 0u) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&a.counters[BV_CTR_EASY], n_easy);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if ((uint32_t)lane < n_easy) a.easy_list[base + (uint32_t)lane] = easy;
    }
    if (n_easy3 != 0u) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&a.counters[BV_CTR_EASY3], n_easy3);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if ((uint32_t)lane < n_easy3) a.easy3_list[base + (uint32_t)lane] = easy3;
    }
#ifdef BV_TEAM_DEBUG
    if (gridDim.x * NW <= 2048u && lane == 0) dbg_[gw] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
}



// ---- the ordinary candidates, four per wave: one site per group of 16 lanes (bv_solver16.h), in two phases -- the LRT,
// then everything that follows it (bv_site_lrt_g16 / bv_site_tail_g16).  Measured as two kernels too (one per phase, and a third
// for the non-candidates): 0.125 ms per 100 k sites against 0.105 ms for the one below -- two kernel boundaries and two tails more.
// Jobs (four sites of one list) are numbered with the sites of three or four active bases
// first (several times the EM runs: longest jobs first).  Jobs and workgroups are dealt to BV_TICKET_SLICES slices (job j and
// workgroup w belong to slice j, w mod BV_TICKET_SLICES); a wave's first job is its rank in the slice, every further one is
// drawn from the slice's ticket counter when the wave gets there -- no draw at the start of the kernel, where all waves
// would queue on one address (~88 M atomics/s: 35 us for 3072 waves), and eight addresses instead of one after that.
#define BV_P1S_SOLVE16_NW 4
struct __attribute__((aligned(16))) BvP1sSolve16Shared {
    double tab_hit[BV_QBINS], tab_miss[BV_QBINS];
    BvP1sWaveScratch ws[BV_P1S_SOLVE16_NW];
    uint32_t vl[BV_P1S_SOLVE16_NW][64];                    // the wave's variant sites since the last flush
};
#define BV_P1S_SOLVE16_OCC 3
struct BvP1sJob {
    const uint32_t *list;
    uint32_t idx;
    bool active;
};
// the k-th job of this wave's slice; its next k
struct BvP1sTickets {
    uint32_t slice, n_slices, slice_waves, k;
    uint32_t *ctr;
    __device__ __forceinline__ void init(uint32_t *counters_base, uint32_t wave) {
        n_slices = gridDim.x < BV_TICKET_SLICES ? gridDim.x : BV_TICKET_SLICES;  // (every slice needs a workgroup)
        slice = blockIdx.x % n_slices;
        const uint32_t slice_wgs = (gridDim.x - slice + n_slices - 1u) / n_slices;
        slice_waves = slice_wgs * BV_P1S_SOLVE16_NW;
        k = (blockIdx.x / n_slices) * BV_P1S_SOLVE16_NW + wave;
        ctr = counters_base + slice * BV_CTR_STRIDE;
    }
    __device__ __forceinline__ uint32_t job() const { return slice + n_slices * k; }
    __device__ __forceinline__ void next(int lane) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(ctr, 1u);
        k = slice_waves + (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
    }
};
__device__ __forceinline__ BvP1sJob bv_p1s_job(const BvP1ShortArgs &a, uint32_t job, uint32_t n_easy, uint32_t n_easy3, int grp) {
    const uint32_t j3 = (n_easy3 + 3u) >> 2;
    BvP1sJob j;
    if (job < j3) { j.list = a.easy3_list; j.idx = job * 4u + (uint32_t)grp; j.active = j.idx < n_easy3; }
    else { j.list = a.easy_list; j.idx = (job - j3) * 4u + (uint32_t)grp; j.active = j.idx < n_easy; }
    return j;
}

// One kernel: a wave takes a job through both phases (the LRT's results stay in registers), and when the jobs are gone it
// takes blocks of 64 non-candidate sites, one per lane.  (lrt + tail + simple site: 60 KB of code, inside the instruction
// cache of 64 KB per pair of CUs; round 2's kernel was 207 KB -- two inlined copies of the Fisher test, the log-factorial
// series at ~60 call sites -- and 1.2 % of its instruction fetches missed.)
__global__ __launch_bounds__(BV_WAVE *BV_P1S_SOLVE16_NW, BV_P1S_SOLVE16_OCC) void bv_p1s_solve16_kernel(BvP1ShortArgs a) {
    __shared__ BvP1sSolve16Shared sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_easy = a.counters[BV_CTR_EASY], n_easy3 = a.counters[BV_CTR_EASY3];
    const uint32_t n_jobs = ((n_easy + 3u) >> 2) + ((n_easy3 + 3u) >> 2);
    for (int i = tid; i < BV_QBINS; i += BV_WAVE * BV_P1S_SOLVE16_NW) {
        sh.tab_hit[i] = a.tables->hit[i];
        sh.tab_miss[i] = a.tables->miss[i];
    }
    __syncthreads();
    BvSolveArgs sa;
    sa.ref_base = a.ref_base; sa.out = a.out; sa.var_list = a.var_list; sa.counters = a.counters;
    sa.min_af = a.min_af; sa.flags = a.flags;
    sa.lnfact.t = a.tables->lnfact; sa.lnfact.n = (int)a.tables->lnfact_n;
    sa.loghit = a.tables->loghit; sa.logmiss = a.tables->logmiss;
    sa.bs = a.bs; sa.q = a.q; sa.pitch = a.pitch; sa.n_samples = a.n_samples;
    const int grp = lane >> 4, gl = lane & 15;
    uint32_t *scratch = sh.ws[wave].grp[grp], *vl = sh.vl[wave];
    uint32_t n_vl = 0;  // variant sites in vl[]
    auto flush_vl = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&a.counters[BV_CTR_VARIANTS], n_vl);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        bv_lrt_sync<0>();
        if ((uint32_t)lane < n_vl) a.var_list[base + (uint32_t)lane] = vl[lane];
        bv_lrt_sync<0>();
        n_vl = 0;
    };
    // ---- first the candidates only the wave solver takes (shallow sites replayed in sample order, phred-0 calls, more than 128
    // bins, min_af <= 0): none in most batches, all of them in a cohort of <= 64 samples.  Dealt round-robin over the grid's waves;
    // one wave, one site, the solver of bv_solver.h on the exported bins.  (A kernel of their own until round 3: mostly empty, it
    // still cost a launch, a kernel boundary and -- beside the other lane's kernels -- 50-110 us of workgroups queueing to leave.)
    {
        const uint32_t n_cand = a.counters[BV_CTR_CANDS];
        const uint32_t n_waves = gridDim.x * BV_P1S_SOLVE16_NW, gw = blockIdx.x * BV_P1S_SOLVE16_NW + (uint32_t)wave;
        uint32_t *bin_code = sh.ws[wave].w.raw, *bin_cnt = bin_code + BV_SLOTS * BV_WAVE, *hq = bin_code + 2 * BV_SLOTS * BV_WAVE;
        BvSolverScratch *sv = &sh.ws[wave].w.sc;
        constexpr int REC_WORDS = (int)(sizeof(bv_site_result) / 4);
#pragma unroll 1
        for (uint32_t t = gw; t < n_cand; t += n_waves) {
            const uint32_t site = a.cand_list[t];
            const BvSiteSummary sm = a.summ[site];
            BvSiteSums S;
#pragma unroll
            for (int b = 0; b < 4; ++b) { S.fwd[b] = sm.fwd[b]; S.rev[b] = sm.rev[b]; }
            S.q0_mask = sm.flags & BV_SUM_Q0_MASK;
            S.badq = (sm.flags & BV_SUM_BADQ) ? 1u : 0u;
            if (lane < REC_WORDS) reinterpret_cast<uint32_t *>(&sv->res)[lane] = 0u;
            {
                uint4 *z = reinterpret_cast<uint4 *>(hq);
#pragma unroll
                for (int i = 0; i < 4 * 128 / 4 / BV_WAVE; ++i) z[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
            }
            bv_lrt_sync<0>();
            // exported bins -> merged counts for the rank sum (all of them) and the EM's bins (phred <= 93), order kept
            const uint32_t *src = a.bins + (size_t)site * BV_S_BIN_STRIDE;
            uint32_t nb = 0;
            for (uint32_t i0 = 0; i0 < sm.nb; i0 += BV_WAVE) {
                const uint32_t i = i0 + (uint32_t)lane;
                const bool have = i < sm.nb;
                const uint32_t w = have ? src[i] : 0u;
                const uint32_t code = w >> 16, cnt = w & 0xFFFFu;
                if (have) hq[code] = cnt;
                const bool valid = have && (code & 127u) < BV_NQ_VALID;
                const unsigned long long m = __ballot(valid);
                const uint32_t pos = nb + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (valid) { bin_code[pos] = code; bin_cnt[pos] = cnt; }
                nb += (uint32_t)__popcll(m);
            }
            S.nb = nb;
            bv_lrt_sync<0>();
            BvHqMerged H{hq};
            if (a.ch != nullptr) {  // chained launch: the ordered gather of a shallow site reads the segment's (biased) planes
                const BvChainC ch = bv_chain_const(a.ch);
                const uint32_t sg = bv_chain_seg(ch, (uint32_t)__builtin_amdgcn_readfirstlane((int)site));
                sa.bs = ch->bs[sg]; sa.q = ch->q[sg];
            }
            if (bv_site_solve<false, BvHqMerged, true>(sa, site, S, bin_code, bin_cnt, H, sv, sh.tab_hit, sh.tab_miss, lane)) {
                if (lane == 0) vl[n_vl] = site;
                if (++n_vl > 60u) flush_vl();
            }
            bv_lrt_sync<0>();
        }
        if (n_vl) flush_vl();  // (the groups' scratch starts clean of the list logic either way; the flush keeps the phases apart)
    }
    BvP1sTickets tk;
    tk.init(a.counters + BV_CTR_TICKET_A, (uint32_t)wave);
    for (; tk.job() < n_jobs; tk.next(lane)) {
        const BvP1sJob jb = bv_p1s_job(a, tk.job(), n_easy, n_easy3, grp);
        // (s_setprio for the jobs of three or four active bases -- the kernel's critical path, one runs 50-83 us of the kernel's
        // 94 -- was measured: no change; their time is dependent-latency, not lost issue slots)
        bool variant = false;
        uint32_t site = 0;
        if (jb.active) {
            site = jb.list[jb.idx];
            const uint32_t *src = a.bins + (size_t)site * BV_S_BIN_STRIDE;
            uint32_t nb, badq;
            BvG16Lrt pre;
            {
                // phase 1 needs the depths only: the strand totals are read again for phase 2 (eight registers that would
                // otherwise sit through the EMs, in a kernel at its register limit)
                const BvSiteSummary sm = a.summ[site];
                uint32_t depth[4], total = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) { depth[b] = sm.fwd[b] + sm.rev[b]; total += depth[b]; }
                nb = sm.nb;
                badq = (sm.flags & BV_SUM_BADQ) ? 1u : 0u;
                BvG16Bins B;
                B.hit = sh.tab_hit; B.miss = sh.tab_miss; B.loghit = sa.loghit; B.logmiss = sa.logmiss;
                B.pm = reinterpret_cast<double *>(scratch) + gl;
#pragma unroll
                for (int s = 0; s < BV_G16_SLOTS; ++s) {
                    const uint32_t i = (uint32_t)(s * 16 + gl);
                    B.w[s] = i < nb ? src[i] : 0u;
                }
                variant = bv_site_lrt_g16(sa, site, depth, total, badq, B, scratch, lane, &pre);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the record's first version is out before phase 2 patches it
            BvSiteSums S;
            {
                const volatile BvSiteSummary *vs = &a.summ[site];
#pragma unroll
                for (int b = 0; b < 4; ++b) { S.fwd[b] = vs->fwd[b]; S.rev[b] = vs->rev[b]; }
            }
            S.q0_mask = 0; S.nb = nb; S.badq = badq;
            bv_site_tail_g16(sa, site, S, src, nb, scratch, lane, &pre);
        }
        // the wave's variant sites of this round, in group order
        const unsigned long long vm = __ballot(variant && gl == 0);
        if (variant && gl == 0) vl[n_vl + (uint32_t)__popcll(vm & ((1ull << lane) - 1ull))] = site;
        n_vl += (uint32_t)__popcll(vm);
        if (n_vl > 60u) flush_vl();
    }
    if (n_vl) flush_vl();
    // ---- the non-candidate sites, one lane per site, in blocks of 64 drawn like the jobs
    const uint32_t n_blocks = (a.n_sites + 63u) >> 6;
    BvP1sTickets tb;
    tb.init(a.counters + BV_CTR_TICKET_B, (uint32_t)wave);
    for (; tb.job() < n_blocks; tb.next(lane)) {
        const uint32_t site = tb.job() * 64u + (uint32_t)lane;
        if (site < a.n_sites) bv_p1s_simple_site(a, sa.lnfact, site);
    }
}

// ------------------------------------------------------------------------------ chained launches (bv_engine_submit_many)
__global__ void bv_chain_gather_ref_kernel(const BvChain *ch, uint32_t n_sites, uint8_t *ref_cat) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const BvChainC c = bv_chain_const(ch);
    if (t < n_sites) ref_cat[t] = c->ref_base[bv_chain_seg(c, t)][t];
}
// one 16-byte piece of a record per thread (13 per record)
__global__ void bv_chain_scatter_out_kernel(const BvChain *ch, uint32_t n_sites, const bv_site_result *out_cat) {
    constexpr uint32_t PIECES = (uint32_t)(sizeof(bv_site_result) / 16);
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = (uint32_t)(i / PIECES), piece = (uint32_t)(i % PIECES);
    const BvChainC c = bv_chain_const(ch);
    if (t < n_sites)
        reinterpret_cast<uint4 *>(&c->out[bv_chain_seg(c, t)][t])[piece] = reinterpret_cast<const uint4 *>(&out_cat[t])[piece];
}
void bv_launch_chain_gather_ref(const BvChain *ch, uint32_t n_sites, uint8_t *ref_cat, hipStream_t stream) {
    hipLaunchKernelGGL(bv_chain_gather_ref_kernel, dim3((n_sites + 255u) / 256u), dim3(256), 0, stream, ch, n_sites, ref_cat);
}
void bv_launch_chain_scatter_out(const BvChain *ch, uint32_t n_sites, const bv_site_result *out_cat, hipStream_t stream) {
    const uint64_t n = (uint64_t)n_sites * (sizeof(bv_site_result) / 16);
    hipLaunchKernelGGL(bv_chain_scatter_out_kernel, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, stream, ch, n_sites, out_cat);
}

// ------------------------------------------------------------------------------ launchers
template <int NW, int K, int U = 1>
static void bv_launch_p1s_stream_cfg(const BvP1ShortArgs &a, hipStream_t stream, uint32_t wg_per_cu) {
    const uint32_t cu = a.n_cu ? a.n_cu : 256u;
    uint32_t grid = cu * wg_per_cu;
    const uint32_t need = (a.n_sites + NW - 1) / NW;  // at least one site per wave
    if (grid > need) grid = need > 0 ? need : 1;
    const uint32_t cap = (a.flags >> 16) & 0xFFu;  // BV_FLAG_GRID_LIMIT
    if (cap && grid > cap) grid = cap;
    if (a.ch != nullptr) hipLaunchKernelGGL((bv_p1s_stream_kernel<NW, K, U, true>), dim3(grid), dim3(BV_WAVE * NW), 0, stream, a);
    else hipLaunchKernelGGL((bv_p1s_stream_kernel<NW, K, U, false>), dim3(grid), dim3(BV_WAVE * NW), 0, stream, a);
}
void bv_launch_p1s_stream(const BvP1ShortArgs &a, hipStream_t stream) {
    // (Measured and dropped, rounds 2-3: 16 waves / CU with 2 slots in flight, 8 with 5, two workgroups of 4 waves, 6 or 5 waves
    // per workgroup, slots of 1 KiB per plane at 8 or 12 waves / CU.)
    // Slots of 2 KiB per plane, 3 slots per wave (8 KiB in flight), 8 waves per CU: measured best (+4-5 % over 1 KiB
    // slots; 4 slots of 2 KiB or 8 of 1 KiB need 82 KB of LDS per workgroup -- one workgroup per CU, 0.43 of peak)
    // One workgroup of 8 waves per CU (not two of 4): the waves of a workgroup share its sites through the LDS cursor, and the
    // two waves that share a SIMD -- the arbiter's favourite and the other -- must be in the same workgroup for that to even
    // them out.
    // (Also under BV_FLAG_LANES, where the other lane's kernels run beside this one.  Measured there, interleaved A/B: this
    // shape 170.5 M sites/s, two workgroups of 4 waves with fixed ranges 167-171 M, two of 4 with the cursor 157 M; 176-178 M
    // since the wave-solver candidates moved into the solve kernel: DESIGN 4.2c.)
    bv_launch_p1s_stream_cfg<8, 3, 2>(a, stream, 1);
}
// `beside_stream`: the kernels will run beside a streaming kernel (the next chunk's pass 1 or an earlier chunk's pass 2) whose
// two workgroups per CU leave 28-32 KB of LDS: grids of ONE workgroup per CU then -- with more, whatever starts in the
// gap between two streaming kernels takes the LDS of a streaming workgroup for its whole (persistent) life.
void bv_launch_p1s_solve(const BvP1ShortArgs &a, hipStream_t stream, bool beside_stream) {
    const uint32_t cu = a.n_cu ? a.n_cu : 256u;
    const uint32_t cap = (a.flags >> 16) & 0xFFu;  // BV_FLAG_GRID_LIMIT
    uint32_t grid16 = beside_stream ? cu : cu * (uint32_t)BV_P1S_SOLVE16_OCC * (4u / BV_P1S_SOLVE16_NW);
    const uint32_t need16 = (a.n_sites + 4 * BV_P1S_SOLVE16_NW - 1) / (4 * BV_P1S_SOLVE16_NW);  // four sites per wave
    if (grid16 > need16) grid16 = need16 > 0 ? need16 : 1;
    if (cap && grid16 > cap) grid16 = cap;
    hipLaunchKernelGGL(bv_p1s_solve16_kernel, dim3(grid16), dim3(BV_WAVE * BV_P1S_SOLVE16_NW), 0, stream, a);
}
