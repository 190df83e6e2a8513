// bv_tiles.hip -- sample-axis tile mode (BASELINE config #5; SURVEY.md sections 5 and 8e).
//
// The reference stores the sample axis in batchfiles of B samples each (`-B/--batch-count`,
// src/basetype_caller.cpp:419-453) and re-joins them per site.  Because every per-site quantity
// of the path is a function of tallies that are ADDITIVE over samples, a site does not have to be
// presented as one full row: column tiles [n_sites][tile_width] can be streamed (e.g. from host
// DRAM, one batchfile at a time) and accumulated into per-site state in HBM; the solve runs once
// at the end.  State per site (u32 words):
//     H1 [ (rev<<2|base) << 8 | phred ]        2048   same layout as the LDS histogram of pass 1
//     Hm [ base << 8  | mapq ]                 1024   mapq tally per called base
//     Hr [ base * W + rank ]                   4 x W  read-position ranks 0..W-1 per called base; W = 1024 unless the job
//                                                     announced longer reads (bv_engine_tiles_begin, with_ranks > 1)
//     Hg [ (group*4 + base) << 7 | phred ]     512 per pop-group
//     Ord                                      BV_TS_ORD_WORDS: the site's first covered cells as (sample index, call, phred, group),
//                                              then the same list per pop-group
// Keeping Hm/Hr per BASE (not per REF/ALT class) is what makes a single sweep enough: the alt set
// is only known after the last tile, and any (ref, alts) partition can be read off per-base tallies.
// A site of at most 64 covered samples has ALL its cells in Ord: at finish they are put in sample order and the site (and its
// pop-groups) replay the reference's per-sample EM literally, as the row kernels do from the row (sum / argmin order,
// src/algorithm.h:24-41): exact ties fall as the reference's do.
// A rank at or beyond W does not fit Hr: the cell goes to a pool of (site, base, rank) entries, and a site that has such cells
// forms its ReadPosRankSum from Hr plus its pool entries -- exact (src/basetype.cpp:201-242), slower.  Only a site with more
// than BV_TS_OVF_LIST such cells, or a pool that overflowed, still gets BV_SITE_RPR_RANGE and a NaN (announce the read length).
//
// Tally: one thread per 16-byte chunk of the tile (coalesced loads), covered cells go to the site's state with
// global atomics.  This per-site-state realisation is the FALLBACK of the tile mode (for jobs whose joined planes do
// not fit the HBM even in site chunks); by default the engine joins the tiles into rows resident in HBM
// (bv_tile_scatter_kernel below) and runs the ordinary two passes on them.
#include "bv_solver.h"
#include "bv_tally.h"  // bv_rpr_rank_mask

#define BV_TS_H1 0u
#define BV_TS_HM 2048u
#define BV_TS_HR 3072u /* [4][rank_win] words; the pop-group tallies follow at hg_off = 3072 + 4 * rank_win */

// One thread per 16-byte chunk of the tile, chunks numbered row after row: consecutive lanes read consecutive chunks of a
// row (coalesced 16-byte loads of all four planes) and add their covered cells to the row's state with global atomics.
__global__ __launch_bounds__(256) void bv_tile_tally_kernel(BvTileArgs a) {
    const uint32_t n_chunks = (a.width + 15u) >> 4;
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (uint64_t)a.n_sites * n_chunks) return;
    const uint32_t site = (uint32_t)(g / n_chunks), ch = (uint32_t)(g % n_chunks);
    uint32_t *S = a.state + (size_t)site * a.stride;
    const size_t at = (size_t)site * a.pitch + (size_t)ch * 16u;
    const bv_u32x4 vb = *reinterpret_cast<const bv_u32x4 *>(a.bs + at);
    const bv_u32x4 vq = *reinterpret_cast<const bv_u32x4 *>(a.q + at);
    bv_u32x4 vm = bv_u32x4{0u, 0u, 0u, 0u}, vr0 = vm, vr1 = vm;
    if (a.mapq) {
        vm = *reinterpret_cast<const bv_u32x4 *>(a.mapq + at);
        vr0 = *reinterpret_cast<const bv_u32x4 *>(a.rpr + at);        // uint16 plane: elements at .. at + 7
        vr1 = *reinterpret_cast<const bv_u32x4 *>(a.rpr + at + 8);    //               elements at + 8 .. at + 15
    }
    const uint32_t wb[4] = {vb.x, vb.y, vb.z, vb.w}, wq[4] = {vq.x, vq.y, vq.z, vq.w}, wm[4] = {vm.x, vm.y, vm.z, vm.w};
    const uint32_t wr[8] = {vr0.x, vr0.y, vr0.z, vr0.w, vr1.x, vr1.y, vr1.z, vr1.w};
    uint32_t maxr = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t smp = ch * 16u + j;
        const uint32_t c = (wb[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        if (smp >= a.width || c > 7u) continue;
        const uint32_t q = (wq[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        const uint32_t b = c & 3u;
        atomicAdd(&S[BV_TS_H1 + ((c << 8) | q)], 1u);
        if (a.mapq) {
            const uint32_t mq = (wm[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            const uint32_t r = (wr[j >> 1] >> (16 * (j & 1))) & bv_rpr_rank_mask(a.rpr_tag);
            atomicAdd(&S[BV_TS_HM + ((b << 8) | mq)], 1u);
            maxr = max(maxr, r);
            if (r < a.rank_win) atomicAdd(&S[BV_TS_HR + b * a.rank_win + r], 1u);
            else {  // beyond the window: into the pool (exact path at finish)
                const uint32_t k = atomicAdd(&a.ovf[0], 1u);
                if (k < a.ovf_cap) { a.ovf[2u + 2u * k] = site; a.ovf[3u + 2u * k] = (b << 16) | r; }
            }
        }
        uint32_t gi = BV_NO_GROUP;
        if (a.n_groups) {
            gi = a.group_id[smp];
            if (gi < a.n_groups) atomicAdd(&S[a.hg_off + (((gi * 4u + b) << 7) | min(q, 127u))], 1u);
        }
        // the site's covered cells while they are few (a plain look first: a deep site stops paying for the atomic), and the
        // same per pop-group (a shallow group of a deep site ties as easily as a shallow site)
        // (counted up to BV_ORD_MAX + 1: a count of exactly BV_ORD_MAX must mean "complete", not "stopped counting")
        if (__builtin_nontemporal_load(&S[a.ord_off]) <= (uint32_t)BV_ORD_MAX) {
            const uint32_t k = atomicAdd(&S[a.ord_off], 1u);
            if (k < (uint32_t)BV_ORD_MAX) {
                S[a.ord_off + 4u + 2u * k] = a.col0 + smp;
                S[a.ord_off + 5u + 2u * k] = (c << 8) | q | (gi << 16);
            }
        }
        if (gi < a.n_groups) {
            uint32_t *GL = S + a.ord_off + (1u + gi) * BV_TS_ORD_WORDS;
            if (__builtin_nontemporal_load(&GL[0]) <= (uint32_t)BV_ORD_MAX) {
                const uint32_t k = atomicAdd(&GL[0], 1u);
                if (k < (uint32_t)BV_ORD_MAX) {
                    GL[4u + 2u * k] = a.col0 + smp;
                    GL[5u + 2u * k] = (c << 8) | q | (gi << 16);
                }
            }
        }
    }
    if (a.mapq && maxr) atomicMax(&a.maxr[site], maxr);
}

struct __attribute__((aligned(16))) BvTileFinishShared {
    uint32_t hist[BV_H2_WORDS];
    double tab_hit[BV_QBINS], tab_miss[BV_QBINS];
    BvSolverScratch sc;
    uint32_t bin_code[BV_SLOTS * BV_WAVE];  // group calls
    uint32_t bin_cnt[BV_SLOTS * BV_WAVE];
    uint32_t cells[BV_ORD_MAX];             // a shallow site's covered cells in sample order: call << 8 | phred | group << 16
    uint32_t n_cells;                       // ... how many (0: the site is not shallow / its list is incomplete)
    uint32_t ovf[BV_TS_OVF_LIST];           // the site's read-position ranks beyond the window: class << 16 | rank
};

// The site's covered cells in sample order, from its list in the state: every lane takes one entry, its place is the number of
// entries with a smaller sample index (indices are distinct).  Returns how many cells the list holds (> BV_ORD_MAX: incomplete).
__device__ __forceinline__ uint32_t bv_tile_sorted_cells(const uint32_t *L, uint32_t *cells, int lane) {
    const uint32_t n = L[0];
    if (n == 0u || n > (uint32_t)BV_ORD_MAX) return n;
    const bool have = (uint32_t)lane < n;
    const uint32_t idx = have ? L[4 + 2 * lane] : 0xFFFFFFFFu, cell = have ? L[5 + 2 * lane] : 0u;
    uint32_t place = 0;
    for (uint32_t j = 0; j < n; ++j) place += ((uint32_t)__shfl((int)idx, (int)j) < idx) ? 1u : 0u;
    if (have) cells[place] = cell;
    bv_lrt_sync<0>();
    return n;
}
// WHO supplies a shallow site's ordered cells here: the list above (bv_site_solve asks through `ordered`)
struct BvTileWork : BvSoloWork {
    const uint32_t *cells;
    uint32_t n_cells;
    __device__ __forceinline__ uint32_t ordered(uint32_t, uint16_t *ord, int lane) const {
        if ((uint32_t)lane < n_cells) ord[lane] = (uint16_t)(cells[lane] & 0xFFFFu);
        return n_cells;
    }
};

// One wave per site: the record from the accumulated state.
__global__ __launch_bounds__(BV_WAVE) void bv_tile_finish_kernel(BvTileFinishArgs a) {
    __shared__ BvTileFinishShared sh;
    const int lane = threadIdx.x;
    const uint32_t site = blockIdx.x;
    const uint32_t *S = a.state + (size_t)site * a.stride;
    {
        const uint4 *g4 = reinterpret_cast<const uint4 *>(S + BV_TS_H1);
        uint4 *h4 = reinterpret_cast<uint4 *>(sh.hist);
#pragma unroll
        for (int i = 0; i < BV_H2_WORDS / 4 / BV_WAVE; ++i) h4[i * BV_WAVE + lane] = g4[i * BV_WAVE + lane];
        for (int i = lane; i < BV_QBINS; i += BV_WAVE) {
            sh.tab_hit[i] = a.tables->hit[i];
            sh.tab_miss[i] = a.tables->miss[i];
        }
    }
    bv_lrt_sync<0>();
    BvSolveArgs sa;
    sa.ref_base = a.ref_base; sa.out = a.out; sa.var_list = a.var_list; sa.counters = a.counters;
    sa.min_af = a.min_af; sa.flags = 0;
    sa.lnfact.t = a.tables->lnfact; sa.lnfact.n = (int)a.tables->lnfact_n;
    sa.loghit = a.tables->loghit; sa.logmiss = a.tables->logmiss;
    sa.bs = nullptr; sa.q = nullptr; sa.pitch = 0; sa.n_samples = 0;  // no rows here: a shallow site's cells come from its list
    BvTileWork work;
    {
        const uint32_t n_list = bv_tile_sorted_cells(S + a.ord_off, sh.cells, lane);
        work.cells = sh.cells;
        work.n_cells = n_list <= (uint32_t)BV_ORD_MAX ? n_list : 0u;
    }
    bv_solve_site_wave<true, BvTileWork>(sa, site, (BV_LDS uint32_t *)sh.hist, (BV_LDS uint32_t *)nullptr, (BV_LDS uint32_t *)nullptr,
                                         (BV_LDS BvSolverScratch *)&sh.sc, (BV_LDS const double *)sh.tab_hit,
                                         (BV_LDS const double *)sh.tab_miss, lane, work);
    // what the solver decided (its staged record is still in LDS)
    const int n_alt = sh.sc.res.n_alt;
    if (n_alt == 0) return;
    int ref = a.ref_base[site];
    if (ref > 4) ref = 4;
    uint32_t depth[4] = {sh.sc.res.depth[0], sh.sc.res.depth[1], sh.sc.res.depth[2], sh.sc.res.depth[3]};
    uint32_t alt_mask = 0;
    unsigned long long n1 = (ref < 4) ? bv_sel4u(depth, ref) : 0ull, n2 = 0;
    int comb = ref, nc = 1;
#pragma unroll
    for (int k = 0; k < BV_MAX_ALT; ++k) {
        if (k < n_alt) {
            const int b = sh.sc.res.alt[k] & 3;
            alt_mask |= 1u << b;
            n2 += bv_sel4u(depth, b);
            comb |= b << (3 * nc);
            ++nc;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);  // the record store of the solver has been issued before the field updates below

    if (a.have_ranks) {
        // MQRankSum / ReadPosRankSum from per-base tallies (caller.cpp:1151-1154)
        unsigned long long below = 0, twoR = 0;
        for (int w = 0; w < 4; ++w) {
            const int v = w * 64 + lane;
            uint32_t rv = 0, av = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t c = S[BV_TS_HM + ((b << 8) | v)];
                if (b == ref) rv += c;
                else if ((alt_mask >> b) & 1u) av += c;
            }
            twoR += bv_ranksum_window(rv, av, n1 + n2, below, lane);
        }
        const double mq_ph = bv_ranksum_phred(twoR, n1, n2);
        double rp_ph = __builtin_nan("");
        const uint32_t maxr = a.maxr[site];
        bool in_range = maxr < a.rank_win;
        uint32_t n_ovf = 0;
        if (!in_range) {
            // the site's cells beyond the window, from the pool: class (0 REF, 1 ALT; others dropped) << 16 | rank
            const uint32_t n_pool = a.ovf[0];
            in_range = n_pool <= a.ovf_cap;  // (a pool that overflowed lost cells: no exact answer)
            for (uint32_t i0 = 0; in_range && i0 < n_pool; i0 += BV_WAVE) {
                const uint32_t i = i0 + (uint32_t)lane;
                uint32_t w = 0;
                bool mine = false;
                if (i < n_pool && a.ovf[2u + 2u * i] == site) {
                    w = a.ovf[3u + 2u * i];
                    const int b = (int)(w >> 16);
                    mine = b == ref || ((alt_mask >> b) & 1u);
                    w = ((b == ref) ? 0u : 1u) << 16 | (w & 0xFFFFu);
                }
                const unsigned long long m = __ballot(mine);
                const uint32_t pos = n_ovf + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if (mine && pos < BV_TS_OVF_LIST) sh.ovf[pos] = w;
                n_ovf += (uint32_t)__popcll(m);
                if (n_ovf > BV_TS_OVF_LIST) in_range = false;
            }
            bv_lrt_sync<0>();
        }
        if (in_range) {
            below = 0; twoR = 0;
            const uint32_t top = maxr < a.rank_win ? maxr : a.rank_win - 1u;
            for (uint32_t w = 0; w * 64u <= top; ++w) {  // ranks beyond the site's largest hold nothing
                const uint32_t v = w * 64u + (uint32_t)lane;
                uint32_t rv = 0, av = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t c = (v < a.rank_win) ? S[BV_TS_HR + (uint32_t)b * a.rank_win + v] : 0u;
                    if (b == ref) rv += c;
                    else if ((alt_mask >> b) & 1u) av += c;
                }
                twoR += bv_ranksum_window(rv, av, n1 + n2, below, lane);
            }
            // ... then the values beyond the window, every one larger than all of the window's: for an entry of value v,
            // below_v = the window's cells + the entries with a smaller value, t_v = the entries of the same value (both classes)
            for (uint32_t i0 = 0; i0 < n_ovf; i0 += BV_WAVE) {
                const uint32_t i = i0 + (uint32_t)lane;
                const bool have = i < n_ovf;
                const uint32_t me = have ? sh.ovf[i] : 0u, v = me & 0xFFFFu;
                uint32_t less = 0, same = 0;
                for (uint32_t j = 0; j < n_ovf; ++j) {
                    const uint32_t o = sh.ovf[j] & 0xFFFFu;  // (wave-uniform address: an LDS broadcast)
                    less += o < v ? 1u : 0u;
                    same += o == v ? 1u : 0u;
                }
                const unsigned long long term = (have && (me >> 16) == 0u) ? (2ull * (n1 + n2) - 2ull * (below + less) - same + 1ull) : 0ull;
                twoR += bv_wave_sum_u64(term);
            }
            rp_ph = bv_ranksum_phred(twoR, n1, n2);
        }
        if (lane == 0) {
            a.out[site].mq_ranksum = mq_ph;
            a.out[site].rpr_ranksum = rp_ph;
            atomicOr(&a.out[site].status, in_range ? BV_SITE_RANKSUM : (BV_SITE_RANKSUM | BV_SITE_RPR_RANGE));
        }
    }

    // per-group calls (caller.cpp:756-759): lrt([REF] + alts) on each group's tallies
    for (uint32_t g = 0; g < a.n_groups; ++g) {
        const uint32_t *h = S + a.hg_off + g * 512u;
        uint32_t nb = 0, gdepth[4], gtotal = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int b = r >> 1;
            const int q = ((r & 1) << 6) | lane;
            const uint32_t c = h[(b << 7) | q];
            const uint32_t cs = bv_wave_sum_u32(c);
            if (r & 1) gdepth[b] += cs; else gdepth[b] = cs;
            const bool valid = (c != 0) && (q < BV_NQ_VALID);
            const unsigned long long m = __ballot(valid);
            const uint32_t pos = nb + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (valid) {
                sh.bin_code[pos] = ((uint32_t)b << 7) | (uint32_t)q;
                sh.bin_cnt[pos] = c;
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
            B.code = sh.bin_code; B.cnt = sh.bin_cnt; B.skip_mask = 0u; B.hit = sh.tab_hit; B.miss = sh.tab_miss;
            B.loghit = a.tables->loghit; B.logmiss = a.tables->logmiss; B.ord = nullptr; B.n_ord = 0;
            if (work.n_cells != 0u && gtotal >= 2u) {
                // a shallow site: the group's cells in sample order (the site's list, filtered), for the literal replay
                const bool mine = (uint32_t)lane < work.n_cells && (sh.cells[lane] >> 16) == g;
                const unsigned long long m = __ballot(mine);
                if (mine) sh.sc.ord[__popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)(sh.cells[lane] & 0xFFFFu);
                bv_lrt_sync<0>();
                if ((uint32_t)__popcll(m) == gtotal) { B.ord = sh.sc.ord; B.n_ord = (int)gtotal; }
            } else if (gtotal >= 2u && gtotal <= (uint32_t)BV_ORD_MAX) {
                // a shallow group of a deep site: its own list (sh.cells is free: the site's list was not complete)
                const uint32_t n_list = bv_tile_sorted_cells(S + a.ord_off + (1u + g) * BV_TS_ORD_WORDS, sh.cells, lane);
                if (n_list == gtotal) {
                    if ((uint32_t)lane < n_list) sh.sc.ord[lane] = (uint16_t)(sh.cells[lane] & 0xFFFFu);
                    bv_lrt_sync<0>();
                    B.ord = sh.sc.ord; B.n_ord = (int)gtotal;
                }
            }
            B.nb = (int)nb;
            bv_lrt<0>(B, gdepth, gtotal, comb, nc, ref, a.min_af, &sh.sc.lrt, 0, lane, L);
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

// Joined-rows mode: a tile's columns go to their place in the resident [n_sites][n_samples_total] planes.
// One launch moves all the planes of a tile (blockIdx.y = plane); 8 bytes per thread when every offset, pitch
// and width of the tile allows it, else 1.
template <typename UNIT>
__global__ __launch_bounds__(256) void bv_tile_scatter_kernel(BvTileScatterArgs a) {
    const BvTileScatterPlane p = a.plane[blockIdx.y];
    const uint32_t upr = p.width_bytes / (uint32_t)sizeof(UNIT);
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (upr == 0) return;
    const uint64_t row = i / upr, u = i % upr;
    if (row >= p.n_rows) return;
    const UNIT v = *reinterpret_cast<const UNIT *>(p.src + row * p.src_pitch + u * sizeof(UNIT));
    *reinterpret_cast<UNIT *>(p.dst + row * p.dst_pitch + p.col_off + u * sizeof(UNIT)) = v;
}
void bv_launch_tile_scatter(const BvTileScatterArgs &a, hipStream_t stream) {
    // planes whose addresses, pitches and width are all multiples of 8 move 8 bytes per thread, the others 1
    BvTileScatterArgs w8, w1;
    w8.n_planes = w1.n_planes = 0;
    w8.max_rows = w1.max_rows = 0;
    uint32_t mw8 = 0, mw1 = 0;
    for (uint32_t k = 0; k < a.n_planes; ++k) {
        const BvTileScatterPlane &p = a.plane[k];
        if (p.width_bytes == 0 || p.n_rows == 0) continue;
        const bool wide = !((p.dst_pitch | p.col_off | p.src_pitch | p.width_bytes | (uint64_t)(uintptr_t)p.dst | (uint64_t)(uintptr_t)p.src) & 7u);
        BvTileScatterArgs &t = wide ? w8 : w1;
        t.plane[t.n_planes++] = p;
        if (p.n_rows > t.max_rows) t.max_rows = p.n_rows;
        uint32_t &mw = wide ? mw8 : mw1;
        if (p.width_bytes > mw) mw = p.width_bytes;
    }
    if (w8.n_planes) {
        const uint64_t total = (uint64_t)(mw8 / 8u) * w8.max_rows;
        hipLaunchKernelGGL(bv_tile_scatter_kernel<uint64_t>, dim3((uint32_t)((total + 255u) / 256u), w8.n_planes), dim3(256), 0, stream, w8);
    }
    if (w1.n_planes) {
        const uint64_t total = (uint64_t)mw1 * w1.max_rows;
        hipLaunchKernelGGL(bv_tile_scatter_kernel<uint8_t>, dim3((uint32_t)((total + 255u) / 256u), w1.n_planes), dim3(256), 0, stream, w1);
    }
}

// Many tiles, one launch: the plane descriptors come from a table in device memory, read through the constant address space
// (scalar loads), blockIdx.y = descriptor.  A job of 200-sample tiles was launch-bound at one ~8 us launch per tile
// (676 GB/s at 1 M samples, round 2).
typedef const __attribute__((address_space(4))) BvTileScatterPlane *BvTileScatterPlaneC;
template <typename UNIT>
__global__ __launch_bounds__(256) void bv_tile_scatter_many_kernel(const BvTileScatterPlane *table) {
    const BvTileScatterPlaneC p = (BvTileScatterPlaneC)(uintptr_t)table + blockIdx.y;
    const uint32_t upr = p->width_bytes / (uint32_t)sizeof(UNIT), n_rows = p->n_rows;
    if (upr == 0) return;
    const uint8_t *src = p->src;
    uint8_t *dst = p->dst + p->col_off;
    const uint64_t sp = p->src_pitch, dp = p->dst_pitch;
    // (a plane shorter than the launch's largest leaves its surplus blocks idle)
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (uint64_t)upr * n_rows; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t row = i / upr, u = i % upr;
        *reinterpret_cast<UNIT *>(dst + row * dp + u * sizeof(UNIT)) = *reinterpret_cast<const UNIT *>(src + row * sp + u * sizeof(UNIT));
    }
}
void bv_launch_tile_scatter_many(const BvTileScatterPlane *d_table, uint32_t n_wide, uint32_t n_narrow, uint64_t units_wide,
                                 uint64_t units_narrow, hipStream_t stream) {
    // the table holds the 8-byte planes first, then the byte-wise ones; grid.x covers the largest plane of its kind, capped
    // (grid-stride loop) so that a launch of many small planes is not a launch of many idle blocks
    if (n_wide) {
        uint64_t gx = (units_wide + 255u) / 256u;
        if (gx > 2048u) gx = 2048u;
        hipLaunchKernelGGL(bv_tile_scatter_many_kernel<uint64_t>, dim3((uint32_t)gx, n_wide), dim3(256), 0, stream, d_table);
    }
    if (n_narrow) {
        uint64_t gx = (units_narrow + 255u) / 256u;
        if (gx > 2048u) gx = 2048u;
        hipLaunchKernelGGL(bv_tile_scatter_many_kernel<uint8_t>, dim3((uint32_t)gx, n_narrow), dim3(256), 0, stream, d_table + n_wide);
    }
}

// The same for the usual job -- consecutive tiles of ONE width and pitch: a block writes 2 KiB of one ROW of the joined
// plane, gathered from as many tiles as that spans (the tile of a destination byte is a division by the width), so the
// writes are whole lines of the resident slab instead of 200-byte pieces a megabyte apart.
#define BV_TILE_JOIN_ROWS 8  /* rows per block: a tile's rows are adjacent in memory (its pitch is its width rounded to 16), so the
                                 reads of a block are runs of 8 x pitch bytes per tile, its writes 2 KiB runs per row */
__global__ __launch_bounds__(256) void bv_tile_join_rows_kernel(BvTileJoinArgs a) {
    // the pointer table is read through the constant address space (scalar loads); its entries are ordinary global pointers
    typedef const uint8_t *BvSrcPtr;
    typedef const __attribute__((address_space(4))) BvSrcPtr *SrcTabC;
    const SrcTabC srcs = (SrcTabC)(uintptr_t)a.srcs;
    const uint32_t row0 = blockIdx.y * BV_TILE_JOIN_ROWS;
    const uint64_t span = (uint64_t)a.n_tiles * a.width_bytes;  // bytes of one destination row covered by this launch
    for (uint64_t o = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 8u; o < span; o += (uint64_t)gridDim.x * 2048u) {
        const uint32_t t = (uint32_t)(o / a.width_bytes), w = (uint32_t)(o - (uint64_t)t * a.width_bytes);
        const uint8_t *src = srcs[t] + w;
        uint64_t v[BV_TILE_JOIN_ROWS];
#pragma unroll
        for (int r = 0; r < BV_TILE_JOIN_ROWS; ++r)
            if (row0 + r < a.n_rows) v[r] = *reinterpret_cast<const uint64_t *>(src + (uint64_t)(row0 + r) * a.src_pitch);
#pragma unroll
        for (int r = 0; r < BV_TILE_JOIN_ROWS; ++r)
            if (row0 + r < a.n_rows) *reinterpret_cast<uint64_t *>(a.dst + (uint64_t)(row0 + r) * a.dst_pitch + a.col_off + o) = v[r];
    }
}
void bv_launch_tile_join_rows(const BvTileJoinArgs &a, hipStream_t stream) {
    const uint64_t span = (uint64_t)a.n_tiles * a.width_bytes;
    uint32_t gx = (uint32_t)((span + 2047u) / 2048u);
    if (gx > 64u) gx = 64u;  // (rows supply the parallelism)
    hipLaunchKernelGGL(bv_tile_join_rows_kernel, dim3(gx, (a.n_rows + BV_TILE_JOIN_ROWS - 1) / BV_TILE_JOIN_ROWS), dim3(256), 0, stream, a);
}

// ------------------------------------------------------------------------------ packed host tiles
// bv_engine_tiles_add_sparse: a tile arrives as its covered cells only -- per site a run of (sample, call, phred, mapq, rank)
// entries, 7 bytes per covered cell -- instead of five dense planes: at the 8 % coverage of a low-pass cohort 0.6 B per cell
// cross the host link, not 5.  A block takes BV_SPARSE_ROWS consecutive sites: their entries are one contiguous run, every
// thread finds its entry's site in the block's row_start values (LDS, five steps).
#define BV_SPARSE_ROWS 32
__device__ __forceinline__ uint32_t bv_sparse_row_of(const uint32_t *rs, uint32_t e) {  // rs[0] <= e < rs[BV_SPARSE_ROWS]: largest r with rs[r] <= e
    uint32_t lo = 0, hi = BV_SPARSE_ROWS;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rs[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}
// joined rows: every entry to its cell of the resident planes (which hold "uncovered" everywhere else: bv_tile_fill_uncovered)
__global__ __launch_bounds__(256) void bv_tile_sparse_scatter_kernel(BvSparseTileArgs a) {
    __shared__ uint32_t rs[BV_SPARSE_ROWS + 1];
    const uint32_t r0 = blockIdx.x * BV_SPARSE_ROWS;
    if (threadIdx.x <= BV_SPARSE_ROWS) rs[threadIdx.x] = a.row_start[min(r0 + threadIdx.x, a.n_sites)];
    __syncthreads();
    const uint32_t e1 = min(rs[BV_SPARSE_ROWS], a.n_entries);
    for (uint32_t e = rs[0] + threadIdx.x; e < e1; e += 256u) {
        const uint32_t row = r0 + bv_sparse_row_of(rs, e), smp = a.sample[e];
        if (smp >= a.width) continue;  // (an entry outside the tile: ignored, as a cell outside a dense tile is)
        const uint32_t c = a.call[e];
        const size_t at = (size_t)row * a.pitch + a.col0 + smp;
        a.bs[at] = (uint8_t)c;
        a.q[at] = a.phred[e];
        if (a.mq) {
            a.mq[at] = a.mapq[e];
            const uint32_t r = a.rank[e];
            a.rp[at] = a.rpr_tag ? BV_RPR_TAGGED(c, r) : (uint16_t)r;
        }
    }
}
void bv_launch_tile_sparse_scatter(const BvSparseTileArgs &a, hipStream_t stream) {
    hipLaunchKernelGGL(bv_tile_sparse_scatter_kernel, dim3((a.n_sites + BV_SPARSE_ROWS - 1) / BV_SPARSE_ROWS), dim3(256), 0, stream, a);
}
__global__ __launch_bounds__(256) void bv_tile_fill_uncovered_kernel(uint4 *bs, uint4 *q, uint4 *mq, uint4 *rp, uint64_t units, uint32_t rword) {
    const uint4 n4 = make_uint4(0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u), z4 = make_uint4(0u, 0u, 0u, 0u), r4 = make_uint4(rword, rword, rword, rword);
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < units; i += (uint64_t)gridDim.x * 256u) {
        bs[i] = n4;
        q[i] = z4;
        if (mq) { mq[i] = z4; rp[2 * i] = r4; rp[2 * i + 1] = r4; }
    }
}
void bv_launch_tile_fill_uncovered(uint8_t *bs, uint8_t *q, uint8_t *mq, uint16_t *rp, uint64_t cells, uint32_t rpr_tag, hipStream_t stream) {
    const uint64_t units = cells / 16u;  // (the planes are multiples of 256 cells)
    uint64_t gx = (units + 255u) / 256u;
    if (gx > 16384u) gx = 16384u;
    hipLaunchKernelGGL(bv_tile_fill_uncovered_kernel, dim3((uint32_t)gx), dim3(256), 0, stream, reinterpret_cast<uint4 *>(bs), reinterpret_cast<uint4 *>(q),
                       reinterpret_cast<uint4 *>(mq), reinterpret_cast<uint4 *>(rp), units, rpr_tag ? 0x80008000u : 0u);
}
// per-site tallies: what bv_tile_tally_kernel adds for a covered cell, per entry
__global__ __launch_bounds__(256) void bv_tile_sparse_tally_kernel(BvSparseTileArgs a) {
    __shared__ uint32_t rs[BV_SPARSE_ROWS + 1];
    const uint32_t r0 = blockIdx.x * BV_SPARSE_ROWS;
    if (threadIdx.x <= BV_SPARSE_ROWS) rs[threadIdx.x] = a.row_start[min(r0 + threadIdx.x, a.n_sites)];
    __syncthreads();
    const uint32_t e1 = min(rs[BV_SPARSE_ROWS], a.n_entries);
    for (uint32_t e = rs[0] + threadIdx.x; e < e1; e += 256u) {
        const uint32_t site = r0 + bv_sparse_row_of(rs, e), smp = a.sample[e], c = a.call[e];
        if (smp >= a.width || c > 7u) continue;
        uint32_t *S = a.state + (size_t)site * a.stride;
        const uint32_t q = a.phred[e], b = c & 3u;
        atomicAdd(&S[BV_TS_H1 + ((c << 8) | q)], 1u);
        if (a.mapq) {
            const uint32_t mq = a.mapq[e], r = a.rank[e];
            atomicAdd(&S[BV_TS_HM + ((b << 8) | mq)], 1u);
            if (r) atomicMax(&a.maxr[site], r);
            if (r < a.rank_win) atomicAdd(&S[BV_TS_HR + b * a.rank_win + r], 1u);
            else {
                const uint32_t k = atomicAdd(&a.ovf[0], 1u);
                if (k < a.ovf_cap) { a.ovf[2u + 2u * k] = site; a.ovf[3u + 2u * k] = (b << 16) | r; }
            }
        }
        uint32_t gi = BV_NO_GROUP;
        if (a.n_groups) {
            gi = a.group_id[smp];
            if (gi < a.n_groups) atomicAdd(&S[a.hg_off + (((gi * 4u + b) << 7) | min(q, 127u))], 1u);
        }
        if (__builtin_nontemporal_load(&S[a.ord_off]) <= (uint32_t)BV_ORD_MAX) {
            const uint32_t k = atomicAdd(&S[a.ord_off], 1u);
            if (k < (uint32_t)BV_ORD_MAX) {
                S[a.ord_off + 4u + 2u * k] = (uint32_t)a.col0 + smp;
                S[a.ord_off + 5u + 2u * k] = (c << 8) | q | (gi << 16);
            }
        }
        if (gi < a.n_groups) {
            uint32_t *GL = S + a.ord_off + (1u + gi) * BV_TS_ORD_WORDS;
            if (__builtin_nontemporal_load(&GL[0]) <= (uint32_t)BV_ORD_MAX) {
                const uint32_t k = atomicAdd(&GL[0], 1u);
                if (k < (uint32_t)BV_ORD_MAX) {
                    GL[4u + 2u * k] = (uint32_t)a.col0 + smp;
                    GL[5u + 2u * k] = (c << 8) | q | (gi << 16);
                }
            }
        }
    }
}
void bv_launch_tile_sparse_tally(const BvSparseTileArgs &a, hipStream_t stream) {
    hipLaunchKernelGGL(bv_tile_sparse_tally_kernel, dim3((a.n_sites + BV_SPARSE_ROWS - 1) / BV_SPARSE_ROWS), dim3(256), 0, stream, a);
}

void bv_launch_tile_tally(const BvTileArgs &a, hipStream_t stream) {
    const uint64_t total = (uint64_t)a.n_sites * ((a.width + 15u) >> 4);
    hipLaunchKernelGGL(bv_tile_tally_kernel, dim3((uint32_t)((total + 255u) / 256u)), dim3(256), 0, stream, a);
}
void bv_launch_tile_finish(const BvTileFinishArgs &a, hipStream_t stream) {
    hipLaunchKernelGGL(bv_tile_finish_kernel, dim3(a.n_sites), dim3(BV_WAVE), 0, stream, a);
}
