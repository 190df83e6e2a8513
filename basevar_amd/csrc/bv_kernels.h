// bv_kernels.h -- launch-argument blocks shared by the kernels (bv_pass1.hip, bv_pass2.hip)
// and the engine (bv_engine.hip).
#pragma once

#include "bv_device.h"

// Device counters, each on a 128-byte line of its own (they are hit by atomics from every
// workgroup; sharing one line would serialise them at the memory-side atomic unit).
#define BV_CTR_STRIDE 32u                  /* words */
// The first BV_CTR_PER_LAUNCH lines are zeroed before every launch; the error counters behind them are STICKY:
// they accumulate over submits and are cleared only by bv_engine_wait, so that a time-out or a zero-frequency
// site of an earlier batch of a pipelined sequence of submits is still reported.
#define BV_CTR_VARIANTS (0u * BV_CTR_STRIDE)  /* number of BV_SITE_VARIANT sites = length of var_list */
#define BV_CTR_TICKET (1u * BV_CTR_STRIDE)    /* pass-1 site ticket counter                           */
#define BV_CTR_CANDS (2u * BV_CTR_STRIDE)     /* short rows: candidates for the wave solver = length of cand_list */
#define BV_CTR_EASY (3u * BV_CTR_STRIDE)      /* short rows: candidates for the 16-lane solver with <= 2 active bases = length of easy_list */
#define BV_CTR_EASY3 (4u * BV_CTR_STRIDE)     /* short rows: the same with >= 3 active bases = length of easy3_list */
// short rows: job tickets of the two kernels of the 16-lane solver, BV_TICKET_SLICES lines each -- workgroup w draws from
// slice w % BV_TICKET_SLICES, which owns every BV_TICKET_SLICES-th job (one address serves ~88 M atomics/s; the slices are
// on lines of their own)
#define BV_TICKET_SLICES 8u
#define BV_CTR_TICKET_A (5u * BV_CTR_STRIDE)
#define BV_CTR_TICKET_B ((5u + BV_TICKET_SLICES) * BV_CTR_STRIDE)
#define BV_CTR_PER_LAUNCH (5u + 2u * BV_TICKET_SLICES) /* lines zeroed per launch                     */
#define BV_CTR_ZEROFREQ (BV_CTR_PER_LAUNCH * BV_CTR_STRIDE)         /* sticky: sites with BV_SITE_ZERO_FREQ       */
#define BV_CTR_TIMEOUT ((BV_CTR_PER_LAUNCH + 1u) * BV_CTR_STRIDE)   /* sticky: pass-1 pipeline time-out flag      */
#define BV_CTR_WORDS ((BV_CTR_PER_LAUNCH + 2u) * BV_CTR_STRIDE)
// what a bounded wait adds to BV_CTR_TIMEOUT when it gives up: which hand-off it was shows in the error text of bv_engine_wait
#define BV_TMO_PUSH 0x1u              /* fused kernel: a streaming wave, its candidate queue full            */
#define BV_TMO_PUSH_VARIANT 0x100u    /* fused kernel: a solver wave, the variant queue full                 */
#define BV_TMO_TAKE 0x10000u          /* fused kernel: a solver wave, a claimed candidate entry never written */
#define BV_TMO_TAKE_VARIANT 0x1000000u /* fused kernel: a streaming wave, a claimed variant entry never written */
#define BV_TMO_RING 0x10000000u       /* long-row kernel: a ring flag (published / filled / drained / team)  */

// A queue of slabs (same row length, no pop-groups) solved by ONE launch of each pass -- bv_engine_submit_many: the
// persistent grid of pass 1 draws its site tickets across the whole queue, so the solve of the last deep sites of one
// slab runs under the stream of the next.  Global site t of segment k is first[k] + its local index; every pointer is
// biased by -first[k] rows / records on the host, so the kernels index it with t directly.
#define BV_MAX_CHAIN 16
struct BvChain {
    uint32_t n;
    uint32_t first[BV_MAX_CHAIN];
    const uint8_t *bs[BV_MAX_CHAIN], *q[BV_MAX_CHAIN], *mapq[BV_MAX_CHAIN], *ref_base[BV_MAX_CHAIN];
    const uint16_t *rpr[BV_MAX_CHAIN];
    bv_site_result *out[BV_MAX_CHAIN];
    bv_group_result *gout[BV_MAX_CHAIN];  // pop-group records, biased by -first x n_groups records; NULL without groups
};
#if defined(__HIPCC__)
// The table is written by the host before the launch and never by a kernel: read it through the constant address space, so
// that wave-uniform look-ups are SCALAR loads (lgkmcnt) -- a vector load would sit in the vmcnt queue of the streaming
// kernels and drain their LDS-DMA rings at every row.
typedef const __attribute__((address_space(4))) BvChain *BvChainC;
__device__ __forceinline__ BvChainC bv_chain_const(const BvChain *p) { return (BvChainC)(uintptr_t)p; }
__device__ __forceinline__ uint32_t bv_chain_seg(BvChainC c, uint32_t site) {
    uint32_t k = 0;
    const uint32_t n = c->n;
#pragma unroll
    for (int i = 1; i < BV_MAX_CHAIN; ++i) k += ((uint32_t)i < n && site >= c->first[i]) ? 1u : 0u;
    return k;
}
#endif

struct BvPass1Args {
    const uint8_t *bs;        // [n_sites][pitch]
    const uint8_t *q;         // [n_sites][pitch]
    const uint8_t *ref_base;  // [n_sites]
    uint64_t pitch;
    uint32_t n_sites;
    uint32_t n_samples;
    uint32_t flags;           // BV_FLAG_*
    double min_af;
    const BvTables *tables;
    bv_site_result *out;      // [n_sites]
    uint32_t *var_list;       // [n_sites]  indices of BV_SITE_VARIANT sites (unordered)
    uint32_t *counters;       // BV_CTR_* words; VARIANTS and TICKET zeroed before the launch
    uint32_t n_cu;            // compute units of the device (hipDeviceProp_t::multiProcessorCount): sizes persistent grids
    const BvChain *ch;        // device memory, or NULL: not chained.  Long rows only (bv_pass1_kernel)
};

struct BvPass2Args {
    const uint8_t *bs;
    const uint8_t *q;         // needed only when n_groups > 0
    const uint8_t *mapq;      // may be NULL together with rpr: rank sums skipped
    const uint16_t *rpr;
    const uint8_t *ref_base;
    const uint8_t *group_id;  // [n_samples] or NULL
    uint64_t pitch;
    uint32_t n_sites;
    uint32_t n_samples;
    uint32_t n_groups;
    double min_af;
    const BvTables *tables;
    bv_site_result *out;
    bv_group_result *gout;    // [n_sites][n_groups] or NULL
    const uint32_t *var_list;
    const uint32_t *counters;
    uint32_t n_cu;
    uint32_t flags;           // BV_FLAG_*
    uint32_t *gitems;         // pop-group calls handed from the tally kernels to the group solve kernels: item v * n_groups + g
    uint32_t gitem_cap;       // (v = position in var_list) = BV_P2G_ITEM_WORDS words; items >= gitem_cap, or gitems == NULL:
                              // the tally kernel solves the group itself (one wave per group)
    const uint8_t *gidp;      // group_id prepared for bv_p2g_stream_kernel: g << 2, or 0x80 for "no group" (bv_launch_gid_prepare)
    const BvChain *ch;        // device memory, or NULL: a chained launch -- planes (and gout) come per segment, biased
    uint32_t ch_cat;          // chained short rows: ref_base / out are the engine's CONTIGUOUS copies, indexed with the global site
                              // number as they are; else (long rows) they come per segment too
    uint32_t rpr_tag;         // 1: the rpr plane is in the tagged layout (BV_SLAB_RPR_TAGGED): rank | base << 13 | nocall << 15
};
// item: [0] = number of bins | state, [1..4] = the group's ACGT depths, [5] = bases with a phred-0 call, [8..] = its bins
// (valid phreds only, (base, phred) order): code << 16 | count for BV_P2G_PENDING items, code << 23 | count for BV_P2G_HARD ones
#define BV_P2G_ITEM_WORDS (8u + BV_SLOTS * BV_WAVE)
// the 16-lane solver (bv_solver16.h) keeps a site's or a group's bins in the registers of its 16 lanes
#define BV_G16_SLOTS 8                      /* bins per lane: 8 x 16 = 128 bins per site */
#define BV_G16_MAX_BINS (16 * BV_G16_SLOTS)
// LDS scratch of one group: the EM's previous marginals [BV_G16_SLOTS][16] doubles (256 words; later the rank sum's counts and
// the staged record), padded so that the four groups of a wave fall on different halves of the LDS banks
#define BV_G16_GRP_WORDS 288
#define BV_P2G_PENDING 0x80000000u  /* for bv_p2g_solve16_kernel: four items per wave */
#define BV_P2G_HARD 0x40000000u     /* for bv_p2g_hard_kernel: one wave per item (shallow group, phred-0 calls, > 128 bins, min_af <= 0) */
#define BV_P2G_SHALLOW 0x20000000u  /* ... and its EMs replay the reference's per-sample order */
// Small groups (bv_p2g_solve_small_kernel): a PENDING item with one of these bits is solved on 4 / 8 lanes, sixteen / eight items
// per wave (four register slots per lane, eight when an item of the job needs them).  The workgroup-per-row tally kernel sets
// them: 4 lanes where three quarters of the SITE's pending groups have at most 16 bins (and this one at most 32), else 8 lanes
// where this one has at most 64.  A wave's consecutive items belong to one site and are mostly of one kind, and what solves an
// item does not depend on the rest of the launch.
#define BV_P2G_L4 0x10000000u
#define BV_P2G_L8 0x08000000u

// short rows (bv_pass1_short.hip): pass 1 as a streaming kernel + a solve kernel that meet in HBM scratch
#define BV_SHORT_ROW_MAX 49152u /* measured crossover with the long-row kernel (one row shared by several tally waves) */
#define BV_S_BIN_STRIDE 512u /* words per site in the bins scratch: every non-empty (base, phred < 128) bin */
struct __attribute__((aligned(16))) BvSiteSummary {  // 48 bytes per site
    uint32_t fwd[4], rev[4];  // covered calls per base, forward / reverse strand
    uint32_t nb;              // exported bins (candidate sites only)
    uint32_t flags;           // BV_SUM_*
    uint32_t pad_[2];
};
#define BV_SUM_Q0_MASK 0xFu /* bit b: base b has a phred-0 call           */
#define BV_SUM_BADQ 0x10u   /* a covered cell had phred > 93               */
#define BV_SUM_CAND 0x20u   /* a candidate: solved from its exported bins  */
struct BvP1ShortArgs {
    const uint8_t *bs, *q, *ref_base;
    uint64_t pitch;
    uint32_t n_sites, n_samples, flags, n_cu;
    double min_af;
    const BvTables *tables;
    bv_site_result *out;
    uint32_t *var_list;
    uint32_t *counters;
    BvSiteSummary *summ;   // [n_sites]
    uint32_t *bins;        // [n_sites][BV_S_BIN_STRIDE]  (code << 16 | count), candidate sites only
    uint32_t *cand_list;   // [n_sites]  candidates that take a whole wave (shallow, phred-0 calls, > 128 bins, min_af <= 0)
    uint32_t *easy_list;   // [n_sites]  candidates solved four per wave (bv_solver16.h), at most two active bases
    uint32_t *easy3_list;  // [n_sites]  the same with three or four active bases: several times the EM runs, so they are kept
                           //            apart -- the four sites of a wave run in lockstep and pay for the slowest
    const BvChain *ch;     // device memory, or NULL: a chained launch -- the call / phred planes come per segment (biased), while
                           // ref_base / out are the engine's contiguous copies indexed with the global site number
    // bv_pass1_fused.hip only: the rank planes, when the kernel is to stream the variant sites' pass-2 rows too (both NULL: pass 2
    // is launched after it as usual)
    const uint8_t *mapq;
    const uint16_t *rpr;
    uint32_t *ovf;         // bv_pass1_fused.hip: [n_sites][4] the workgroups' overflow lists of variant sites (site, class table, depths, lut)
    uint32_t rpr_tag;      // 1: the rpr plane is in the tagged layout (BV_SLAB_RPR_TAGGED); the pass-2 rows then stream mapq + ranks only
};
// chained short-row launches: reference bases of all segments -> one array; records of all segments <- one array
void bv_launch_chain_gather_ref(const BvChain *ch, uint32_t n_sites, uint8_t *ref_cat, hipStream_t stream);
void bv_launch_chain_scatter_out(const BvChain *ch, uint32_t n_sites, const bv_site_result *out_cat, hipStream_t stream);
void bv_launch_p1s_stream(const BvP1ShortArgs &a, hipStream_t stream);
void bv_launch_p1s_solve(const BvP1ShortArgs &a, hipStream_t stream, bool beside_stream = false);

// sample-axis tile mode (bv_tiles.hip)
struct BvTileArgs {
    const uint8_t *bs;        // [n_sites][pitch] tile planes (device)
    const uint8_t *q;
    const uint8_t *mapq;      // may be NULL together with rpr
    const uint16_t *rpr;
    const uint8_t *group_id;  // [width] group of each sample of THIS tile, or NULL
    uint64_t pitch;
    uint32_t n_sites;
    uint32_t width;           // samples in this tile
    uint32_t n_groups;
    uint32_t stride;          // state words per site
    uint32_t rank_win;        // read-position ranks 0 .. rank_win - 1 are tallied (multiple of 1024)
    uint32_t hg_off;          // word offset of the pop-group tallies in a site's state
    uint32_t *state;          // [n_sites][stride]
    uint32_t *maxr;           // [n_sites] largest read-position rank seen
    uint32_t ord_off;         // word offset of the site's covered-cell list (BV_TS_ORD_WORDS words) in its state
    uint32_t col0;            // index of the tile's first sample in the job (the list is put in sample order by it)
    uint32_t *ovf;            // pool of read-position ranks >= rank_win: [0] entries appended, then (site, base << 16 | rank) pairs
    uint32_t ovf_cap;         // entries the pool holds
    uint32_t rpr_tag;         // 1: the tile's rpr plane is in the tagged layout (BV_SLAB_RPR_TAGGED)
};
// Per-site list of covered cells (per-site-tally realisation): word 0 = cells appended, then BV_ORD_MAX x (sample index,
// call << 8 | phred | group << 16).  Complete -- and used, sorted by sample index, for the reference's per-sample replay --
// exactly when the site has at most BV_ORD_MAX covered cells.
#define BV_TS_ORD_WORDS (4u + 2u * 64u)
#define BV_TS_OVF_LIST 2048u /* ranks beyond the window a SITE can have and still get an exact ReadPosRankSum */
struct BvTileFinishArgs {
    const uint32_t *state;
    const uint32_t *maxr;
    const uint8_t *ref_base;
    uint32_t n_sites;
    uint32_t n_groups;
    uint32_t stride;
    uint32_t have_ranks;
    uint32_t rank_win, hg_off;
    uint32_t ord_off;         // see BvTileArgs
    const uint32_t *ovf;
    uint32_t ovf_cap;
    double min_af;
    const BvTables *tables;
    bv_site_result *out;
    bv_group_result *gout;
    uint32_t *var_list;
    uint32_t *counters;
};
// packed host tiles (bv_engine_tiles_add_sparse): the covered cells of a tile only, row after row
struct BvSparseTileArgs {
    const uint32_t *row_start;  // [n_sites + 1] entries of site s: [row_start[s], row_start[s + 1])
    const uint16_t *sample;     // [n_entries] sample index inside the tile
    const uint8_t *call, *phred, *mapq;  // [n_entries]; mapq may be NULL together with rank
    const uint16_t *rank;       // [n_entries] plain read-position ranks
    uint32_t n_sites, width, n_entries;
    uint32_t rpr_tag;           // joined rows: write the rank words tagged (BV_SLAB_RPR_TAGGED)
    // joined-rows realisation: the resident planes and the tile's first column
    uint8_t *bs, *q, *mq;
    uint16_t *rp;
    uint64_t pitch, col0;
    // per-site-tally realisation: as BvTileArgs
    const uint8_t *group_id;    // [width] or NULL
    uint32_t n_groups, stride, rank_win, hg_off, ord_off, ovf_cap;
    uint32_t *state, *maxr, *ovf;
};
void bv_launch_tile_sparse_scatter(const BvSparseTileArgs &a, hipStream_t stream);
void bv_launch_tile_sparse_tally(const BvSparseTileArgs &a, hipStream_t stream);
// the whole joined planes as "nobody covered": calls 'N', phred / mapq 0, ranks 0 (tagged: 0x8000)
void bv_launch_tile_fill_uncovered(uint8_t *bs, uint8_t *q, uint8_t *mq, uint16_t *rp, uint64_t cells, uint32_t rpr_tag, hipStream_t stream);
void bv_launch_tile_tally(const BvTileArgs &a, hipStream_t stream);
void bv_launch_tile_finish(const BvTileFinishArgs &a, hipStream_t stream);
// joined-rows tile mode: `width_bytes` bytes of each of n_rows rows go from src (row pitch src_pitch) to
// dst + col_off (row pitch dst_pitch), for up to five planes in one launch
struct BvTileScatterPlane {
    uint8_t *dst;
    const uint8_t *src;
    uint64_t dst_pitch, src_pitch, col_off;
    uint32_t width_bytes, n_rows;
};
struct BvTileScatterArgs {
    BvTileScatterPlane plane[5];
    uint32_t n_planes, max_rows;
};
void bv_launch_tile_scatter(const BvTileScatterArgs &a, hipStream_t stream);
// the planes of MANY tiles in one launch (two: 8-byte and byte-wise planes), descriptors in device memory: `n_wide` planes that
// move 8 bytes per thread first, then `n_narrow`; units_* = units of the largest plane of each kind
#define BV_TILE_MANY_MAX 256  /* tiles per launch of bv_engine_tiles_add_many */
// consecutive tiles of one width and pitch, one plane kind: row `r` of tile t goes to dst + r * dst_pitch + col_off + t * width_bytes
struct BvTileJoinArgs {
    uint8_t *dst;
    const uint8_t *const *srcs;  // [n_tiles] device table: this plane of every tile
    uint64_t dst_pitch, src_pitch, col_off;
    uint32_t width_bytes, n_tiles, n_rows;  // all byte quantities multiples of 8
};
void bv_launch_tile_join_rows(const BvTileJoinArgs &a, hipStream_t stream);
void bv_launch_tile_scatter_many(const BvTileScatterPlane *d_table, uint32_t n_wide, uint32_t n_narrow, uint64_t units_wide,
                                 uint64_t units_narrow, hipStream_t stream);

// host-callable launchers (defined next to the kernels)
void bv_launch_pass1(const BvPass1Args &a, hipStream_t stream);
void bv_launch_pass2(const BvPass2Args &a, hipStream_t stream);
void bv_launch_p2g_solve16(const BvPass2Args &a, hipStream_t stream);
void bv_launch_p2g_stream(const BvPass2Args &a, hipStream_t stream);
bool bv_p2g_all_items(const BvPass2Args &a);
bool bv_p2g_streams(const BvPass2Args &a);  // whether bv_launch_pass2 takes the LDS-DMA group tally (needs a.gidp)
void bv_launch_gid_prepare(const uint8_t *gid, uint8_t *gidp, uint32_t n_bytes, uint32_t n_groups, hipStream_t stream);
// pop-groups [lo, lo + n) as 0 .. n - 1, every other sample BV_NO_GROUP: the group plane of one round of pass 2 (bv_engine.hip)
#define BV_GROUPS_PER_ROUND 32u
void bv_launch_gid_round(const uint8_t *gid, uint8_t *out, uint32_t n_bytes, uint32_t lo, uint32_t n, hipStream_t stream);
size_t bv_pass2_lds_bytes(uint32_t n_groups);
// short rows of at least three 4 KiB slots (4,097 .. 49,152 samples), chained launches included: pass 1 -- and, with the rank
// planes in a.mapq / a.rpr, the variant sites' pass-2 rows -- as ONE persistent kernel (bv_pass1_fused.hip)
bool bv_p1s_fused_takes(const BvP1ShortArgs &a);
void bv_launch_p1s_fused(const BvP1ShortArgs &a, hipStream_t stream);
