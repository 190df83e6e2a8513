/*
 * basevar_amd.h -- C ABI of the MI355X-native per-site basetype likelihood engine.
 *
 * This is the drop-in boundary for ONE path of ShujiaHuang/basevar: the per-site
 * caller (reference: src/basetype.cpp:22-295, src/algorithm.h:44-255,
 * htslib/kfunc.c:39-143,245-313) as it is driven from _basevar_caller
 * (src/basetype_caller.cpp:667-765) and the arithmetic of _out_vcf_line /
 * _out_cvg_line (src/basetype_caller.cpp:1103-1260).
 *
 * The reference has no plugin API; its seam is the C++ class `BaseType`
 * (src/basetype.h:64-153) plus the free functions `strand_bias` and
 * `ref_vs_alt_ranksumtest` (src/basetype.h:168-181), all called once per site.
 * This ABI is the batched equivalent: S sites x N samples per submit.
 * Every entry point below names the reference interface it replaces.
 *
 * Plain C: pointers and sizes only; no C++/torch types cross this boundary.
 * All functions return 0 (BV_OK) on success and a negative bv_status on error;
 * bv_last_error() gives the message (C++ wrapper rethrows std::runtime_error,
 * the reference's error type, src/basetype.cpp:54-56,113-115,272).
 */
#ifndef BASEVAR_AMD_H
#define BASEVAR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BV_ABI_VERSION 1

/* ---- cell encoding of the `base_strand` plane ---------------------------------
 * One byte per (site, sample): the first character of the reference's per-sample token
 * (src/basetype.cpp:50; domain per src/basetype_caller.cpp:1060-1077) plus the strand
 * (src/basetype.cpp:257-264).
 *   bits 0-1: base call  0 'A'  1 'C'  2 'G'  3 'T'   (index in BASES, src/basetype.h:19)
 *   bit 2   : 1 = reverse strand '-', 0 = forward '+'
 *   bit 3   : 1 = not a base call; then bits 0-1 select  0 'N' (uncovered, strand '.')
 *             1 '+' (insertion token)  2 '-' (deletion token); bit 2 is ignored
 *   bits 4-7: must be zero.
 * A covered base call is therefore a value 0..7 that directly indexes the kernel's
 * (strand, base) histogram row; every other value is skipped with one bit test.
 */
#define BV_CELL_BASE_MASK 0x03u
#define BV_CELL_REV 0x04u
#define BV_CELL_NOCALL 0x08u
#define BV_CELL_N 0x08u
#define BV_CELL_INS 0x09u
#define BV_CELL_DEL 0x0Au
/* codes used by bv_slab.ref_base[] and bv_site_result.alt[] */
#define BV_BASE_A 0u
#define BV_BASE_C 1u
#define BV_BASE_G 2u
#define BV_BASE_T 3u
#define BV_BASE_OTHER 4u

/* qual plane: phred value = (reference quality char) - 33, src/basetype.cpp:47.
 * Valid domain 0..93 (chars '!'..'~').  Larger values set BV_SITE_BAD_QUAL. */
#define BV_MAX_PHRED 93u
#define BV_MAX_ALT 4 /* ref not in ACGT + four active bases (basetype.cpp:172-177) */
#define BV_NO_GROUP 0xFFu
#define BV_MAX_GROUPS 255u /* group ids are bytes, 0xFF = none.  The reference takes any number (a std::map,
                              src/basetype_caller.cpp:372-410); beyond 32 the engine runs pass 2 once per 32 groups */

typedef enum bv_status {
    BV_OK = 0,
    BV_ERR_INVALID_ARG = -1,
    BV_ERR_NO_DEVICE = -2,   /* no HIP device / kernel image not loadable: fail loudly */
    BV_ERR_HIP = -3,         /* a HIP runtime call failed; see bv_last_error()       */
    BV_ERR_TOO_LARGE = -4,   /* slab exceeds cfg.max_sites / max_samples             */
    BV_ERR_SITE = -5         /* at least one site raised a reference-style exception */
} bv_status;

typedef enum bv_mem_kind {
    BV_MEM_DEVICE = 0, /* planes and outputs are device (HBM) pointers of cfg.device */
    BV_MEM_HOST = 1    /* planes and outputs are host pointers; engine stages them   */
} bv_mem_kind;

/* per-site status bits (bv_site_result.status) */
#define BV_SITE_COVERED 0x1u   /* total_depth > 0  (CVG row emitted, caller.cpp:1246)  */
#define BV_SITE_VARIANT 0x2u   /* alt set non-empty (VCF row emitted, caller.cpp:745)  */
#define BV_SITE_BAD_QUAL 0x4u  /* a covered cell had phred > 93                        */
#define BV_SITE_ZERO_FREQ 0x8u /* reference would throw at basetype.cpp:113-115        */
#define BV_SITE_RANKSUM 0x10u  /* mapq/rpr rank sums were computed (planes present)    */
#define BV_SITE_SOR_OVERFLOW 0x20u /* int product in SOR exceeded 2^31 (basetype.cpp:286 is UB there) */
#define BV_SITE_RPR_RANGE 0x40u    /* tile mode only: a read-position rank >= 1024 was seen; rpr_ranksum = NaN */
#define BV_SITE_LOG_APPROX 0x80u   /* a site of <= 64 covered samples was replayed in the reference's per-sample order, but with the
                                      device library's log() instead of the host libm's (bv_engine_host_log_exact() == 0: the host's
                                      libm is not the glibc the restatement knows): where two allele subsets tie to the last bit the
                                      pick may differ from the reference's; every value is still within 1e-6 */

/* Input: SoA planes [n_sites][pitch], one row per genomic site, one cell per sample.
 * Replaces `struct BatchInfo` (src/basetype.h:25-43) for a whole batch of sites. */
typedef struct bv_slab {
    uint32_t n_sites;
    uint32_t n_samples;
    uint64_t pitch;              /* cells per row, >= n_samples, multiple of 16       */
    const uint8_t *base_strand;  /* [n_sites][pitch]  align_bases[i][0] + map_strands */
    const uint8_t *qual;         /* [n_sites][pitch]  align_base_quals - 33           */
    const uint8_t *mapq;         /* [n_sites][pitch]  mapqs; may be NULL              */
    const uint16_t *rpr;         /* [n_sites][pitch]  base_pos_ranks; may be NULL     */
    const uint8_t *ref_base;     /* [n_sites] toupper(ref_base[0]) as BV_BASE_*; 4 = not ACGT  */
    const uint8_t *group_id;     /* [n_samples] pop-group index or BV_NO_GROUP; may be NULL; any alignment (the engine
                                    keeps its own padded copy)                                */
    uint32_t n_groups;           /* 0 if no groups (caller.cpp:746)                   */
    uint32_t mem_kind;           /* bv_mem_kind                                       */
} bv_slab;

/* Output: one fixed-size record per site (208 bytes).
 * Replaces the BaseType getters (src/basetype.h:121-151), StrandBiasInfo
 * (src/basetype.h:57-62) and the INFO arithmetic of _out_vcf_line
 * (src/basetype_caller.cpp:1113-1164). */
typedef struct bv_site_result {
    uint32_t depth[4];    /* get_base_depth('A','C','G','T'), basetype.cpp:58 (bit-exact) */
    uint32_t total_depth; /* get_total_depth(), basetype.cpp:59                          */
    uint32_t status;      /* BV_SITE_* bits                                              */
    uint32_t cvg_sb[4];   /* ref_fwd, ref_rev, alt_fwd, alt_rev; alt = all non-ref ACGT  */
                          /* (_out_cvg_line, caller.cpp:1236-1245)                       */
    double cvg_fs;        /* StrandBiasInfo.fs  of that call, basetype.cpp:277-282       */
    double cvg_sor;       /* StrandBiasInfo.sor of that call, basetype.cpp:286           */
    uint8_t n_alt;        /* get_alt_bases().size(), basetype.cpp:172-177                */
    uint8_t alt[BV_MAX_ALT]; /* alt base codes in reference order                        */
    uint8_t n_em;         /* diagnostic: number of EM runs (<= 10)                       */
    uint16_t em_iters;    /* diagnostic: EM loop iterations summed over all EM runs      */
    double af[BV_MAX_ALT];  /* get_lrt_af(alt[i]), basetype.cpp:175 (CM_AF)              */
    double caf[BV_MAX_ALT]; /* depth[alt]/total_depth, caller.cpp:1122 (CM_CAF)          */
    double qual;          /* get_var_qual(), basetype.cpp:180-196                        */
    double chi2;          /* last chi_sqrt_value of lrt(), basetype.cpp:160              */
    double qd;            /* qual / sum depth[alt], caller.cpp:1160-1161                 */
    uint32_t var_sb[4];   /* strand_bias w.r.t. chosen ALTs, caller.cpp:1164             */
    double var_fs;
    double var_sor;
    double mq_ranksum;    /* ref_vs_alt_ranksumtest(mapqs), caller.cpp:1151 (caller truncates to int) */
    double rpr_ranksum;   /* ... (base_pos_ranks), caller.cpp:1154                       */
    double bq_ranksum;    /* ... (align_base_quals), caller.cpp:1157                     */
} bv_site_result;

/* Per (site, group) record (48 bytes), valid for BV_SITE_VARIANT sites only.
 * Replaces __gb()/lrt([REF]+alts) (src/basetype_caller.cpp:756-777) and the
 * "<group>_AF=" INFO values (caller.cpp:1182-1196). */
typedef struct bv_group_result {
    uint8_t n_alt;
    uint8_t alt[BV_MAX_ALT];
    uint8_t reserved[3];
    uint32_t total_depth; /* ACGT depth of the group's samples                           */
    uint32_t reserved2;
    double af[BV_MAX_ALT];
} bv_group_result;

/* diagnostic: pass 1 stops after the tally (depth[] / total_depth only are valid); used by
 * bench.py --ablate to time the HBM streaming part of pass 1 without the solver */
#define BV_FLAG_TALLY_ONLY 0x1u
#define BV_FLAG_SKIP_FISHER 0x2u /* diagnostic: strand-bias Fisher tests return p = 1 */
#define BV_FLAG_SKIP_LRT 0x4u    /* diagnostic: no EM / LRT (no site is called variant) */
#define BV_FLAG_GRID_LIMIT(n) (((uint32_t)(n) & 0xFFu) << 16) /* diagnostic / tests: at most n workgroups per persistent short-row kernel,
                                     so that a small input walks the many-sites-per-wave paths (list flushes, 64-site blocks) */
#define BV_FLAG_GROUP_INLINE 0x40u /* diagnostic / tests: pop-group calls are solved inside the pass-2 tally kernel, one wave per group
                                     (the path taken when the item scratch cannot hold every (variant site, group)) */
#define BV_FLAG_PASS2_SWEEP 0x20u /* diagnostic: short rows take the plain-load pass-2 kernels (not the LDS-DMA one) */
#define BV_FLAG_WAVE_SOLVER 0x10u /* diagnostic: short-row candidates and pop-group calls all take the one-per-wave solver (none the 16-lane one) */
#define BV_FLAG_SPLIT(n) (((uint32_t)(n) & 0xFu) << 24) /* tuning / tests: short-row batches (<= 49,152 samples) run as a software
                                     pipeline of n chunks of consecutive sites over two streams (solve kernels of chunk c under the
                                     streaming kernel of chunk c + 1); 0 = the engine's default (by batch size), 1 = no pipeline.
                                     Records do not depend on n. */
#define BV_FLAG_SHORT_ROW_FORM(n) (((uint32_t)(n) & 0xFu) << 12) /* diagnostic / A-B runs: which kernels rows of 4,097 .. 49,152 samples take.
                                     0 = the engine's default: ONE persistent kernel for pass 1 and the variant sites' rank-sum rows
                                     (streaming and solver waves side by side; csrc/bv_pass1_fused.hip); 10 = that kernel for pass 1, pass 2 a
                                     launch of its own; 9 = round 3's three launches (streaming kernel, solve kernel, pass-2 kernel).
                                     Records do not depend on it. */
#define BV_FLAG_HOST_ORDERED 0x80u /* BV_MEM_HOST planes: the engine's copy stream waits for everything queued on the caller's `stream`
                                     before it reads them (for callers that fill their pinned planes with asynchronous work on that
                                     stream).  Default (flag clear): host planes must be COMPLETE in host memory when bv_engine_submit /
                                     bv_engine_tiles_add / bv_engine_tiles_finish is called and stay untouched until bv_engine_wait --
                                     the copies run on streams of the engine's own, ahead of `stream`, under earlier kernels */
#define BV_FLAG_LANES 0x10000000u /* two lanes: device-resident submits (bv_engine_submit, BV_MEM_DEVICE) alternate between two internal
                                     streams with a scratch set each, so that the kernels of consecutive submits overlap -- on short rows the
                                     solve kernels of one batch (issue-bound, no HBM traffic) run under the streaming kernels of the next.
                                     The `stream` argument then only ORDERS the submit behind the caller's earlier work on that stream; the
                                     records are complete after bv_engine_wait(), or on a stream that called bv_engine_join() after the
                                     submit.  The slabs and record buffers of submits in flight must be distinct.  Records do not depend
                                     on the flag. */
#define BV_FLAG_SPARSE_TIMING 0x20000000u /* the per-pass timing events (bv_engine_timing_get, bv_engine_kernel_ms) are recorded for one
                                             launch in eight only: four event records cost ~15 us per launch, which a host that queues
                                             small batches back to back notices (8,192 sites x 10 k samples: +11 % sites/s).
                                             bv_engine_kernel_ms then reports the last TIMED launch.  Records do not depend on the flag. */
#define BV_FLAG_TILE_STATE 0x8u  /* tile mode: always accumulate per-site tallies (the fallback for jobs whose
                                    joined planes do not fit the HBM) instead of joining the tiles into rows */

typedef struct bv_engine_config {
    int32_t device;        /* HIP device ordinal                                        */
    uint32_t max_sites;    /* largest n_sites per submit (sizes scratch)                */
    uint32_t max_samples;  /* largest n_samples: sizes the host staging area (0 = no staging) and the
                              log-factorial table of the strand-bias test (0 = 1 Mi entries)   */
    uint32_t flags;        /* BV_FLAG_* bits, normally 0                                */
    double min_af;         /* BaseType ctor arg 2 (basetype.cpp:30): already the        */
                           /* float-rounded value of caller.cpp:122; see bv_min_af()    */
} bv_engine_config;

typedef struct bv_engine bv_engine;

/* Library / ABI version string, e.g. "basevar_amd 0.1 abi1 gfx950". */
const char *bv_version(void);

/* min_af exactly as the reference derives it: (double)std::min(float(100)/n_files,
 * user_min_af) -- src/basetype_caller.cpp:122, src/basetype_utils.h:80,94. */
double bv_min_af(uint32_t n_samples, float user_min_af);

/* Lifetime.  One engine == one HIP stream + scratch on cfg.device.  Thread-compatible:
 * use one engine per host thread/GPU (mirrors one BaseType per ThreadPool worker,
 * src/basetype_caller.cpp:485-510). */
int bv_engine_create(const bv_engine_config *cfg, bv_engine **out);
int bv_engine_destroy(bv_engine *e);

/* Asynchronously run the whole per-site path over a slab:
 *   BaseType(bi, min_af) + lrt()                  caller.cpp:742-743
 *   strand_bias (CVG and VCF flavours)            caller.cpp:1245, 1164
 *   3 x ref_vs_alt_ranksumtest                    caller.cpp:1151-1157
 *   per-group __gb() when slab->n_groups > 0      caller.cpp:756-759
 * BV_MEM_HOST slabs: the planes must be fully written when the call is made and must not change before bv_engine_wait
 *           returns (they are copied by the engine's own copy streams, not in `stream` order; BV_FLAG_HOST_ORDERED changes that).
 * `out`   : [n_sites] records, same mem_kind as the slab; a device buffer must be 16-byte aligned.
 * `gout`  : [n_sites][n_groups] records or NULL when n_groups == 0.
 * `stream`: hipStream_t to launch on, or NULL for the engine's own stream.  (NULL is also the handle of
 *           the legacy default stream: a caller that wants ordering with other work must pass an
 *           explicit stream.)
 * No allocation crosses the ABI; caller owns slab and outputs. */
int bv_engine_submit(bv_engine *e, const bv_slab *slab, bv_site_result *out,
                     bv_group_result *gout, void *stream);

/* n_slabs device-resident slabs as ONE launch per pass: equivalent to n_slabs calls of bv_engine_submit, but the persistent
 * kernels draw their sites across the whole queue, so the fixed tail of a launch -- the solve of its last deep sites, ~0.1 ms
 * during which the chip has nothing left to stream -- is paid once per queue, not once per slab.  Meant for hosts that hold
 * several small batches (a few thousand sites each): 8,192-site batches run at the rate of one large batch (100 k samples:
 * 14.9 -> 20 M sites/s; 10 k samples: 52 -> 143 M).  Chained, 16 slabs per launch, when the slabs are BV_MEM_DEVICE, share
 * n_samples, pitch, the presence of rank planes and the pop-group assignment (the same group_id array and n_groups), and
 * have at most cfg.max_sites sites together; anything else is submitted slab by slab.  bv_engine_last_variant_count then
 * counts the last launch.  Every record is byte-identical to the one a submit of its own slab writes.
 * bv_engine_submit_many takes slabs without pop-groups; bv_engine_submit_many_g also takes gouts[k] = slab k's
 * [n_sites][n_groups] records (NULL entries for slabs without groups).
 * Replaces nothing in the reference (its workers take one position at a time, basetype_caller.cpp:738-762). */
int bv_engine_submit_many(bv_engine *e, uint32_t n_slabs, const bv_slab *slabs, bv_site_result *const *outs, void *stream);
int bv_engine_submit_many_g(bv_engine *e, uint32_t n_slabs, const bv_slab *slabs, bv_site_result *const *outs,
                            bv_group_result *const *gouts, void *stream);

/* Enqueue on `stream` (NULL: the engine's own) a wait for every submit issued so far.  Needed only with BV_FLAG_LANES (there
 * the submits run on internal streams); otherwise a no-op for the stream the submits were given. */
int bv_engine_join(bv_engine *e, void *stream);

/* The engine's own HIP stream (hipStream_t), so that a caller can order other work on it -- e.g.
 * wrap it (torch.cuda.ExternalStream) and issue the RCCL gather of the records behind the kernels. */
void *bv_engine_stream(bv_engine *e);

/* Block until every submit since the last wait has finished, on whatever streams they were issued (submits of one
 * engine share its scratch and are therefore serialised, also across streams).  Returns BV_ERR_SITE if any site of any
 * of those submits set BV_SITE_ZERO_FREQ (the reference would have thrown); the error counters are sticky across submits
 * and cleared by this call. */
int bv_engine_wait(bv_engine *e);

/* ---- sample-axis tile mode -----------------------------------------------------
 * The reference keeps the sample axis in batchfiles of B samples (`-B`, src/basetype_caller.cpp:
 * 419-453) and re-joins one row from each per site (:589-601).  Every quantity of the path is a
 * function of tallies that are additive over samples, so the engine can instead take column
 * tiles [n_sites][tile_width] one at a time (e.g. one batchfile after the other, streamed from
 * host DRAM: BASELINE config #5) and accumulate per-site state in HBM:
 *     bv_engine_tiles_begin(e, n_sites, n_samples_total, n_groups, with_ranks)
 *     bv_engine_tiles_add(e, tile, stream)      for every tile; tile->n_sites must match;
 *                                               tile->group_id covers the tile's samples
 *     bv_engine_tiles_finish(e, ref_base, out, gout, mem_kind, stream)   then bv_engine_wait()
 * Two realisations, chosen at bv_engine_tiles_begin:
 *   joined rows (default)  the tiles are copied into one [n_sites][n_samples_total] slab kept in HBM
 *                          (5 B per cell: 82 GB for 16 Ki sites x 1 M samples) and finish() runs the ordinary
 *                          two passes on it -- results are those of bv_engine_submit on the joined rows, bit for bit;
 *   per-site tallies       when that slab does not fit (or with BV_FLAG_TILE_STATE): additive per-site state
 *                          (~30 KB per site), global atomics; equal results to 1e-6 except that (a) read-position ranks
 *                          beyond the announced bound are not supported there (BV_SITE_RPR_RANGE, rpr_ranksum = NaN) and
 *                          (b) it has no rows left at finish(), so sites of <= 64 covered samples are not replayed in the
 *                          reference's per-sample order: where two allele subsets tie to the last bit, its pick may differ. */
/* `with_ranks`: 0 = tiles carry no mapq/rpr planes; 1 = they do; a value > 1 also announces an upper bound on the
 * read-position ranks (read length), which only the per-site-tally realisation needs: it keeps exact tallies of ranks
 * below max(1024, with_ranks rounded up to 1024) and flags sites beyond that (BV_SITE_RPR_RANGE).  Host tiles go through
 * a ring of staging buffers filled by a copy stream; a tile whose planes lie in one host allocation, one after the
 * other (bv_tile_packed_layout), crosses the link as ONE copy. */
int bv_engine_tiles_begin(bv_engine *e, uint32_t n_sites, uint32_t n_samples_total, uint32_t n_groups,
                          int with_ranks);
/* Layout of a packed host tile of `width` samples: pitch (cells per row, width rounded up to 16), the byte offsets of
 * base_strand, qual, mapq, rpr, group_id in ONE allocation of *total_bytes (absent planes: offset 0), each 256-aligned. */
int bv_tile_packed_layout(uint32_t n_sites, uint32_t width, int with_ranks, int with_groups, uint64_t *pitch,
                          uint64_t offsets[5], uint64_t *total_bytes);
int bv_engine_tiles_add(bv_engine *e, const bv_slab *tile, void *stream);
/* n_tiles tiles in the order given: equivalent to n_tiles calls of bv_engine_tiles_add.  Device-resident tiles of a joined-rows
 * job (a GPU-side producer: decoded batchfiles, another kernel's output) are moved to their columns by ONE launch per 256
 * tiles instead of one per tile -- a job of 200-sample tiles is otherwise launch-bound (8 us per tile against 0.4 us of
 * copying at 1 M samples x 8 Ki sites).  Host tiles and the per-site-tally realisation are added tile by tile. */
int bv_engine_tiles_add_many(bv_engine *e, uint32_t n_tiles, const bv_slab *tiles, void *stream);
int bv_engine_tiles_finish(bv_engine *e, const uint8_t *ref_base, bv_site_result *out, bv_group_result *gout,
                           uint32_t mem_kind, void *stream);

/* HIP-event timings (ms) of the last submit's kernels on the stream they ran on:
 * pass 1 (tally + solve, all sites) and pass 2 (rank sums + groups, variant sites).
 * Valid after bv_engine_wait(). */
int bv_engine_kernel_ms(bv_engine *e, float *pass1_ms, float *pass2_ms);

/* Accumulated HIP-event timings since the last reset: every submit records its own event
 * triplet (ring of 256 submits); totals are over all completed submits.  Used by bench.py
 * to quote the average launch duration of each pass over the timed region. */
int bv_engine_timing_reset(bv_engine *e);
int bv_engine_timing_get(bv_engine *e, double *pass1_total_ms, double *pass2_total_ms, uint32_t *n_submits);
/* The same with pass 1 split: on short rows (<= 49,152 samples) pass 1 is a streaming kernel (the HBM-bound one: tally of
 * every row) followed by a solve kernel; `stream_total_ms` is the streaming kernel alone, `pass1_total_ms` both.  On
 * long rows pass 1 is one kernel and the two figures coincide. */
int bv_engine_timing_get_ex(bv_engine *e, double *stream_total_ms, double *pass1_total_ms, double *pass2_total_ms,
                            uint32_t *n_submits);

/* ---- the host's log() on the device ---------------------------------------------
 * The reference's EM takes log() of per-sample marginals with the host libm (src/algorithm.h:243) and compares sums of
 * them; at tie-prone shallow sites (<= 64 covered samples; pop-groups of that size: where two allele subsets score
 * within 1e-7 of each other) the engine replays that arithmetic in the reference's
 * order, with the host libm's own log algorithm restated on the device.  The libm data table is located in the
 * running process and accepted only after the restated algorithm matched log() bit for bit on ~10^6 probes.
 *   bv_host_log_probe  1 when that check passes (needs no GPU); copies the 274 doubles of the table when table != NULL
 *   bv_host_log_eval   the restated algorithm on the host (table from bv_host_log_probe)
 *   bv_engine_host_log_exact   1 when engine e uses it; 0: the device library's log() (ulps from the host's)
 *   bv_engine_host_log_eval    y[i] = the DEVICE restatement at x[i] (host pointers; diagnostic used by the tests) */
#define BV_HOST_LOG_TABLE_DOUBLES 274
int bv_host_log_probe(double *table);
double bv_host_log_eval(const double *table, double x);
int bv_engine_host_log_exact(const bv_engine *e);
int bv_engine_host_log_eval(bv_engine *e, const double *x, double *y, uint32_t n);

/* Number of BV_SITE_VARIANT sites found by the last submit (valid after wait). */
int bv_engine_last_variant_count(bv_engine *e, uint32_t *n_variant);

/* Thread-safe: message of the last error on this engine (or global if e == NULL). */
const char *bv_last_error(const bv_engine *e);

/* ---- NUMA placement of host buffers (BASELINE config #5: tiles streamed from host DRAM on 8 GPUs) -----
 * The reference reads its batchfiles wherever the OS puts them (src/basetype_caller.cpp:586-611); a host that streams
 * pinned tiles to several GPUs should keep each GPU's tiles on the NUMA node its PCIe link hangs off.
 *   bv_device_numa_node            the node of HIP device `device` (sysfs numa_node of its PCI function), -1 when the
 *                                  platform does not say; pci_bdf (may be NULL) receives "dddd:bb:dd.f"
 *   bv_bind_thread_to_device_node  restricts the CALLING thread to the CPUs of that node (within its current affinity
 *                                  mask), so that buffers it allocates and first touches afterwards are node-local;
 *                                  returns the node, or -1 and changes nothing */
int bv_device_numa_node(int device, char *pci_bdf, size_t pci_bdf_len);
int bv_bind_thread_to_device_node(int device);

/* ---- measurement helper (bench only; not part of the reference surface) --------
 * Fill device planes with the synthetic pileup of SURVEY.md section 8(d) using a
 * counter-based RNG (stateless in (seed, site, sample)), so any rank can generate
 * any site range.  All plane pointers are device pointers; mapq/rpr may be NULL. */
typedef struct bv_synth_params {
    uint64_t seed;
    uint64_t site_offset; /* global index of row 0 (for sharding across ranks)       */
    float coverage;       /* P(cell covered), e.g. 0.08                              */
    float indel_frac;     /* fraction of covered cells that are indel tokens         */
    float qual_mean, qual_sd;
    uint32_t qual_min, qual_max;
} bv_synth_params;

int bv_synth_fill(int device, const bv_synth_params *p, uint32_t n_sites, uint32_t n_samples,
                  uint64_t pitch, uint8_t *base_strand, uint8_t *qual, uint8_t *mapq,
                  uint16_t *rpr, uint8_t *ref_base, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BASEVAR_AMD_H */
