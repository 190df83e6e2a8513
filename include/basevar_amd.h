/*
 * basevar_amd.h -- C ABI of the MI355X-native per-site basetype likelihood engine: the drop-in boundary for ONE path of
 * ShujiaHuang/basevar, the per-site caller (src/basetype.cpp:22-295, src/algorithm.h:44-255, htslib/kfunc.c:39-143,245-313)
 * as driven from _basevar_caller (src/basetype_caller.cpp:667-765), plus the arithmetic of _out_vcf_line / _out_cvg_line
 * (:1103-1260).  The reference has no plugin API; its seam is the C++ class `BaseType` (src/basetype.h:64-153) and the free
 * functions `strand_bias` / `ref_vs_alt_ranksumtest` (:168-181), called once per site.  This ABI is the batched equivalent --
 * S sites x N samples per submit -- and every entry point names the reference interface it replaces.
 *
 * Plain C: pointers and sizes only.  Functions return BV_OK (0) or a negative bv_status; bv_last_error() has the message (the
 * C++ wrapper rethrows std::runtime_error, the reference's error type, src/basetype.cpp:54-56,113-115,272).
 * Diagnostics, tuning switches and measurement helpers: basevar_amd_diag.h.
 */
#ifndef BASEVAR_AMD_H
#define BASEVAR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BV_ABI_VERSION 2 /* 2: bv_slab.layout (BV_SLAB_RPR_TAGGED) */

/* ---- cell encoding of the `base_strand` plane: one byte per (site, sample) = the first character of the reference's
 * per-sample token (src/basetype.cpp:50; domain per src/basetype_caller.cpp:1060-1077) plus the strand (basetype.cpp:257-264).
 *   bits 0-1: base call  0 'A'  1 'C'  2 'G'  3 'T'   (index in BASES, src/basetype.h:19);  bit 2: 1 = reverse strand '-'
 *   bit 3   : 1 = not a base call; bits 0-1 then select 0 'N' (uncovered, strand '.')  1 '+' (insertion)  2 '-' (deletion)
 *   bits 4-7: must be zero.   A covered call is a value 0..7 = the kernel's (strand, base) histogram row. */
#define BV_CELL_BASE_MASK 0x03u
#define BV_CELL_REV 0x04u
#define BV_CELL_NOCALL 0x08u
#define BV_CELL_N 0x08u
#define BV_CELL_INS 0x09u
#define BV_CELL_DEL 0x0Au
/* codes of bv_slab.ref_base[] and bv_site_result.alt[]: A, C, G, T, anything else */
enum { BV_BASE_A = 0, BV_BASE_C = 1, BV_BASE_G = 2, BV_BASE_T = 3, BV_BASE_OTHER = 4 };

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
#define BV_SITE_RPR_RANGE 0x40u    /* per-site-tally tile jobs only: the site has > 2,048 read-position ranks beyond the announced bound, or the
                                      job > 4 Mi such cells: rpr_ranksum is NaN (fewer take an exact path at finish) */
#define BV_SITE_LOG_APPROX 0x80u   /* a shallow site was replayed in the reference's per-sample order with the DEVICE library's log()
                                      (bv_engine_host_log_exact() == 0): exact ties may be picked differently; values within 1e-6 */

/* bv_slab.layout.  BV_SLAB_RPR_TAGGED: every word of the `rpr` plane also carries its cell's call in the three bits a
 * read-position rank of at most 8,191 leaves free:   rpr = rank | (call & 3) << 13 | (call >> 3) << 15   (BV_RPR_TAGGED(call,
 * rank); call = the cell's base_strand byte: base in bits 13-14, "not a base call" in bit 15, the strand is not needed).  The
 * ref_vs_alt_ranksumtest of a variant site (src/basetype.cpp:201-242 on mapqs / base_pos_ranks, caller.cpp:1151-1154) then reads
 * mapq + rpr only -- SURVEY 8d's 3 bytes per cell -- instead of base_strand + mapq + rpr.  The PRODUCER chooses it per slab, when
 * every rank of the slab is <= BV_RPR_TAG_MAX_RANK (every short-read cohort); the engine takes the tags on trust, as it takes
 * the planes.  Records are byte-identical to those of the plain layout.  Plane widths do not change. */
#define BV_SLAB_RPR_TAGGED 0x1u
#define BV_RPR_TAG_MAX_RANK 0x1FFFu
#define BV_RPR_TAGGED(call, rank) ((uint16_t)(((rank) & 0x1FFFu) | (((call) & 3u) << 13) | ((((call) >> 3) & 1u) << 15)))

/* Input: SoA planes [n_sites][pitch], one row per site, one cell per sample: `struct BatchInfo` (src/basetype.h:25-43), batched */
typedef struct bv_slab {
    uint32_t n_sites;
    uint32_t n_samples;
    uint64_t pitch;              /* cells per row, >= n_samples, multiple of 16       */
    const uint8_t *base_strand;  /* [n_sites][pitch]  align_bases[i][0] + map_strands */
    const uint8_t *qual;         /* [n_sites][pitch]  align_base_quals - 33           */
    const uint8_t *mapq;         /* [n_sites][pitch]  mapqs; may be NULL              */
    const uint16_t *rpr;         /* [n_sites][pitch]  base_pos_ranks; may be NULL; see BV_SLAB_RPR_TAGGED */
    const uint8_t *ref_base;     /* [n_sites] toupper(ref_base[0]) as BV_BASE_*; 4 = not ACGT  */
    const uint8_t *group_id;     /* [n_samples] pop-group index or BV_NO_GROUP; may be NULL; any alignment */
    uint32_t n_groups;           /* 0 if no groups (caller.cpp:746)                   */
    uint32_t mem_kind;           /* bv_mem_kind                                       */
    uint32_t layout;             /* BV_SLAB_* bits, 0 = the plain planes above        */
    uint32_t reserved_;          /* must be zero                                      */
} bv_slab;

/* Output: one fixed-size record per site (208 bytes).  Replaces the BaseType getters (src/basetype.h:121-151), StrandBiasInfo
 * (:57-62) and the INFO arithmetic of _out_vcf_line (src/basetype_caller.cpp:1113-1164). */
typedef struct bv_site_result {
    uint32_t depth[4];    /* get_base_depth('A','C','G','T'), basetype.cpp:58 (bit-exact) */
    uint32_t total_depth; /* get_total_depth(), basetype.cpp:59                          */
    uint32_t status;      /* BV_SITE_* bits                                              */
    uint32_t cvg_sb[4];   /* ref_fwd, ref_rev, alt_fwd, alt_rev; alt = all non-ref ACGT (_out_cvg_line, caller.cpp:1236-1245) */
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

/* Per (site, group) record (48 bytes), valid for BV_SITE_VARIANT sites only.  Replaces __gb()/lrt([REF]+alts)
 * (src/basetype_caller.cpp:756-777) and the "<group>_AF=" INFO values (caller.cpp:1182-1196). */
typedef struct bv_group_result {
    uint8_t n_alt;
    uint8_t alt[BV_MAX_ALT];
    uint8_t reserved[3];
    uint32_t total_depth; /* ACGT depth of the group's samples                           */
    uint32_t reserved2;
    double af[BV_MAX_ALT];
} bv_group_result;

/* ---- engine flags (bv_engine_config.flags).  Diagnostic / tuning bits live in basevar_amd_diag.h; none of them changes a record. */
#define BV_FLAG_TILE_STATE 0x8u         /* tile mode: per-site tallies even when the joined rows would fit (see tile mode)          */
#define BV_FLAG_HOST_ORDERED 0x80u      /* BV_MEM_HOST planes: the engine's copy stream first waits for everything queued on the
                                           caller's `stream` (callers that fill pinned planes asynchronously on it).  Default:
                                           host planes are COMPLETE when submit / tiles_add / tiles_finish is called and stay
                                           untouched until bv_engine_wait; the copies run ahead of `stream`, under earlier kernels */
#define BV_FLAG_LANES 0x10000000u       /* device-resident submits alternate between two internal streams with a scratch set each
                                           (kernels of consecutive submits overlap).  `stream` then only ORDERS the submit behind
                                           the caller's earlier work; records are complete after bv_engine_wait, or on a stream
                                           that called bv_engine_join.  Slabs and record buffers in flight must be distinct        */
#define BV_FLAG_SPARSE_TIMING 0x20000000u /* record the per-pass timing events for one launch in eight only (four event records cost
                                           ~15 us per launch); bv_engine_kernel_ms then reports the last TIMED launch               */

typedef struct bv_engine_config {
    int32_t device;        /* HIP device ordinal                                        */
    uint32_t max_sites;    /* largest n_sites per submit (sizes scratch)                */
    uint32_t max_samples;  /* largest n_samples: sizes the host staging area (0 = no staging) and the
                              log-factorial table of the strand-bias test (0 = 1 Mi entries)   */
    uint32_t flags;        /* BV_FLAG_* bits, normally 0                                */
    double min_af;         /* BaseType ctor arg 2 (basetype.cpp:30): already the float-rounded value of caller.cpp:122, see bv_min_af() */
} bv_engine_config;

typedef struct bv_engine bv_engine;

const char *bv_version(void); /* e.g. "basevar_amd 0.2 abi2 gfx950" */

/* min_af as the reference derives it: (double)std::min(float(100)/n_files, user_min_af), basetype_caller.cpp:122 */
double bv_min_af(uint32_t n_samples, float user_min_af);

/* One engine == one HIP stream + scratch on cfg.device.  Thread-compatible: one engine per host thread / GPU (mirrors one
 * BaseType per ThreadPool worker, src/basetype_caller.cpp:485-510). */
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

/* n_slabs device-resident slabs as ONE launch per pass: equivalent to n_slabs calls of bv_engine_submit (every record is
 * byte-identical), but the fixed tail of a launch (~0.1 ms) is paid once per queue -- for hosts that hold several small batches
 * (8,192-site batches then run at the rate of one large batch).  Chained, 16 slabs per launch, when the slabs are BV_MEM_DEVICE,
 * share n_samples, pitch, the presence of rank planes and the pop-group assignment (same group_id array, n_groups <= 32) and have
 * at most cfg.max_sites sites together; anything else is submitted slab by slab.  _g also takes gouts[k] = slab k's group records.
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
 * The reference keeps the sample axis in batchfiles of B samples (`-B`, src/basetype_caller.cpp:419-453) and re-joins one
 * row from each per site (:589-601).  The engine takes column tiles [n_sites][tile_width] one at a time instead (one batchfile
 * after the other, streamed from host DRAM: BASELINE config #5):
 *     bv_engine_tiles_begin(e, n_sites, n_samples_total, n_groups, with_ranks)
 *     bv_engine_tiles_add(e, tile, stream)    per tile; tile->n_sites must match; tile->group_id covers the tile's samples
 *     bv_engine_tiles_finish(e, ref_base, out, gout, mem_kind, stream)   then bv_engine_wait()
 * Two realisations, chosen at begin:
 *   joined rows (default)  tiles are copied into one [n_sites][n_samples_total] slab in HBM (5 B per cell) and finish() runs the
 *                          ordinary two passes on it: the records of bv_engine_submit on the joined rows, bit for bit;
 *   per-site tallies       when that slab does not fit (or BV_FLAG_TILE_STATE; <= 32 groups): additive per-site state (~30 KB per
 *                          site) -- equal to 1e-6; sites of <= 64 covered samples are replayed in the reference's per-sample order
 *                          from the first 64 covered cells kept per site; read-position ranks beyond the announced bound take
 *                          an exact slower path at finish().
 * `with_ranks`: 0 = no mapq/rpr planes; 1 = present; > 1 also announces an upper bound on the read-position ranks (read length)
 * for the per-site-tally realisation (exact tallies below max(1024, bound rounded up to 1024)).  Host tiles go through a ring
 * of staging buffers filled by a copy stream; a tile whose planes lie in ONE host allocation (bv_tile_packed_layout: pitch = width
 * rounded up to 16, plane offsets 256-aligned) crosses the link as one copy.  bv_engine_tiles_add_many = n_tiles calls of
 * bv_engine_tiles_add; device-resident tiles of a joined-rows job then move in ONE launch per 256 tiles. */
int bv_engine_tiles_begin(bv_engine *e, uint32_t n_sites, uint32_t n_samples_total, uint32_t n_groups,
                          int with_ranks);
int bv_tile_packed_layout(uint32_t n_sites, uint32_t width, int with_ranks, int with_groups, uint64_t *pitch,
                          uint64_t offsets[5], uint64_t *total_bytes);
int bv_engine_tiles_add(bv_engine *e, const bv_slab *tile, void *stream);
int bv_engine_tiles_add_many(bv_engine *e, uint32_t n_tiles, const bv_slab *tiles, void *stream);
int bv_engine_tiles_finish(bv_engine *e, const uint8_t *ref_base, bv_site_result *out, bv_group_result *gout,
                           uint32_t mem_kind, void *stream);

/* Packed host tiles (BASELINE config #5 is bound by the host link: 5 B per cell of which 92 % say "nobody covered"): a tile as
 * its covered cells only, site after site -- 7 bytes per covered cell, ~0.6 B per cell at 8 % coverage.  Entry k of site s
 * (row_start[s] <= k < row_start[s + 1]) is the cell of sample `sample[k]` of the tile: the token, quality, mapping quality and
 * read-position rank the reference reads for it from its batchfile row (src/basetype_caller.cpp:688-715); every cell without an
 * entry is an 'N'.  Same job protocol (begin / add ... / finish), same two realisations and byte-identical records as
 * bv_engine_tiles_add of the dense tile; dense and packed tiles may alternate within a job.  `rpr` holds plain ranks; with
 * layout = BV_SLAB_RPR_TAGGED (ranks <= 8,191) the engine writes the joined rows' rank words tagged itself.  Host tiles laid out
 * by bv_sparse_tile_packed_layout (one allocation: row_start, sample, base_strand, qual, mapq, rpr, group_id) cross the link as
 * one copy. */
typedef struct bv_sparse_tile {
    uint32_t n_sites, n_samples;  /* sites of the job; samples (columns) of this tile, <= 65,536 */
    uint32_t n_entries, n_groups; /* covered cells of the tile (= row_start[n_sites]); the job's group count */
    const uint32_t *row_start;    /* [n_sites + 1] */
    const uint16_t *sample;       /* [n_entries] column inside the tile */
    const uint8_t *base_strand, *qual, *mapq;  /* [n_entries]; mapq and rpr NULL for a job without rank planes */
    const uint16_t *rpr;          /* [n_entries] */
    const uint8_t *group_id;      /* [n_samples] or NULL */
    uint32_t mem_kind, layout;    /* bv_mem_kind; BV_SLAB_* of the job's tiles */
} bv_sparse_tile;
int bv_engine_tiles_add_sparse(bv_engine *e, const bv_sparse_tile *tile, void *stream);
int bv_sparse_tile_packed_layout(uint32_t n_sites, uint32_t n_entries, uint32_t width, int with_ranks, int with_groups,
                                 uint64_t offsets[7], uint64_t *total_bytes);

/* 1 when engine e replays tie-prone shallow sites (<= 64 covered samples) with the host libm's own log() restated on the
 * device, verified bit-exact at creation (the reference takes log() with the host libm, src/algorithm.h:243); 0: the device
 * library's log() -- values within 1e-6, exact ties undecided (BV_SITE_LOG_APPROX).  Probes: basevar_amd_diag.h. */
int bv_engine_host_log_exact(const bv_engine *e);

/* Number of BV_SITE_VARIANT sites found by the last submit (valid after wait). */
int bv_engine_last_variant_count(bv_engine *e, uint32_t *n_variant);

/* Thread-safe: message of the last error on this engine (or global if e == NULL). */
const char *bv_last_error(const bv_engine *e);

/* NUMA placement of host buffers (BASELINE config #5: tiles streamed from host DRAM on 8 GPUs; the reference reads its batchfiles
 * wherever the OS puts them, src/basetype_caller.cpp:586-611).  bv_device_numa_node: the node of HIP device `device` (sysfs
 * numa_node of its PCI function; -1: unknown), pci_bdf (may be NULL) receives "dddd:bb:dd.f".  bv_bind_thread_to_device_node
 * restricts the CALLING thread to that node's CPUs (within its affinity mask), so that buffers it allocates and first touches
 * afterwards are node-local; returns the node, or -1 and changes nothing. */
int bv_device_numa_node(int device, char *pci_bdf, size_t pci_bdf_len);
int bv_bind_thread_to_device_node(int device);

#ifdef __cplusplus
}
#endif
#endif /* BASEVAR_AMD_H */
