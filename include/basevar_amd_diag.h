/*
 * basevar_amd_diag.h -- diagnostics, tuning switches, fault injection and measurement helpers of the engine behind
 * include/basevar_amd.h.  NOT part of the reference-facing surface: nothing here replaces an interface of
 * ShujiaHuang/basevar, a host that calls variants needs none of it, and no flag below changes a record (the tests hold
 * every one of them to the records of the default path).  Used by tests/, bench.py and tools/.
 */
#ifndef BASEVAR_AMD_DIAG_H
#define BASEVAR_AMD_DIAG_H

#include "basevar_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- bv_engine_config.flags: diagnostic and A/B bits ------------------------------------------------------------- */
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
#define BV_FLAG_NO_DOM 0x1000000u /* A-B runs / tests: deep rows keep the plain LDS adds in the rank-sum tallies (no dominant-value count, bv_lds_add16_dom) */
#define BV_FLAG_P2_TAIL_DMA 0x2000000u /* A-B runs / tests: the fused short-row kernel streams the variant sites' rank-sum rows through its LDS-DMA
                                     rings (rounds 4-6; what slabs with plain ranks always take), not into registers (csrc/bv_pass1_fused.hip: bv_f_p2_rows).
                                     Records do not depend on it. */
#define BV_FLAG_WAVE_SOLVER 0x10u /* diagnostic: short-row candidates and pop-group calls all take the one-per-wave solver (none the 16-lane one) */
#define BV_FLAG_LONG_ROW_FORM(n) (((uint32_t)(n) & 0xFu) << 8) /* tests: 2 = rows of more than 49,152 samples take the long-row kernel
                                     WITHOUT the team helpers whatever the launch size (default: launches of up to 65,536 sites spread
                                     their last solves over the idle tally waves).  Records do not depend on it. */
#define BV_FLAG_SHORT_ROW_FORM(n) (((uint32_t)(n) & 0xFu) << 12) /* tests / A-B runs: which kernels rows of 4,097 .. 49,152 samples take.
                                     0 = default: ONE persistent kernel for pass 1 and the variant sites' rank-sum rows
                                     (csrc/bv_pass1_fused.hip); 10 = that kernel for pass 1, pass 2 a launch of its own; 9 = the kernels
                                     shorter rows take: a streaming kernel, a solve kernel, a pass-2 kernel (csrc/bv_pass1_short.hip).
                                     Records do not depend on it. */
/* test only: a hand-off is reserved but never written -- the first candidate-queue entry of every workgroup of the fused
 * short-row kernel, the first ring slot of the long-row kernel's workgroup 0 --, so that its consumer runs into its bounded wait:
 * the launch must end (no hung GPU), bv_engine_wait must return BV_ERR_HIP naming the time-out, and the records are invalid.
 * bv_engine_create refuses the bit (BV_ERR_INVALID_ARG) unless the environment variable BASEVAR_AMD_FAULT_INJECT is set. */
#define BV_FLAG_FAULT_LOST_HANDOFF 0x40000000u

/* Which kernels the last launch of pass 1 took (valid after the submit; bench.py names the dominant kernel and its bytes by it) */
#define BV_FORM_SHORT_ROWS 0x1u   /* rows of <= 49,152 samples: the short-row kernels                                          */
#define BV_FORM_ONE_KERNEL 0x2u   /* pass 1 was ONE kernel: bv_pass1_kernel (long rows) / bv_p1s_fused_kernel (short rows);
                                     clear: bv_p1s_stream_kernel + bv_p1s_solve16_kernel                                       */
#define BV_FORM_PASS2_FUSED 0x4u  /* ... and that kernel streamed the variant sites' rank-sum rows too (no pass-2 launch)       */
int bv_engine_last_launch_form(bv_engine *e, uint32_t *form);

/* ---- per-pass HIP-event timings (bench.py) --------------------------------------------------------------------- */
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
int bv_engine_host_log_eval(bv_engine *e, const double *x, double *y, uint32_t n);

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
    uint32_t layout;      /* BV_SLAB_* bits of the slab to fill: BV_SLAB_RPR_TAGGED writes rpr = BV_RPR_TAGGED(call, rank) */
    uint32_t reserved_;
} bv_synth_params;

int bv_synth_fill(int device, const bv_synth_params *p, uint32_t n_sites, uint32_t n_samples,
                  uint64_t pitch, uint8_t *base_strand, uint8_t *qual, uint8_t *mapq,
                  uint16_t *rpr, uint8_t *ref_base, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BASEVAR_AMD_DIAG_H */
