// bv_pass1_fused.hip -- short rows (4,097 .. 49,152 samples per site): pass 1 and the variant sites' pass-2 rows as ONE
// persistent kernel.
//
// bv_pass1_short.hip + bv_pass2.hip run these rows as three launches -- a streaming kernel, a solve kernel, the rank-sum
// kernel of pass 2: while the first and third run the chip's vector units are ~40 % busy, while the second runs HBM is idle,
// and its 0.10 ms per 100 k sites (16 % of the step at 10 k samples) is fully exposed.  Here all three are roles of one
// workgroup of 12 waves (one per CU, three per SIMD):
//
//   waves 0-7  (stream)  draw sites from the workgroup's cursor and stream their rows through private LDS rings filled
//                        by LDS-DMA, exactly one site per draw, three 4 KiB slots in flight per wave across row
//                        boundaries.  The per-slot control is straight-line scalar code (two asm blocks: a full slot, a
//                        row's last slot with two precomputed exec masks) -- the streaming kernel of bv_pass1_short.hip
//                        spends ~110 scalar instructions and ~15 branches per slot on the same job.  Per row: the
//                        strand x base totals, the candidate test, ONE 48-byte summary store; a candidate's bins leave
//                        through a 512-byte LDS stage as two 64-lane stores.  A row's stores are counted: the three slot
//                        waits that follow it allow exactly that many more operations outstanding (s_waitcnt vmcnt(8 + S)),
//                        so a candidate row no longer drains the ring.
//   waves 8-11 (solve)   take the candidates from two queues in LDS (three or four active bases first), four sites per
//                        wave on 16-lane groups (bv_solver16.h), and finish the non-candidate sites one lane per site in
//                        blocks of 64 as soon as every streaming wave has passed them.  A variant site goes into a third
//                        queue with what its rank sums need (class table, REF / ALT depths) -- between the two phases of
//                        its solve: the LRT decides the alleles, QUAL and the strand-bias tests then run beside the site's row.
//   pass-2 rows          (FUSE2: rank planes given) a wave past its pass-1 rows takes variant sites from that queue and tallies
//                        mapq + ranks (+ calls) as bv_pass2_dma_kernel tallies them -- from registers, by plain loads
//                        (bv_f_p2_rows; the solver waves too, once they are out of jobs), or with BV_FLAG_P2_TAIL_DMA through the
//                        ring (again four 1 KiB pieces per slot: the counted waits are unchanged).  The solvers' last jobs
//                        run UNDER these rows, and pass 2 has no launch, fill or drain of its own.
//   no row to stream     a streaming wave whose ring is idle (waiting for a variant row, or done) solves with the ring as
//                        scratch; only such waves (12 KiB each) take the candidates that need the wave solver of bv_solver.h
//                        (shallow sites, phred-0 calls, more than 128 bins).
//
// Hand-off.  A row's summary and bins go to HBM scratch as before; they are PUBLISHED (queue entry / the wave's `pub` mark in
// LDS) one row later, behind an s_waitcnt that covers exactly the stores of that row (vmcnt counts in issue order), and read
// by the solvers through the L2 (sc1 loads; summaries of neighbouring sites share lines).  Producer and consumer are waves
// of one CU: one L2, no cross-XCD visibility involved.  Streaming waves never wait for solver waves except on a full queue,
// and every such wait is bounded (BV_F_SPIN_MAX -> BV_CTR_TIMEOUT: a loud failure, not a hung GPU).
//
// Two register allocations.  The streaming loop (bv_f_stream_until_idle) and the row re-do (bv_f_p2_redo) are functions of
// their own, NOT inlined: see the comment at the first.  The solver step has one call site for both kinds of waves.
//
// Reference functions realised: those of bv_pass1_short.hip and bv_pass2_dma_kernel (src/basetype.cpp:22-295,
// src/algorithm.h:44-255, htslib/kfunc.c:39-143,197-313; rank sums: src/basetype.cpp:201-242 via caller.cpp:1151-1154); every
// record is byte-identical to the three-launch form's (same bins, same order, same solver code).  HBM-bound by design
// (2 B per cell of every row + 4 B per cell of a variant row, the call byte of a variant row read twice); no MFMA
// (categorical tallies).  Measurements, rejected variants: DESIGN.md section 4.3.
#define BV_LNFACT_TABLE_ONLY 1  /* rows of at most 65,535 samples: see bv_lnfact */
#include "bv_kernels.h"

#include "bv_solver.h"
#include "bv_solver16.h"
#include "bv_tally.h"

#include "bv_short.h"
#include "bv_pass2_sweep.h"

#include <type_traits>

#ifndef BV_F_NS
#define BV_F_NS 8                         /* streaming waves per workgroup */
#endif
#ifndef BV_F_NV
#define BV_F_NV 4                         /* dedicated solver waves per workgroup (7 + 5 measured -1.2 %; round 5, lean solver: 8 + 3 -0.8 %, 8 + 2 -4 %) */
#endif
static_assert(BV_F_NV >= 1, "the streaming waves wait on a full queue: somebody must be emptying it");
#define BV_F_NW (BV_F_NS + BV_F_NV)
#ifndef BV_F_K
#define BV_F_K 3                          /* ring slots per streaming wave */
#endif
// the counted waits (every slot is four vector-memory operations): BV_F_W_ALL -- at most the K slots in flight are outstanding;
// BV_F_W_OLDEST -- the oldest of K slots has landed; _1 / _3 -- the same with one / three younger stores outstanding.
// (Round 6 tried counts kept per slot -- a slot of the tagged pass-2 rows then needs no fourth, one-lane load -- with the wait picked
// by a chain of scalar compares: the chain cost 4-5 % of the launch, the saved loads gained nothing measurable.  Not kept.)
#if BV_F_K == 3
#define BV_F_W_ALL "s_waitcnt vmcnt(12)"
#define BV_F_W_OLDEST "s_waitcnt vmcnt(8)"
#define BV_F_W_OLDEST_1 "s_waitcnt vmcnt(9)"
#define BV_F_W_OLDEST_3 "s_waitcnt vmcnt(11)"
#elif BV_F_K == 2
#define BV_F_W_ALL "s_waitcnt vmcnt(8)"
#define BV_F_W_OLDEST "s_waitcnt vmcnt(4)"
#define BV_F_W_OLDEST_1 "s_waitcnt vmcnt(5)"
#define BV_F_W_OLDEST_3 "s_waitcnt vmcnt(7)"
#elif BV_F_K == 4
#define BV_F_W_ALL "s_waitcnt vmcnt(16)"
#define BV_F_W_OLDEST "s_waitcnt vmcnt(12)"
#define BV_F_W_OLDEST_1 "s_waitcnt vmcnt(13)"
#define BV_F_W_OLDEST_3 "s_waitcnt vmcnt(15)"
#else
#error "BV_F_K: 2, 3 or 4 ring slots"
#endif
// -DBV_TEAM_DEBUG -DBV_PHASE_DEBUG: where the streaming waves' cycles go (s_memtime around the phases of a slot, summed over
// the launch's waves in units of 16 cycles; printed by bv_engine_wait with the [fused debug] lines)
#ifdef BV_PHASE_DEBUG
#define BV_PH_DECL uint32_t ph_[24]; for (int i_ = 0; i_ < 24; ++i_) ph_[i_] = 0u; uint32_t ph_t_ = (uint32_t)__builtin_amdgcn_s_memtime(); const uint32_t ph_t0_ = ph_t_; uint32_t ph_i0_ = 0, ph_i1_ = 0, ph_i2_ = 0, ph_n1_ = 0, ph_n2_ = 0
#define BV_PH(i) do { const uint32_t n_ = (uint32_t)__builtin_amdgcn_s_memtime(); ph_[(i) + ((st & BV_FS_P1_FIN) ? 12 : 0)] += n_ - ph_t_; ph_t_ = n_; } while (0)
#define BV_PH_COUNT(i) (++ph_[(i) + ((st & BV_FS_P1_FIN) ? 12 : 0)])
#define BV_PH_FLUSH(ctr, lane) do { ph_[9] = (uint32_t)__builtin_amdgcn_s_memtime() - ph_t0_; ph_[11] = ph_i1_; ph_[23] = ph_i2_; if ((lane) == 0) { atomicAdd(&(ctr)[BV_CTR_WORDS + 4244], ph_n1_); atomicAdd(&(ctr)[BV_CTR_WORDS + 4245], ph_n2_); } if ((lane) == 0) for (int i_ = 0; i_ < 24; ++i_) atomicAdd(&(ctr)[BV_CTR_WORDS + 4220 + i_], (i_ % 12) == 7 ? ph_[i_] : ph_[i_] >> 4); } while (0)
#else
#define BV_PH_DECL
#define BV_PH(i)
#define BV_PH_COUNT(i)
#define BV_PH_FLUSH(ctr, lane)
#endif
#define BV_F_SLOT_WORDS 1024              /* pass-1 rows: 2 KiB of calls, 2 KiB of phreds; pass-2 rows: 1 KiB of calls, 1 KiB of mapq, 2 KiB of ranks */
#define BV_F_QCAP 256                     /* entries per candidate queue (ring buffers) */
#define BV_F_QVCAP 128                    /* entries of the variant queue */
#define BV_F_EMPTY 0xFFFFFFFFu
// Every wait on another wave's LDS write is bounded (~1-2 s of s_sleep): a wave that gives up sets the sticky BV_CTR_TIMEOUT
// counter -- the submit then fails loudly in bv_engine_wait -- instead of hanging the GPU on a protocol error.
#define BV_F_SPIN_MAX (1u << 22)
// No issue priorities (s_setprio).  Round 4 ran the solver waves at priority 3: their chains of dependent FP64 operations lost
// 2-3 x beside two streaming waves per SIMD at equal priority (28 candidates waiting at the end of the pass-1 rows against 1).
// What they were really losing to was scratch memory -- reloads of hoisted loop invariants inside those chains, a memory trip
// each (round 5, see bv_f_stream_until_idle and the kernel's loop).  With those gone the priority COSTS: equal priorities
// measured +3 % at 100 k sites (187 against 181 M sites/s, eight interleaved pairs), +4 % at 524 k (200-207 against 192-201);
// streaming waves above the solvers (2 over 0) +2 %, (3 over 1) -1 %.
#define BV_F_MIN_JOB 4u                   /* sites per job of the 16-lane solver while rows are still streaming */
// control words in LDS
#define BV_FC_CURSOR 0                    /* sites of the workgroup's range handed out so far */
#define BV_FC_Q3_TAIL 1                   /* candidates with >= 3 active bases: reserved / claimed positions */
#define BV_FC_Q3_HEAD 2
#define BV_FC_Q2_TAIL 3                   /* candidates with <= 2 active bases */
#define BV_FC_Q2_HEAD 4
#define BV_FC_QH_TAIL 5                   /* candidates for the wave solver: entries of the workgroup's slice of cand_list (HBM) */
#define BV_FC_QH_HEAD 6
#define BV_FC_BLK_HEAD 7                  /* blocks of 64 sites (non-candidates) claimed */
#define BV_FC_NDONE 8                     /* streaming waves that have published their last pass-1 row */
#define BV_FC_QV_TAIL 9                   /* variant sites whose rank-sum rows (pass 2) are to be streamed */
#define BV_FC_QV_HEAD 10
#define BV_FC_BUSY 11                     /* solver jobs in flight (each may still add to the variant queue) */
#define BV_FC_OV_TAIL 12                  /* variant sites that did not fit the LDS queue: entries of the workgroup's overflow list (HBM) written */
#define BV_FC_OV_HEAD 13                  /* ... and claimed by streaming waves */
#define BV_FC_OV_LOCK 14                  /* one producer wave at a time appends to the list */
// (A workgroup owns a contiguous range of sites.  Dealing the launch's last eighth in 16-site chunks from a global counter --
// the XCDs stream at rates 7 % apart -- was built and measured in round 4: the pass-1 rows end within 21 us instead of 30, but
// the median moves up by as much and the launch ends with its last VARIANT rows, which stay local: 166-174 against 175-176 M
// sites/s.  docs/history/DESIGN_round4.md.)
// a streaming wave's flags that outlive a call of bv_f_stream_until_idle
#define BV_FS_P_DONE 1u                   /* no row will ever come again */
#define BV_FS_CUR_DONE 8u                 /* the cursor is exhausted */
#define BV_FS_P1_FIN 16u                  /* the wave's last pass-1 row is published, the wave counted in NDONE */
// kinds of rows in a streaming wave's ring
#define BV_FK_P1 1u                       /* calls + phreds of a site: the pass-1 tally */
#define BV_FK_P2 2u                       /* calls + mapq + ranks of a variant site: the two rank sums of pass 2 */

union __attribute__((aligned(16))) BvFusedRing {
    uint32_t slot[BV_F_K][BV_F_SLOT_WORDS];
    struct {                                  // while the wave has no row in flight
        BvP1sWaveScratch ws;
        uint32_t vl[64];
    } solve;
};
struct __attribute__((aligned(16))) BvFusedShared {
    uint32_t hist[BV_F_NS][BV_S_HWORDS + BV_S_OVF + 8];  // per streaming wave: [(rev<<2)|base][phred < 128], overflow rows; pass-2 rows: hm[2][256], hr[2][256]
    BvFusedRing ring[BV_F_NS];
    uint32_t grp[BV_F_NV][4][BV_G16_GRP_WORDS];          // the solver waves' group scratches
    uint32_t vl[BV_F_NV][64];                            // ... and their variant sites since the last flush
    double tab_hit[BV_QBINS], tab_miss[BV_QBINS];
    double tab_loghit[BV_QBINS], tab_logmiss[BV_QBINS];  // (from HBM these cost the 16-lane solver a memory trip per slot of bins)
    uint32_t stage[BV_F_NS][128];                        // a candidate's compacted bins on their way out
    uint32_t q3[BV_F_QCAP], q2[BV_F_QCAP];
    uint32_t qv[BV_F_QVCAP][4];                          // site, class table, n_ref | n_alt << 16, 2-bit lut
    uint32_t pub[BV_F_NS];                               // every pass-1 row of wave w below site pub[w] is published
    uint32_t ctl[16];
};
static_assert(sizeof(BvFusedShared) <= 160 * 1024, "one workgroup per CU must fit the LDS");
static_assert(4 * BV_G16_GRP_WORDS >= BV_S_HWORDS, "a solver wave's group scratch serves as its rank-sum histogram (bv_f_p2_rows)");
static_assert(BV_F_K < 3 || sizeof(BvFusedRing) == sizeof(uint32_t) * BV_F_K * BV_F_SLOT_WORDS, "the solver scratch must fit the ring");

// ---- LDS-DMA of one slot: four 1 KiB pieces (64 lanes x 16 bytes from p_i + v_i) to d0, d0 + 1 KiB, ..., always four (the
// counted waits rely on it).  M0 is written inside the statement; the s_add between the write and the load is the wait state.
__device__ __forceinline__ void bv_f_glds4(uint32_t d0, const uint8_t *p0, uint32_t v0, const uint8_t *p1, uint32_t v1,
                                           const uint8_t *p2, uint32_t v2, const uint8_t *p3, uint32_t v3) {
    uint32_t keep, t;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[d0]\n\t"
        "s_add_u32 %[t], %[d0], 0x400\n\t"
        "global_load_lds_dwordx4 %[v0], %[p0] nt\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_add_u32 %[t], %[d0], 0x800\n\t"
        "global_load_lds_dwordx4 %[v1], %[p1] nt\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_add_u32 %[t], %[d0], 0xc00\n\t"
        "global_load_lds_dwordx4 %[v2], %[p2] nt\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[v3], %[p3] nt\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep), [t] "=&s"(t)
        : [d0] "s"(d0), [p0] "s"(p0), [p1] "s"(p1), [p2] "s"(p2), [p3] "s"(p3), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3)
        : "memory", "scc");
}
// A row's last slot: lanes past the row's end load nothing (mA: pieces 0 and 2, mB: pieces 1 and 3 -- never empty: a piece
// that lies wholly past the end is "loaded" by lane 0 alone from valid bytes of the slot; its cells are masked in the tally).
__device__ __forceinline__ void bv_f_glds4_masked(uint32_t d0, const uint8_t *p0, uint32_t v0, const uint8_t *p1, uint32_t v1,
                                                  const uint8_t *p2, uint32_t v2, const uint8_t *p3, uint32_t v3,
                                                  unsigned long long mA, unsigned long long mB) {
    uint32_t keep, t;
    unsigned long long sv;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[mA]\n\t"
        "s_mov_b32 m0, %[d0]\n\t"
        "s_add_u32 %[t], %[d0], 0x800\n\t"
        "global_load_lds_dwordx4 %[v0], %[p0] nt\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_add_u32 %[t], %[d0], 0x400\n\t"
        "global_load_lds_dwordx4 %[v2], %[p2] nt\n\t"
        "s_mov_b64 exec, %[mB]\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_add_u32 %[t], %[d0], 0xc00\n\t"
        "global_load_lds_dwordx4 %[v1], %[p1] nt\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[v3], %[p3] nt\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep), [t] "=&s"(t), [sv] "=&s"(sv)
        : [d0] "s"(d0), [p0] "s"(p0), [p1] "s"(p1), [p2] "s"(p2), [p3] "s"(p3), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3),
          [mA] "s"(mA), [mB] "s"(mB)
        : "memory", "scc");
}
// A pass-2 slot of the tagged rank layout (BV_SLAB_RPR_TAGGED): no call bytes -- 1 KiB of mapq and 2 KiB of ranks to the slot's
// second, third and fourth KiB.  Still FOUR loads, because the counted waits count four per slot whatever its kind: the first is
// lane 0 alone, 16 bytes of the mapq piece into the unused first KiB (the same line the second load fetches: no traffic of its own).
__device__ __forceinline__ void bv_f_glds4_tag(uint32_t d0, const uint8_t *p1, uint32_t v1, const uint8_t *p2, uint32_t v2, uint32_t v3) {
    uint32_t keep, t;
    unsigned long long sv;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "s_mov_b32 m0, %[d0]\n\t"
        "s_add_u32 %[t], %[d0], 0x400\n\t"
        "global_load_lds_dwordx4 %[v1], %[p1] nt\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_add_u32 %[t], %[d0], 0x800\n\t"
        "global_load_lds_dwordx4 %[v1], %[p1] nt\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_add_u32 %[t], %[d0], 0xc00\n\t"
        "global_load_lds_dwordx4 %[v2], %[p2] nt\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[v3], %[p2] nt\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep), [t] "=&s"(t), [sv] "=&s"(sv)
        : [d0] "s"(d0), [p1] "s"(p1), [p2] "s"(p2), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3)
        : "memory", "scc");
}
// Four pieces, each under a lane mask of its own (never empty): the last slot of a pass-2 row, whose 2 KiB of ranks end at twice the
// lane count of its 1 KiB of mapq.
__device__ __forceinline__ void bv_f_glds4_m4(uint32_t d0, const uint8_t *p0, uint32_t v0, unsigned long long m0, const uint8_t *p1, uint32_t v1,
                                              unsigned long long m1, const uint8_t *p2, uint32_t v2, unsigned long long m2, const uint8_t *p3,
                                              uint32_t v3, unsigned long long m3) {
    // (a row's last slot only: the operands are made scalar here, whatever the compiler believes about them at the call site)
    auto u32 = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    auto u64 = [&](unsigned long long x) { return ((unsigned long long)u32((uint32_t)(x >> 32)) << 32) | u32((uint32_t)x); };
    d0 = u32(d0); m0 = u64(m0); m1 = u64(m1); m2 = u64(m2); m3 = u64(m3);
    p0 = (const uint8_t *)(uintptr_t)u64((uintptr_t)p0); p1 = (const uint8_t *)(uintptr_t)u64((uintptr_t)p1);
    p2 = (const uint8_t *)(uintptr_t)u64((uintptr_t)p2); p3 = (const uint8_t *)(uintptr_t)u64((uintptr_t)p3);
    uint32_t keep, t;
    unsigned long long sv;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b32 m0, %[d0]\n\t"
        "s_mov_b64 exec, %[m0_]\n\t"
        "s_add_u32 %[t], %[d0], 0x400\n\t"
        "global_load_lds_dwordx4 %[v0], %[p0] nt\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_mov_b64 exec, %[m1_]\n\t"
        "s_add_u32 %[t], %[d0], 0x800\n\t"
        "global_load_lds_dwordx4 %[v1], %[p1] nt\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_mov_b64 exec, %[m2_]\n\t"
        "s_add_u32 %[t], %[d0], 0xc00\n\t"
        "global_load_lds_dwordx4 %[v2], %[p2] nt\n\t"
        "s_mov_b32 m0, %[t]\n\t"
        "s_mov_b64 exec, %[m3_]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[v3], %[p3] nt\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep), [t] "=&s"(t), [sv] "=&s"(sv)
        : [d0] "s"(d0), [p0] "s"(p0), [p1] "s"(p1), [p2] "s"(p2), [p3] "s"(p3), [v0] "v"(v0), [v1] "v"(v1), [v2] "v"(v2), [v3] "v"(v3),
          [m0_] "s"(m0), [m1_] "s"(m1), [m2_] "s"(m2), [m3_] "s"(m3)
        : "memory", "scc");
}
// One wave-level compare-and-swap on an LDS word by lane 0, the old value in an SGPR (no divergent branch, no vector-memory
// operation: see bv_lds_fetch_add_wave)
__device__ __forceinline__ uint32_t bv_f_lds_cas_wave(uint32_t lds_addr, uint32_t expect, uint32_t desired) {
    uint32_t r, tc, td;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "v_mov_b32 %[tc], %[cmp]\n\t"
        "v_mov_b32 %[td], %[val]\n\t"
        "ds_cmpst_rtn_b32 %[tc], %[adr], %[tc], %[td]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %[r], %[tc]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [r] "=&s"(r), [tc] "=&v"(tc), [td] "=&v"(td), [sv] "=&s"(sv)
        : [adr] "v"(lds_addr), [cmp] "s"(expect), [val] "s"(desired)
        : "memory");
    return r;
}
// One wave-level fetch-and-add on a word of device memory by lane 0, the old value in an SGPR (as the compiler emits a returning
// agent-scope atomicAdd: `global_atomic_add ... sc0`), waited for inside the statement: the wave's LDS-DMA ring drains with it.
__device__ __forceinline__ uint32_t bv_f_global_fetch_add_wave(const uint32_t *p, uint32_t v) {
    uint32_t r, t, z;
    unsigned long long sv;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "v_mov_b32 %[t], %[val]\n\t"
        "v_mov_b32 %[z], 0\n\t"
        "global_atomic_add %[t], %[z], %[t], %[base] sc0\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_readfirstlane_b32 %[r], %[t]\n\t"
        "s_mov_b64 exec, %[sv]"
        : [r] "=&s"(r), [t] "=&v"(t), [z] "=&v"(z), [sv] "=&s"(sv)
        : [val] "s"(v), [base] "s"(p)
        : "memory");
    return r;
}
// a reference base through the scalar cache (lgkmcnt: a vector load would sit in the vmcnt queue of the ring)
typedef const __attribute__((address_space(4))) uint32_t *bv_c32;
__device__ __forceinline__ uint32_t bv_f_ref_scalar(const uint8_t *ref_base, uint32_t site) {
    const uintptr_t p = (uintptr_t)ref_base + site;
    const uint32_t w = *(bv_c32)(p & ~(uintptr_t)3);
    return (w >> (8u * (uint32_t)(p & 3))) & 0xFFu;
}
__device__ __forceinline__ uint32_t bv_f_lds_read_u(const uint32_t *p) {  // p: a word of the kernel's __shared__ block
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)*(const volatile __attribute__((address_space(3))) uint32_t *)p);
}
__device__ __forceinline__ uint32_t bv_f_keep_mask(int kept) {  // dword mask of the first `kept` bytes
    return kept >= 4 ? 0xFFFFFFFFu : (kept <= 0 ? 0u : (1u << (8 * kept)) - 1u);
}

// the cells of two 16-byte chunks per lane (first and second KiB of a slot) -> the wave's histogram
// (The dense-row bank swizzle of the long-row kernel, bv_tally_chunk<.., SWZ>, was tried here too, per wave and row: at full
// coverage it gained nothing -- 120 against 122 M sites/s at 10,000 samples --, rows of coverage 0.3 lost 12 % to its VALU work and
// the un-swizzle, and its registers cost the sparse rows 3 %: not kept.)
__device__ __forceinline__ void bv_f_tally2(bv_u32x4 vbA, bv_u32x4 vqA, bv_u32x4 vbB, bv_u32x4 vqB, uint32_t *hist, uint32_t one) {
    // A phred byte >= 128 (invalid input) would carry into its neighbour under the shift below: such a slot takes the
    // exact cell-by-cell path (wave-uniform branch; never taken on valid data).
    const uint32_t hi = ((vqA.x | vqA.y | vqA.z) | (vqA.w | vqB.x | vqB.y) | (vqB.z | vqB.w)) & 0x80808080u;
    if (__builtin_expect(__ballot(hi != 0u) != 0ull, 0)) {
        const uint32_t wb[8] = {vbA.x, vbA.y, vbA.z, vbA.w, vbB.x, vbB.y, vbB.z, vbB.w};
        const uint32_t wq[8] = {vqA.x, vqA.y, vqA.z, vqA.w, vqB.x, vqB.y, vqB.z, vqB.w};
#pragma unroll 1
        for (int j = 0; j < 32; ++j) {
            const uint32_t c = (wb[j >> 2] >> (8 * (j & 3))) & 0xFFu, p = (wq[j >> 2] >> (8 * (j & 3))) & 0xFFu;
            if (c < 8u) atomicAdd(p < 128u ? &hist[(c << 7) | p] : &hist[BV_S_HWORDS + c], 1u);
        }
        return;
    }
    // X = call << 8 | phred << 1 = twice the word index of the 8 x 128 histogram; call < 8 <=> X < 0x800
    vqA.x <<= 1; vqA.y <<= 1; vqA.z <<= 1; vqA.w <<= 1;
    vqB.x <<= 1; vqB.y <<= 1; vqB.z <<= 1; vqB.w <<= 1;
    bv_tally_chunk<1>(vbA, vqA, hist, one);
    bv_tally_chunk<1>(vbB, vqB, hist, one);
}

// ------------------------------------------------------------------------------ the solver side
struct BvFusedSolver {
    BvSolveArgs sa;
    uint32_t *grp;   // this wave's four group scratches (LDS)
    uint32_t *vl;    // this wave's variant sites since the last flush (LDS)
    uint32_t n_vl;
    BvP1sWaveScratch *big;  // the wave solver's scratch, or NULL: this wave cannot take the hard candidates
    bool fuse2;      // variant sites also go to the workgroup's variant queue (their pass-2 rows are streamed by this kernel)
};
__device__ __forceinline__ void bv_f_flush_vl(const BvP1ShortArgs &a, BvFusedSolver &v, int lane) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&a.counters[BV_CTR_VARIANTS], v.n_vl);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    bv_lrt_sync<0>();
    if ((uint32_t)lane < v.n_vl) a.var_list[base + (uint32_t)lane] = v.vl[lane];
    bv_lrt_sync<0>();
    v.n_vl = 0;
}
// claim `least` to `most` entries of a queue: returns the number claimed (0: too few there, or another wave was faster) and
// the first position
__device__ __forceinline__ uint32_t bv_f_claim(uint32_t *ctl, int tail_i, int head_i, uint32_t least, uint32_t most, uint32_t &first, int lane) {
    const uint32_t h = bv_f_lds_read_u(&ctl[head_i]), t = bv_f_lds_read_u(&ctl[tail_i]);
    if (t - h < least) return 0u;
    uint32_t n = t - h;
    if (n > most) n = most;
    uint32_t old = 0;
    if (lane == 0) old = atomicCAS(&ctl[head_i], h, h + n);
    old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
    first = h;
    return old == h ? n : 0u;
}
// the entry at `pos` (its producer reserved the position before writing it: wait for the site number), handed back empty
__device__ __forceinline__ uint32_t bv_f_take(uint32_t *q, uint32_t pos) {
    volatile __attribute__((address_space(3))) uint32_t *e = (volatile __attribute__((address_space(3))) uint32_t *)q + (pos & (BV_F_QCAP - 1u));
    uint32_t s = *e, spins = 0;
    while (s == BV_F_EMPTY && ++spins < BV_F_SPIN_MAX) { __builtin_amdgcn_s_sleep(4); s = *e; }
    *e = BV_F_EMPTY;
    return s;  // (BV_F_EMPTY after a time-out: the caller flags it)
}
// What the rank sums of pass 2 need of a variant site (bv_pass2_dma_kernel forms the same from the record): the class of
// every base -- byte b of L: 0x80 REF, 0x81 ALT, 0xFF neither (caller.cpp:1151-1157) --, the 2-bit form of it for the window
// sweeps, and the REF / ALT depths.  aw0 / aw1: the record's bytes n_alt, alt[0..3] as stored at bv_site_result::n_alt.
__device__ __forceinline__ void bv_f_p2_facts(int ref, const uint32_t depth[4], uint32_t aw0, uint32_t aw1, uint32_t &L, uint32_t &n12,
                                              uint32_t &lut) {
    L = 0xFFFFFFFFu; lut = 0xAAu;
    uint32_t n1 = 0, n2 = 0;
    const int n_alt = (int)(aw0 & 0xFFu);
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
    n12 = n1 | (n2 << 16);  // both at most the row length (<= 49,152)
}
// Variant sites into the workgroup's variant queue; called by the WHOLE wave, `want` marks the lanes that have one (their
// records are complete in memory).  The queue in LDS takes them while it is at most BV_F_QV_ROOM full: the positions are
// reserved in one atomic step per lane and counted at once, so whatever number of lanes of whatever number of waves pass that
// test together (at most 12 waves x 4), the queue cannot overflow and nobody ever waits for a slot of an entry that is not
// yet claimed.  Beyond that the sites go to the workgroup's OVERFLOW LIST in HBM (a.ovf: one 16-byte entry per site of the
// workgroup's range, so it cannot fill), appended by one wave at a time, published by the LDS word OV_TAIL behind an
// s_waitcnt that covers the entries' stores.
// Why not simply wait for room (as until round 5): the queue is emptied by the streaming waves, which themselves wait -- for
// room in the candidate queues that the solver waves empty, or, once past their pass-1 rows, inside this very function as
// solvers.  A workgroup with more variant sites in flight than the queue holds (few workgroups, many variants: 732 sites per
// workgroup in the campaign that found it) then stopped until the bounded waits gave up.  Now no solver ever waits on a
// streaming wave.
#define BV_F_QV_ROOM (BV_F_QVCAP - 48u)
__device__ __forceinline__ void bv_f_push_variants(const BvP1ShortArgs &a, BvFusedShared &sh, uint32_t B0, bool want, uint32_t site, uint32_t L,
                                                   uint32_t n12, uint32_t lut, int lane) {
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return;
    if (bv_f_lds_read_u(&sh.ctl[BV_FC_QV_TAIL]) - bv_f_lds_read_u(&sh.ctl[BV_FC_QV_HEAD]) <= BV_F_QV_ROOM) {
        if (want) {
            const uint32_t pos = atomicAdd(&sh.ctl[BV_FC_QV_TAIL], 1u);
            volatile __attribute__((address_space(3))) uint32_t *e =
                (volatile __attribute__((address_space(3))) uint32_t *)&sh.qv[pos & (BV_F_QVCAP - 1u)][0];
            // (the entry that had this slot is claimed -- the occupancy test -- but its reader may be a few instructions from
            // handing the slot back)
            uint32_t spins = 0;
            while (e[0] != BV_F_EMPTY && spins < BV_F_SPIN_MAX) { __builtin_amdgcn_s_sleep(1); ++spins; }
            if (spins == BV_F_SPIN_MAX) atomicOr(&a.counters[BV_CTR_TIMEOUT], BV_TMO_PUSH_VARIANT);  // (nothing overwritten; loud)
            else {
                e[1] = L; e[2] = n12; e[3] = lut;
                e[0] = site;  // (LDS operations of one wave execute in order: the entry is whole when its site number appears)
            }
        }
        return;
    }
    // ---- the overflow list
    uint32_t spins = 0, got = 1u;
    do {
        if (lane == 0) got = atomicCAS(&sh.ctl[BV_FC_OV_LOCK], 0u, 1u);
        got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
        if (got != 0u) __builtin_amdgcn_s_sleep(2);
    } while (got != 0u && ++spins < BV_F_SPIN_MAX);
    if (got != 0u) { if (lane == 0) atomicOr(&a.counters[BV_CTR_TIMEOUT], BV_TMO_PUSH_VARIANT); return; }
    const uint32_t base = bv_f_lds_read_u(&sh.ctl[BV_FC_OV_TAIL]);
    if (want) {
        uint32_t *e = a.ovf + 4u * (size_t)(B0 + base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
        *reinterpret_cast<uint4 *>(e) = make_uint4(site, L, n12, lut);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the entries are in the L2 before their count says so
    if (lane == 0) {
        *(volatile __attribute__((address_space(3))) uint32_t *)&sh.ctl[BV_FC_OV_TAIL] = base + (uint32_t)__popcll(m);
        *(volatile __attribute__((address_space(3))) uint32_t *)&sh.ctl[BV_FC_OV_LOCK] = 0u;
    }
}

// four candidates, one per group of 16 lanes: positions first .. first + n - 1 of queue q
// A variant site enters the variant queue BETWEEN the two phases of its solve (bv_solver16.h): its rank-sum row needs the LRT's
// alleles only, and the jobs behind a workgroup's last pass-1 row are the end of the launch -- with the push behind phase 2 (until
// round 6) the last rows waited for the QUAL and the strand-bias tests of their sites too.  Phase 2 and the row's wave then write
// the same record at the same time: different fields, and the status word by atomic OR on both sides (phase 1 stores it whole).
__device__ __forceinline__ void bv_f_job16(const BvP1ShortArgs &a, BvFusedShared &sh, BvFusedSolver &v, uint32_t B0, uint32_t *q, uint32_t first,
                                           uint32_t n, int lane) {
    const int grp = lane >> 4, gl = lane & 15;
    uint32_t *scratch = v.grp + grp * BV_G16_GRP_WORDS;
    bool variant = false, live = false;
    uint32_t site = 0, pL = 0, pn12 = 0, plut = 0, nb = 0, badq = 0;
    const uint32_t *src = a.bins;
    BvG16Lrt pre;
    pre.status = 0; pre.aw0 = 0; pre.aw1 = 0; pre.chi2 = 0.; pre.ref = 4;
#ifdef BV_PHASE_DEBUG
    uint32_t jp_t_ = (uint32_t)__builtin_amdgcn_s_memtime(), jp_[6] = {0, 0, 0, 0, 0, 0};
#define BV_JP(i) do { const uint32_t n_ = (uint32_t)__builtin_amdgcn_s_memtime(); jp_[i] += n_ - jp_t_; jp_t_ = n_; } while (0)
#else
#define BV_JP(i)
#endif
    if ((uint32_t)grp < n) site = bv_f_take(q, first + (uint32_t)grp);
    if ((uint32_t)grp < n && site == BV_F_EMPTY) {
        if (gl == 0) atomicOr(&a.counters[BV_CTR_TIMEOUT], BV_TMO_TAKE);
    } else if ((uint32_t)grp < n) {
        live = true;
        src = a.bins + (size_t)site * BV_S_BIN_STRIDE;
        uint4 s0, s1, s2;
        bv_load3_l2(&a.summ[site], s0, s1, s2);
        BvG16Bins B;
        uint32_t depth[4] = {s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w};
        const uint32_t total = depth[0] + depth[1] + depth[2] + depth[3];
        nb = s2.x;
        badq = (s2.y & BV_SUM_BADQ) ? 1u : 0u;
        B.hit = sh.tab_hit; B.miss = sh.tab_miss; B.loghit = sh.tab_loghit; B.logmiss = sh.tab_logmiss;
        B.pm = reinterpret_cast<double *>(scratch) + gl;
#pragma unroll
        for (int s = 0; s < BV_G16_SLOTS; ++s) {
            const uint32_t i = (uint32_t)(s * 16 + gl);
            B.w[s] = i < nb ? src[i] : 0u;
        }
#ifdef BV_PHASE_DEBUG
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        BV_JP(0);  // the entry, the summary's and the bins' loads
        variant = bv_site_lrt_g16(v.sa, site, depth, total, badq, B, scratch, lane, &pre);
        if (variant && v.fuse2) bv_f_p2_facts(pre.ref, depth, pre.aw0, pre.aw1, pL, pn12, plut);
    }
    BV_JP(1);  // phase 1: the LRT and the record's first version
    const unsigned long long vm = __ballot(variant && gl == 0);
    if (variant && gl == 0) v.vl[v.n_vl + (uint32_t)__popcll(vm & ((1ull << lane) - 1ull))] = site;
    v.n_vl += (uint32_t)__popcll(vm);
    if (v.fuse2 && vm != 0ull) {
        // the records' first versions are complete (the rank sums' waves add to them)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bv_f_push_variants(a, sh, B0, variant && gl == 0, site, pL, pn12, plut, lane);
    }
    BV_JP(2);  // the record's stores complete, the variant sites queued
    if (live) {
        BvSiteSums S;
        // phase 2 needs the strand totals and, for the rank sum, the bins again (registers that would otherwise sit through the
        // EMs).  ONE trip: the bins' loads are issued first, and the summary load's own s_waitcnt vmcnt(0) covers them and the
        // stores of the record's first version, which phase 2 patches
        uint32_t w2[BV_G16_SLOTS];
#pragma unroll
        for (int s = 0; s < BV_G16_SLOTS; ++s) {
            const uint32_t i = (uint32_t)(s * 16 + gl);
            w2[s] = (variant && i < nb) ? src[i] : 0u;
        }
        uint4 s0, s1, s2;
        bv_load3_l2(&a.summ[site], s0, s1, s2);
        S.fwd[0] = s0.x; S.fwd[1] = s0.y; S.fwd[2] = s0.z; S.fwd[3] = s0.w;
        S.rev[0] = s1.x; S.rev[1] = s1.y; S.rev[2] = s1.z; S.rev[3] = s1.w;
        S.q0_mask = 0; S.nb = nb; S.badq = badq;
        BV_JP(3);  // phase 2's loads
        bv_site_tail_g16(v.sa, site, S, src, nb, scratch, lane, &pre, w2);
    }
    BV_JP(4);  // phase 2: rank sum, QUAL, strand-bias tests
    if (v.n_vl > 56u) bv_f_flush_vl(a, v, lane);
#ifdef BV_PHASE_DEBUG
    if (lane == 0) for (int i = 0; i < 5; ++i) atomicAdd(&a.counters[BV_CTR_WORDS + 4250 + i], jp_[i] >> 4);
#endif
#undef BV_JP
}
// one candidate that needs the wave solver (shallow site: ordered replay; phred-0 calls; more than 128 bins; min_af <= 0)
__device__ __forceinline__ void bv_f_job_hard(const BvP1ShortArgs &a, BvFusedShared &sh, BvFusedSolver &v, uint32_t B0, uint32_t site, int lane) {
    uint32_t *bin_code = v.big->w.raw, *bin_cnt = bin_code + BV_SLOTS * BV_WAVE, *hq = bin_code + 2 * BV_SLOTS * BV_WAVE;
    BvSolverScratch *sv = &v.big->w.sc;
    constexpr int REC_WORDS = (int)(sizeof(bv_site_result) / 4);
    uint4 s0, s1, s2;
    bv_load3_l2(&a.summ[site], s0, s1, s2);
    BvSiteSums S;
    S.fwd[0] = s0.x; S.fwd[1] = s0.y; S.fwd[2] = s0.z; S.fwd[3] = s0.w;
    S.rev[0] = s1.x; S.rev[1] = s1.y; S.rev[2] = s1.z; S.rev[3] = s1.w;
    const uint32_t sm_nb = s2.x;
    S.q0_mask = s2.y & BV_SUM_Q0_MASK;
    S.badq = (s2.y & BV_SUM_BADQ) ? 1u : 0u;
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
    for (uint32_t i0 = 0; i0 < sm_nb; i0 += BV_WAVE) {
        const uint32_t i = i0 + (uint32_t)lane;
        const bool have = i < sm_nb;
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
        v.sa.bs = ch->bs[sg]; v.sa.q = ch->q[sg];
    }
    if (bv_site_solve<false, BvHqMerged, true>(v.sa, site, S, bin_code, bin_cnt, H, sv, sh.tab_hit, sh.tab_miss, lane)) {
        if (lane == 0) v.vl[v.n_vl] = site;
        ++v.n_vl;
        if (v.fuse2) {
            // the facts from the record as stored (read back through the L2: its stores are complete behind the wait)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint4 r0, r1, r2;
            bv_load3_l2(&a.out[site], r0, r1, r2);  // depth[4]; total, status, cvg_sb[0..1]; cvg_sb[2..3], cvg_fs
            const uint32_t *rw = reinterpret_cast<const uint32_t *>(&a.out[site].n_alt);
            const uint32_t aw0 = __builtin_nontemporal_load(rw), aw1 = __builtin_nontemporal_load(rw + 1);
            const uint32_t depth[4] = {r0.x, r0.y, r0.z, r0.w};
            int ref = v.sa.ref_base[site];
            if (ref > 4) ref = 4;
            uint32_t pL, pn12, plut;
            bv_f_p2_facts(ref, depth, aw0, aw1, pL, pn12, plut);
            bv_f_push_variants(a, sh, B0, lane == 0, site, pL, pn12, plut, lane);
        }
        if (v.n_vl > 56u) bv_f_flush_vl(a, v, lane);
    }
    bv_lrt_sync<0>();
}

// One unit of solver work, in this order: a job of candidates with three or four active bases (the longest jobs), a job of
// the others, (waves with the big scratch, once every streaming wave is past its last pass-1 row) a wave-solver candidate,
// a block of 64 non-candidate sites that every streaming wave has passed.  Returns 1: did something; 0: nothing to do right
// now; 2: nothing left to do, ever (no streaming wave will publish another pass-1 row, queues and blocks are empty -- jobs of
// other waves may still be running).
__device__ __forceinline__ int bv_f_solver_step(const BvP1ShortArgs &a, BvFusedShared &sh, BvFusedSolver &v, uint32_t B0, uint32_t B1,
                                                     int lane) {
    const uint32_t n_blocks = (B1 - B0 + 63u) >> 6;
    uint32_t first, n = 0;
    // (read before everything else: once every streaming wave is done, every block is ready and no candidate queue grows any more)
    const uint32_t n_done = bv_f_lds_read_u(&sh.ctl[BV_FC_NDONE]);
    // While rows are still streaming only FULL jobs are taken (four sites): a wave that runs off with the one candidate
    // that has just arrived spends a whole job on it, and the queue behind it grows -- the solver waves have ~60 % of the
    // streaming time's worth of work when every job is full.
    const uint32_t least = n_done == (uint32_t)BV_F_NS ? 1u : BV_F_MIN_JOB;
    // Is there anything to claim at all?  Only then is the wave counted in BUSY: a wave that merely polls must not hold the
    // count up -- the streaming waves wait for BUSY == 0 to know that no variant row is still to come (bv_f_no_row_ever), and
    // twelve waves polling through an unconditional count left it non-zero nearly all the time: a launch whose workgroups
    // all ran dry together (8 sites each) ended after SECONDS, whenever all counts happened to be down at once (round 5).
    const bool claimable = bv_f_lds_read_u(&sh.ctl[BV_FC_Q3_TAIL]) - bv_f_lds_read_u(&sh.ctl[BV_FC_Q3_HEAD]) >= least ||
                           bv_f_lds_read_u(&sh.ctl[BV_FC_Q2_TAIL]) - bv_f_lds_read_u(&sh.ctl[BV_FC_Q2_HEAD]) >= least ||
                           (v.big != nullptr && n_done == (uint32_t)BV_F_NS &&
                            bv_f_lds_read_u(&sh.ctl[BV_FC_QH_TAIL]) != bv_f_lds_read_u(&sh.ctl[BV_FC_QH_HEAD]));
    if (claimable) {
        // a job in flight is counted BEFORE its entries leave the queue (a wave that finds the queues empty and no job
        // counted knows that no variant site is still to come)
        if (lane == 0) atomicAdd(&sh.ctl[BV_FC_BUSY], 1u);
        uint32_t *q = sh.q3;
        n = bv_f_claim(sh.ctl, BV_FC_Q3_TAIL, BV_FC_Q3_HEAD, least, 4u, first, lane);
        if (n == 0u) { q = sh.q2; n = bv_f_claim(sh.ctl, BV_FC_Q2_TAIL, BV_FC_Q2_HEAD, least, 4u, first, lane); }
#ifdef BV_PHASE_DEBUG
        const uint32_t jt0_ = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
        if (n != 0u) bv_f_job16(a, sh, v, B0, q, first, n, lane);   // (ONE call site: the solver is ~50 KB of code)
#ifdef BV_PHASE_DEBUG
        if (n != 0u && lane == 0) {
            const uint32_t dt_ = ((uint32_t)__builtin_amdgcn_s_memtime() - jt0_) >> 4, late_ = n_done == (uint32_t)BV_F_NS ? 4u : 0u;
            atomicAdd(&a.counters[BV_CTR_WORDS + 4212 + late_], dt_); atomicAdd(&a.counters[BV_CTR_WORDS + 4213 + late_], 1u);
            atomicAdd(&a.counters[BV_CTR_WORDS + 4214 + late_], n); if (q == sh.q3) atomicAdd(&a.counters[BV_CTR_WORDS + 4215 + late_], 1u);
        }
#endif
        if (n == 0u && v.big != nullptr && n_done == (uint32_t)BV_F_NS &&
            (n = bv_f_claim(sh.ctl, BV_FC_QH_TAIL, BV_FC_QH_HEAD, 1u, 1u, first, lane)) != 0u) {
            // (every streaming wave ran s_waitcnt vmcnt(0) behind its last list entry before it counted itself done)
            const uint32_t site = (uint32_t)__builtin_amdgcn_readfirstlane((int)__builtin_nontemporal_load(&a.cand_list[B0 + first]));
            bv_f_job_hard(a, sh, v, B0, site, lane);
        }
        if (lane == 0) atomicSub(&sh.ctl[BV_FC_BUSY], 1u);
#ifdef BV_TEAM_DEBUG  /* when the workgroup's last solver job ended */
        if (n != 0u && lane == 0) atomicMax(&a.counters[BV_CTR_WORDS + (blockIdx.x < 512u ? blockIdx.x : 511u) * 8u + 5u], (uint32_t)__builtin_amdgcn_s_memrealtime());
#endif
        if (n != 0u) return 1;
    }
    {
        const uint32_t bh = bv_f_lds_read_u(&sh.ctl[BV_FC_BLK_HEAD]);
        if (bh < n_blocks) {
            uint32_t ready = n_blocks;
            if (n_done < (uint32_t)BV_F_NS) {
                uint32_t m = 0xFFFFFFFFu;
#pragma unroll
                for (int w = 0; w < BV_F_NS; ++w) {
                    const uint32_t p = bv_f_lds_read_u(&sh.pub[w]);
                    m = p < m ? p : m;
                }
                ready = m >= B1 ? n_blocks : ((m - B0) >> 6);
            }
            if (bh < ready) {
                uint32_t old = 0;
                if (lane == 0) old = atomicCAS(&sh.ctl[BV_FC_BLK_HEAD], bh, bh + 1u);
                old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
                if (old == bh) {
                    const uint32_t site = B0 + bh * 64u + (uint32_t)lane;
                    if (site < B1) bv_p1s_simple_site<true>(a, v.sa.lnfact, site);
                }
                return 1;
            }
        }
    }
    if (n_done == (uint32_t)BV_F_NS) {
        // nothing was claimable a moment ago and no producer is left: done, unless a queue got its last entries in between
        const bool q_left = bv_f_lds_read_u(&sh.ctl[BV_FC_Q3_TAIL]) != bv_f_lds_read_u(&sh.ctl[BV_FC_Q3_HEAD]) ||
                            bv_f_lds_read_u(&sh.ctl[BV_FC_Q2_TAIL]) != bv_f_lds_read_u(&sh.ctl[BV_FC_Q2_HEAD]) ||
                            (v.big != nullptr && bv_f_lds_read_u(&sh.ctl[BV_FC_QH_TAIL]) != bv_f_lds_read_u(&sh.ctl[BV_FC_QH_HEAD])) ||
                            bv_f_lds_read_u(&sh.ctl[BV_FC_BLK_HEAD]) < n_blocks;
        if (!q_left && v.n_vl) bv_f_flush_vl(a, v, lane);
        return q_left ? 1 : 2;
    }
    return 0;
}
// ------------------------------------------------------------------------------ the streaming side
// "No variant row will ever come again" (FUSE2), read in THIS order: (1) every streaming wave has published its last pass-1
// row -- the candidate queues only shrink from here on; (2) the candidate queues are empty -- whoever took their last entries
// had counted itself in BUSY before it claimed them (bv_f_solver_step); (3) no job is counted -- those jobs are over, and a job
// pushes its variant sites before it leaves the count; (4) the variant queue is empty -- nothing can refill it.  (BUSY read
// before the queues would leave a window: BUSY == 0, then a solver counts itself and claims the last entries, then the queues
// and the variant queue are all seen empty while that job's variant rows are still to come.)
__device__ __forceinline__ bool bv_f_no_row_ever(const uint32_t *ctl) {
    if (bv_f_lds_read_u(&ctl[BV_FC_NDONE]) != (uint32_t)BV_F_NS) return false;
    if (bv_f_lds_read_u(&ctl[BV_FC_Q3_TAIL]) != bv_f_lds_read_u(&ctl[BV_FC_Q3_HEAD]) ||
        bv_f_lds_read_u(&ctl[BV_FC_Q2_TAIL]) != bv_f_lds_read_u(&ctl[BV_FC_Q2_HEAD]) ||
        bv_f_lds_read_u(&ctl[BV_FC_QH_TAIL]) != bv_f_lds_read_u(&ctl[BV_FC_QH_HEAD]))
        return false;
    if (bv_f_lds_read_u(&ctl[BV_FC_BUSY]) != 0u) return false;
    return bv_f_lds_read_u(&ctl[BV_FC_QV_TAIL]) == bv_f_lds_read_u(&ctl[BV_FC_QV_HEAD]) &&
           bv_f_lds_read_u(&ctl[BV_FC_OV_TAIL]) == bv_f_lds_read_u(&ctl[BV_FC_OV_HEAD]);
}
typedef volatile __attribute__((address_space(3))) uint32_t bv_lds_vu32;
__device__ __forceinline__ void bv_f_push(bv_lds_vu32 *q, bv_lds_u32 *tail, uint32_t site, uint32_t *counters, bool lose_first, int lane) {
    const uint32_t pos = bv_lds_fetch_add_wave((uint32_t)(uintptr_t)tail, 1u);
    bv_lds_vu32 *e = q + (pos & (BV_F_QCAP - 1u));
    // (a slot still occupied: the solvers are BV_F_QCAP candidates behind -- they never wait for a streaming wave, so this ends)
    uint32_t spins = 0;
    while ((uint32_t)__builtin_amdgcn_readfirstlane((int)*e) != BV_F_EMPTY && spins < BV_F_SPIN_MAX) { __builtin_amdgcn_s_sleep(8); ++spins; }
    if (spins == BV_F_SPIN_MAX) {
        // gave up: the entry that sits there is not overwritten (its consumer may still come), this site is not solved, and the
        // submit fails loudly (the consumer of position `pos` times out in its turn)
        if (lane == 0) atomicOr(&counters[BV_CTR_TIMEOUT], BV_TMO_PUSH);  // (OR, not add: repeated events must not carry into the next field, or wrap the word to 0)
        return;
    }
    // (BV_FLAG_FAULT_LOST_HANDOFF, tests: the first entry of every workgroup's queue is reserved and never written -- the solver
    // wave that claims it must give up after its bounded wait)
    if (lose_first && pos == 0u) return;
    if (lane == 0) *e = site;
}
// publish a pass-1 row whose stores are complete.  Candidates of the 16-lane solver: a place in their queue -- one LDS atomic,
// one LDS write (the slot is free once its last consumer has handed it back empty).  Candidates of the wave solver: a place in
// the workgroup's slice of cand_list in HBM (unbounded: they are taken up only when every streaming wave is past its last
// pass-1 row, see bv_f_solver_step, so no streaming wave ever waits on them); that store is one more in the vmcnt queue than
// the slot waits allow for -- a conservative wait, never a wrong one.
__device__ __forceinline__ void bv_f_publish(const BvP1ShortArgs &a, BvFusedShared &sh, uint32_t B0, uint32_t site, uint32_t kind, int lane) {
    if (kind < 2u) return;
    if (kind == 4u) {
        const uint32_t pos = bv_lds_fetch_add_wave((uint32_t)(uintptr_t)(bv_lds_u32 *)&sh.ctl[BV_FC_QH_TAIL], 1u);
        if (lane == 0) a.cand_list[B0 + pos] = site;  // (a slot per site of the workgroup's range: the list cannot outgrow them)
    } else if (kind == 3u) {
        bv_f_push((bv_lds_vu32 *)sh.q3, (bv_lds_u32 *)&sh.ctl[BV_FC_Q3_TAIL], site, a.counters, (a.flags & BV_FLAG_FAULT_LOST_HANDOFF) != 0u, lane);
    } else {
        bv_f_push((bv_lds_vu32 *)sh.q2, (bv_lds_u32 *)&sh.ctl[BV_FC_Q2_TAIL], site, a.counters, (a.flags & BV_FLAG_FAULT_LOST_HANDOFF) != 0u, lane);
    }
}
// A variant row with a read-position rank beyond the 256-rank window of the fast tally (long reads): the exact window sweeps of
// bv_pass2_sweep.h -- plain loads, as bv_pass2_dma_kernel falls back to them -- into the wave's histogram, the two rank sums
// stored at once.  Rare, and not inlined: the streaming loop keeps its registers; the loads' waits drain the ring.
__device__ __attribute__((noinline)) void bv_f_p2_redo(const uint8_t *bs_, const uint8_t *mapq_, const uint16_t *rpr_, bv_site_result *out_, uint64_t pitch,
                                                       uint32_t n_samples, uint32_t site_, uint32_t lut_, uint32_t n12_, uint32_t hist_lds_) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t site = (uint32_t)__builtin_amdgcn_readfirstlane((int)site_), lut_tag = (uint32_t)__builtin_amdgcn_readfirstlane((int)lut_);
    const uint32_t lut = lut_tag & 0xFFu;  // bit 31: the rank plane is tagged (BV_SLAB_RPR_TAGGED)
    const uint32_t n12 = (uint32_t)__builtin_amdgcn_readfirstlane((int)n12_);
    uint32_t *h = (uint32_t *)(bv_lds_u32 *)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hist_lds_);
    bv_site_result *out = (bv_site_result *)(__attribute__((address_space(1))) bv_site_result *)out_;
    BvPass2Args as;
    as.bs = (const uint8_t *)(__attribute__((address_space(1))) const uint8_t *)bs_;
    as.mapq = (const uint8_t *)(__attribute__((address_space(1))) const uint8_t *)mapq_;
    as.rpr = (const uint16_t *)(__attribute__((address_space(1))) const uint16_t *)rpr_;
    as.q = nullptr; as.group_id = nullptr; as.pitch = pitch; as.n_samples = n_samples; as.n_groups = 0;
    const unsigned long long n1 = n12 & 0xFFFFu, n2 = n12 >> 16;
    uint32_t *hm = h, *hr = h + 512;
    auto zero = [&](uint32_t *p, int words) {
        uint4 *z = reinterpret_cast<uint4 *>(p);
        for (int i = lane; i < words / 4; i += BV_WAVE) z[i] = make_uint4(0, 0, 0, 0);
    };
    zero(h, 4 * 256);
    bv_lrt_sync<0>();
    BvP2Ctx cx;
    cx.hm = hm; cx.hr = hr; cx.hg = nullptr; cx.lut = lut; cx.win_lo = 0; cx.n_groups = 0; cx.maxr = 0; cx.half = false;
    cx.rmask = bv_rpr_rank_mask(lut_tag >> 31);
    bv_p2_sweep<BV_WAVE, true, true, false, 256>(cx, as, site, lane);
    const uint32_t maxr = (uint32_t)bv_wave_max_i32((int)cx.maxr);
    bv_lrt_sync<0>();
    {
        unsigned long long below = 0, twoR = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) twoR += bv_ranksum_window(hm[w * 64 + lane], hm[256 + w * 64 + lane], n1 + n2, below, lane);
        const double ph = bv_ranksum_phred(twoR, n1, n2);
        if (lane == 0) out[site].mq_ranksum = ph;
    }
    unsigned long long below = 0, twoR = 0;
    for (uint32_t win_lo = 0;; win_lo += 256u) {
        const int nblk = (maxr < win_lo + 256u) ? (int)((maxr - win_lo) >> 6) + 1 : 4;
        for (int w = 0; w < nblk; ++w) twoR += bv_ranksum_window(hr[w * 64 + lane], hr[256 + w * 64 + lane], n1 + n2, below, lane);
        if (maxr < win_lo + 256u) break;
        bv_lrt_sync<0>();
        zero(hr, 2 * 256);
        bv_lrt_sync<0>();
        cx.win_lo = win_lo + 256u;
        bv_p2_sweep<BV_WAVE, true, false, false, 256>(cx, as, site, lane);
        bv_lrt_sync<0>();
    }
    const double ph = bv_ranksum_phred(twoR, n1, n2);
    if (lane == 0) {
        out[site].rpr_ranksum = ph;
        atomicOr(&out[site].status, BV_SITE_RANKSUM);
    }
    bv_lrt_sync<0>();
}

// pass-2 results of up to 64 rows of one wave, one row per lane: the two rank sums as exact integers, turned into phred values
// (erfc, log10: scalar work per site) and stored for all of them at once
struct BvFusedStash {
    uint32_t site, n12;
    unsigned long long tw_m, tw_r;
    uint32_t n;  // rows held (wave-uniform)
};
__device__ __forceinline__ void bv_f_stash_flush(const BvP1ShortArgs &a, BvFusedStash &t, int lane) {
    if ((uint32_t)lane < t.n) {
        const unsigned long long n1 = t.n12 & 0xFFFFu, n2 = t.n12 >> 16;
        const double ph_m = bv_ranksum_phred(t.tw_m, n1, n2), ph_r = bv_ranksum_phred(t.tw_r, n1, n2);
        a.out[t.site].mq_ranksum = ph_m;
        a.out[t.site].rpr_ranksum = ph_r;
        atomicOr(&a.out[t.site].status, BV_SITE_RANKSUM);
    }
    t.n = 0;
}

// Streams rows until none is in flight and none can be drawn right now.  `st_io`: the wave's flags that outlive a call (cursor
// exhausted / last pass-1 row published / no row will ever come again).  On return the ring is idle (it can serve as solver
// scratch), no pass-1 row of this wave is unpublished and the pass-2 results of its rows are stored.
//
// NOT inlined.  Inlined into the kernel's loop beside the solver (which wants all of its 168 registers), the allocator parked
// this loop's lane constants in scratch memory: a scratch load behind an s_waitcnt vmcnt(0) -- a drained ring -- per row.  As a
// function of its own it has a register allocation of its own.  What it needs arrives so that everything stays what it is in
// a kernel: the argument block is read from the kernel's own kernarg segment (scalar loads, wave-uniform by construction; its
// pointers are marked as device memory), the LDS block comes as an LDS ADDRESS and is turned back into a pointer here (a generic
// pointer handed across a call would make every access a flat one), the lane number is formed here.
// not_tail_called: clang marks the kernel's call `tail`, and a function with a tail-marked caller keeps the callee-saved
// registers of the AMDGPU convention -- its prologue stored the 60 of them it uses (v40-47, v56-63, ...) to scratch memory,
// 15 KB per wave and call, and the L2 wrote 85 MB of that back per launch.  Without the mark LLVM's inter-procedural register
// allocation applies to this local, non-recursive function: no saves here, the kernel keeps what IT has live across the call
// (next to nothing: see the lane number in the kernel's loop).
#define BV_F_GLOBAL(T, p) ((T *)(__attribute__((address_space(1))) T *)(p))
// P2RING: the ring carries the variant sites' pass-2 rows too (FUSE2 with plain ranks, or BV_FLAG_P2_TAIL_DMA); false: pass-1 rows only
// -- the instance FUSE2 launches of tagged slabs take: without the pass-2 tally in its loop it has no scalar registers spilled there.
template <bool FUSE2, bool P2RING = FUSE2>
__device__ __attribute__((noinline, not_tail_called)) uint32_t bv_f_stream_until_idle(uint32_t ka_lo_, uint32_t ka_hi_, uint32_t sh_lds_, uint32_t wave_, uint32_t B0_,
                                                                     uint32_t B1_, uint32_t st_in_) {
    const uint32_t sh_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_lds_);
    const int wave = __builtin_amdgcn_readfirstlane((int)wave_);
    const uint32_t B0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)B0_), B1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)B1_);
    uint32_t st_io = (uint32_t)__builtin_amdgcn_readfirstlane((int)st_in_);
    BvFusedShared &sh = *(BvFusedShared *)(__attribute__((address_space(3))) BvFusedShared *)(uintptr_t)sh_lds;
    BvP1ShortArgs a;
    {
        // (the kernel hands its kernarg segment pointer over: the intrinsic itself read null in this function)
        const uint64_t kp = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)ka_hi_) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)ka_lo_);
        const __attribute__((address_space(4))) BvP1ShortArgs *ka = (const __attribute__((address_space(4))) BvP1ShortArgs *)(uintptr_t)kp;
        a.bs = BV_F_GLOBAL(const uint8_t, ka->bs); a.q = BV_F_GLOBAL(const uint8_t, ka->q); a.ref_base = BV_F_GLOBAL(const uint8_t, ka->ref_base);
        a.pitch = ka->pitch; a.n_sites = ka->n_sites; a.n_samples = ka->n_samples; a.flags = ka->flags; a.n_cu = ka->n_cu;
        a.min_af = ka->min_af; a.tables = BV_F_GLOBAL(const BvTables, ka->tables); a.out = BV_F_GLOBAL(bv_site_result, ka->out);
        a.var_list = BV_F_GLOBAL(uint32_t, ka->var_list); a.counters = BV_F_GLOBAL(uint32_t, ka->counters);
        a.summ = BV_F_GLOBAL(BvSiteSummary, ka->summ); a.bins = BV_F_GLOBAL(uint32_t, ka->bins);
        a.cand_list = BV_F_GLOBAL(uint32_t, ka->cand_list); a.easy_list = BV_F_GLOBAL(uint32_t, ka->easy_list);
        a.easy3_list = BV_F_GLOBAL(uint32_t, ka->easy3_list); a.ch = BV_F_GLOBAL(const BvChain, ka->ch);
        a.mapq = BV_F_GLOBAL(const uint8_t, ka->mapq); a.rpr = BV_F_GLOBAL(const uint16_t, ka->rpr);
        a.ovf = BV_F_GLOBAL(uint32_t, ka->ovf);
        a.rpr_tag = ka->rpr_tag;
    }
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t *hist = sh.hist[wave];
    uint32_t *stage = sh.stage[wave];
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(bv_lds_u32 *)sh.ring[wave].slot[0]);
    const uint32_t *ring = sh.ring[wave].slot[0];
    const uint32_t cursor_lds = (uint32_t)(uintptr_t)(bv_lds_u32 *)&sh.ctl[BV_FC_CURSOR];
    const uint32_t qvhead_lds = (uint32_t)(uintptr_t)(bv_lds_u32 *)&sh.ctl[BV_FC_QV_HEAD];
    const uint32_t ovhead_lds = (uint32_t)(uintptr_t)(bv_lds_u32 *)&sh.ctl[BV_FC_OV_HEAD];
    // ---- the geometry of a row (the same for every row of a kind)
    const uint32_t n_chunks = (a.n_samples + 15u) >> 4;  // 16-byte chunks of a plane's row
    const int tail = (int)(a.n_samples & 15u);
    // pass-1 rows: slots of 128 chunks (2 KiB of calls + 2 KiB of phreds)
    const uint32_t n_slots1 = (n_chunks + 127u) >> 7;
    const uint32_t last1 = n_chunks - (n_slots1 - 1u) * 128u;  // chunks of a row's last slot that lie inside the row: 1 .. 128
    const uint32_t va = (uint32_t)lane * 16u, vb = va + 1024u;
    const uint32_t vbl = last1 > 64u ? vb : va;
    const unsigned long long mA1 = last1 >= 64u ? ~0ull : ((1ull << last1) - 1ull);
    const unsigned long long mB1 = last1 > 64u ? (last1 >= 128u ? ~0ull : ((1ull << (last1 - 64u)) - 1ull)) : 1ull;
    // pass-2 rows: slots of 64 chunks (1 KiB of calls, 1 KiB of mapq, 2 KiB of ranks: 32 bytes per lane in two pieces)
    const uint32_t n_slots2 = (n_chunks + 63u) >> 6;
    const uint32_t last2 = n_chunks - (n_slots2 - 1u) * 64u;   // 1 .. 64
    // The 2 KiB of ranks arrive as two CONTIGUOUS KiB (lane l: bytes 16 l .. 16 l + 15 of each), not as the lane's own 32 bytes in two
    // halves: a piece of 64 x 16 bytes at a stride of 32 touches sixteen 128-byte lines instead of eight, and its twin the same
    // sixteen again -- in the launch's last phase, where only such rows stream, requesting a slot took 1,260 cycles against 420
    // for a pass-1 slot and the slot then came late (1,870 cycles of waiting per slot; round 6, -DBV_PHASE_DEBUG).  The lane's own
    // 32 bytes are put together where the slot is read: two ds_read_b128 at a stride of 32.
    const unsigned long long mA2 = last2 >= 64u ? ~0ull : ((1ull << last2) - 1ull);
    // a row's last slot: the lanes of the two rank pieces that lie inside the row (2 x last2 chunks of 16 bytes)
    const unsigned long long mR0 = last2 >= 32u ? ~0ull : ((1ull << (2u * last2)) - 1ull);
    const unsigned long long mR1 = last2 > 32u ? (last2 >= 64u ? ~0ull : ((1ull << (2u * last2 - 64u)) - 1ull)) : 1ull;
    const uint32_t vrl = last2 > 32u ? vb : va;  // (a second piece wholly past the row's end: lane 0 alone re-reads valid bytes)
    // the last slot's cells past the row's end are forced to 'N': per lane and dword, the mask of the bytes that stay
    uint32_t keepA[4], keepB[4], keep2[4];
    {
        const uint32_t cA = (uint32_t)lane, cB = 64u + (uint32_t)lane;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int partial = tail ? tail - 4 * d : 4;
            keepA[d] = cA >= last1 ? 0u : ((cA == last1 - 1u) ? bv_f_keep_mask(partial) : 0xFFFFFFFFu);
            keepB[d] = cB >= last1 ? 0u : ((cB == last1 - 1u) ? bv_f_keep_mask(partial) : 0xFFFFFFFFu);
            keep2[d] = cA >= last2 ? 0u : ((cA == last2 - 1u) ? bv_f_keep_mask(partial) : 0xFFFFFFFFu);
        }
    }
    uint32_t one;
    asm volatile("v_mov_b32 %0, 1" : "=v"(one));
    // the tagged rank layout (BV_SLAB_RPR_TAGGED): a pass-2 row is mapq + ranks, the class of a cell comes from its rank word
    const bool tag = P2RING && a.rpr_tag != 0u;
    const uint32_t hi_mask = bv_rpr_hi_mask(tag ? 1u : 0u);

    // ---- prefetch side: the next slot to request
    const uint8_t *p0 = a.bs, *p1 = a.q, *p2 = a.q;  // pass 1: calls, phreds; pass 2: calls, mapq, ranks
    uint32_t p_left = 0, p_kind = 0;    // slots of the prefetch row still to request; its kind
    uint32_t ring_w = 0, ring_r = 0, inflight = 0;
    // rows drawn: the one being tallied and the one after it (the prefetch runs at most one row ahead: every row has >= BV_F_K slots)
    // x / y: pass 1: reference base / -; pass 2: class table / n_ref | n_alt << 16; z: pass 2: the sweeps' 2-bit table
    uint32_t c_site = 0, c_kind = 0, c_x = 0, c_y = 0, c_z = 0, n_site = 0, n_kind = 0, n_x = 0, n_y = 0, n_z = 0;
    uint32_t st = st_io;
    BV_PH_DECL;
    constexpr uint32_t P_DONE = BV_FS_P_DONE, C_HAVE = 2u, N_HAVE = 4u, CUR_DONE = BV_FS_CUR_DONE, P1_FIN = BV_FS_P1_FIN;
    auto issue = [&]() __attribute__((always_inline)) {
        if (p_left == 0u) {
            if (st & P_DONE) return;
            uint32_t s = 0, kind = 0, x = 0, y = 0, z = 0;
            // the next row: a pass-1 row while the cursor has any, then the variant sites' pass-2 rows.  (Until round 6 a wave took a
            // variant row first whenever 64 of them were waiting.  Rows of the two kinds mixed cost more than the same rows one
            // kind after the other -- thresholds of 2 / 8 / 24 measured -9 / -9 / -5 % at 100,000 sites, "never" +2.5 % at 524,288,
            // where the queue does reach 64 --, and nothing needs it: a full variant queue spills to the overflow list.)
            if (!(st & CUR_DONE)) {
                const uint32_t c = bv_lds_fetch_add_wave(cursor_lds, 1u);
                if (c < B1 - B0) { s = B0 + c; kind = BV_FK_P1; x = bv_f_ref_scalar(a.ref_base, s); }
                else st |= CUR_DONE;
            }
            // (a wave past its pass-1 rows streams what variant rows there are; one that solved first instead measured 168
            // against 175 M sites/s: HBM idles while twelve waves solve)
            if (P2RING && kind == 0u) {
                const uint32_t h = bv_f_lds_read_u(&sh.ctl[BV_FC_QV_HEAD]);
                if (h != bv_f_lds_read_u(&sh.ctl[BV_FC_QV_TAIL]) && bv_f_lds_cas_wave(qvhead_lds, h, h + 1u) == h) {
                    const uint32_t *e = sh.qv[h & (BV_F_QVCAP - 1u)];
                    for (uint32_t spins = 0; (s = bv_f_lds_read_u(&e[0])) == BV_F_EMPTY && spins < BV_F_SPIN_MAX; ++spins) __builtin_amdgcn_s_sleep(4);
                    x = bv_f_lds_read_u(&e[1]); y = bv_f_lds_read_u(&e[2]); z = bv_f_lds_read_u(&e[3]);
                    if (lane == 0) *(bv_lds_vu32 *)&e[0] = BV_F_EMPTY;
                    if (s == BV_F_EMPTY) { if (lane == 0) atomicOr(&a.counters[BV_CTR_TIMEOUT], BV_TMO_TAKE_VARIANT); }  // (timed out: no row, flagged)
                    else kind = BV_FK_P2;
                }
            }
            if (P2RING && kind == 0u) {
                // ... or from the overflow list (rare: more variant sites waiting than the LDS queue takes).  Its entries are in
                // HBM: one vector load, the ring drains
                const uint32_t oh = bv_f_lds_read_u(&sh.ctl[BV_FC_OV_HEAD]);
                if (oh != bv_f_lds_read_u(&sh.ctl[BV_FC_OV_TAIL]) && bv_f_lds_cas_wave(ovhead_lds, oh, oh + 1u) == oh) {
                    bv_u32x4 e;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(e) : "v"(a.ovf + 4u * (size_t)(B0 + oh)) : "memory");
                    s = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x); x = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
                    y = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.z); z = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.w);
                    kind = BV_FK_P2;
                }
            }
            if (kind == 0u) {
                // no row right now.  None ever again: the cursor is exhausted, every streaming wave has published its last
                // pass-1 row, no candidate waits, no solver job runs (each is counted before its entries leave their
                // queue, until its variant sites are in theirs), and the variant queue is empty.
                if ((st & CUR_DONE) && (!FUSE2 || bv_f_no_row_ever(sh.ctl))) st |= P_DONE;
                return;
            }
            const uint64_t off = (uint64_t)s * a.pitch;
            // (a chained launch: the planes of the segment that holds the site, biased so that the global site number indexes
            // them -- read through the constant address space: scalar loads, outside the counted vmcnt queue)
            const uint8_t *r_bs = a.bs, *r_q = a.q, *r_mq = a.mapq, *r_rp = reinterpret_cast<const uint8_t *>(a.rpr);
            if (a.ch != nullptr) {
                const BvChainC ch = bv_chain_const(a.ch);
                const uint32_t sg = bv_chain_seg(ch, s);
                r_bs = ch->bs[sg]; r_q = ch->q[sg]; r_mq = ch->mapq[sg]; r_rp = reinterpret_cast<const uint8_t *>(ch->rpr[sg]);
            }
            p0 = bv_uniform_ptr(r_bs + off);
            if (kind == BV_FK_P1) { p1 = bv_uniform_ptr(r_q + off); p_left = n_slots1; }
            else { p1 = bv_uniform_ptr(r_mq + off); p2 = bv_uniform_ptr(r_rp + 2u * off); p_left = n_slots2; }
            p_kind = kind;
            if (!(st & C_HAVE)) { c_site = s; c_kind = kind; c_x = x; c_y = y; c_z = z; st |= C_HAVE; }
            else { n_site = s; n_kind = kind; n_x = x; n_y = y; n_z = z; st |= N_HAVE; }
        }
        BV_PH(10);
#ifdef BV_PHASE_DEBUG
        if (ring_w == 0u) ph_i0_ = (uint32_t)__builtin_amdgcn_s_memtime();  // (ring position 0 only: one in K slots is timed)
#endif
        const uint32_t d0 = ring_lds + ring_w * (BV_F_SLOT_WORDS * 4u);
        if (!P2RING || p_kind == BV_FK_P1) {
            if (p_left > 1u) { bv_f_glds4(d0, p0, va, p0, vb, p1, va, p1, vb); p0 += 2048; p1 += 2048; }
            else bv_f_glds4_masked(d0, p0, va, p0, vbl, p1, va, p1, vbl, mA1, mB1);
        } else if (tag) {
            if (p_left > 1u) { bv_f_glds4_tag(d0, p1, va, p2, va, vb); p1 += 1024; p2 += 2048; }
            else bv_f_glds4_m4(d0, p1, va, 1ull, p1, va, mA2, p2, va, mR0, p2, vrl, mR1);
        } else {
            if (p_left > 1u) { bv_f_glds4(d0, p0, va, p1, va, p2, va, p2, vb); p0 += 1024; p1 += 1024; p2 += 2048; }
            else bv_f_glds4_m4(d0, p0, va, mA2, p1, va, mA2, p2, va, mR0, p2, vrl, mR1);
        }
        --p_left;
        ring_w = (ring_w + 1u == (uint32_t)BV_F_K) ? 0u : ring_w + 1u;
        ++inflight;
    };

    // the previous pass-1 row of this wave: published once its stores are known to be complete
    uint32_t prev_site = 0, prev_kind = 0;  // kind 0: none; 1: not a candidate; 2 / 3: queue q2 / q3; 4: wave solver
    uint32_t wsel = 0;                      // stores of the previous row still to be allowed for in the slot waits: 0 none / unknown, 1, 3
    BvFusedStash stash;
    stash.site = 0; stash.n12 = 0; stash.tw_m = 0; stash.tw_r = 0; stash.n = 0;
    // the end of this wave's pass-1 rows: its last one published, the wave counted in NDONE
    auto finish_p1 = [&]() __attribute__((always_inline)) {
        if (prev_kind != 0u) {
            if (inflight == (uint32_t)BV_F_K) asm volatile(BV_F_W_ALL ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bv_f_publish(a, sh, B0, prev_site, prev_kind, lane);
            if (prev_kind == 4u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (its cand_list entry; rare)
            prev_kind = 0;
        }
        if (lane == 0) {
            sh.pub[wave] = 0xFFFFFFFFu;
#ifdef BV_TEAM_DEBUG  /* per workgroup: first / last streaming wave past its pass-1 rows, and what was left to solve then */
            const uint32_t now = (uint32_t)__builtin_amdgcn_s_memrealtime();
            uint32_t *dbg_ = a.counters + BV_CTR_WORDS + (blockIdx.x < 512u ? blockIdx.x : 511u) * 8u;
            if (atomicAdd(&sh.ctl[BV_FC_NDONE], 1u) == 0u) dbg_[1] = now;
            else if (sh.ctl[BV_FC_NDONE] == (uint32_t)BV_F_NS) {
                dbg_[2] = now;
                dbg_[4] = (sh.ctl[BV_FC_Q3_TAIL] - sh.ctl[BV_FC_Q3_HEAD]) | ((sh.ctl[BV_FC_Q2_TAIL] - sh.ctl[BV_FC_Q2_HEAD]) << 16);
                dbg_[6] = FUSE2 ? sh.ctl[BV_FC_QV_TAIL] - sh.ctl[BV_FC_QV_HEAD] : ((B1 - B0 + 63u) >> 6) - sh.ctl[BV_FC_BLK_HEAD];
            }
#else
            atomicAdd(&sh.ctl[BV_FC_NDONE], 1u);
#endif
        }
        st |= P1_FIN;
    };

#pragma unroll 1
    for (int k = 0; k < BV_F_K; ++k) issue();
    BV_PH(2);
#pragma unroll 1
    while (st & C_HAVE) {
        const uint32_t site = c_site;
        const bool is_p1 = !P2RING || c_kind == BV_FK_P1;
        const uint32_t n_slots = is_p1 ? n_slots1 : n_slots2;
        uint32_t hi_acc = 0, dom = BV_DOM_NONE;
#pragma unroll 1
        for (uint32_t j = 0; j < n_slots; ++j) {
            // The oldest slot in flight has landed once at most 8 younger loads are outstanding -- plus, for the three waits
            // that follow a pass-1 row's epilogue, that row's S stores, which are younger than the slot waited for (vmcnt
            // counts in issue order).  Unknown S, or fewer than K slots in flight: the conservative wait.
            if (inflight != (uint32_t)BV_F_K) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (j >= (uint32_t)BV_F_K || wsel == 0u) asm volatile(BV_F_W_OLDEST ::: "memory");
            else if (wsel == 1u) asm volatile(BV_F_W_OLDEST_1 ::: "memory");
            else asm volatile(BV_F_W_OLDEST_3 ::: "memory");
            BV_PH(0); BV_PH_COUNT(7);
#ifdef BV_PHASE_DEBUG
            if (ring_r == 0u) { if (st & BV_FS_P1_FIN) { ph_i2_ += ph_t_ - ph_i0_; ++ph_n2_; } else { ph_i1_ += ph_t_ - ph_i0_; ++ph_n1_; } }  // requested -> found landed
#endif
            const uint32_t *rs = ring + ring_r * BV_F_SLOT_WORDS + lane * 4;
            // (a pass-2 slot: the lane's 16 ranks are 32 contiguous bytes of the slot's second half)
            const uint32_t *rs2 = (!P2RING || c_kind == BV_FK_P1) ? rs + 512 : rs + 512 + lane * 4;
            const uint32_t o3 = (!P2RING || c_kind == BV_FK_P1) ? 256u : 4u;
            bv_u32x4 w0 = *reinterpret_cast<const bv_u32x4 *>(rs);
            bv_u32x4 w1 = *reinterpret_cast<const bv_u32x4 *>(rs + 256);
            bv_u32x4 w2 = *reinterpret_cast<const bv_u32x4 *>(rs2);
            bv_u32x4 w3 = *reinterpret_cast<const bv_u32x4 *>(rs2 + o3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // in registers: the slot may be refilled
            ring_r = (ring_r + 1u == (uint32_t)BV_F_K) ? 0u : ring_r + 1u;
            --inflight;
            BV_PH(1);
            issue();
            BV_PH(2);
            const uint32_t N4 = 0x08080808u;
            if (is_p1) {
                // w0 / w1: calls of the slot's first / second KiB; w2 / w3: their phreds
                if (j + 1u == n_slots) {
                    w0.x = (w0.x & keepA[0]) | (N4 & ~keepA[0]); w0.y = (w0.y & keepA[1]) | (N4 & ~keepA[1]);
                    w0.z = (w0.z & keepA[2]) | (N4 & ~keepA[2]); w0.w = (w0.w & keepA[3]) | (N4 & ~keepA[3]);
                    w1.x = (w1.x & keepB[0]) | (N4 & ~keepB[0]); w1.y = (w1.y & keepB[1]) | (N4 & ~keepB[1]);
                    w1.z = (w1.z & keepB[2]) | (N4 & ~keepB[2]); w1.w = (w1.w & keepB[3]) | (N4 & ~keepB[3]);
                    // (stale phred bytes of lanes that loaded nothing must not look like invalid input)
                    w2.x &= keepA[0]; w2.y &= keepA[1]; w2.z &= keepA[2]; w2.w &= keepA[3];
                    w3.x &= keepB[0]; w3.y &= keepB[1]; w3.z &= keepB[2]; w3.w &= keepB[3];
                }
                bv_f_tally2(w0, w2, w1, w3, hist, one);
                BV_PH(3);
            } else {
                // w0: calls; w1: mapq; w2 / w3: ranks 0-7 / 8-15 of the lane's 16 cells.  The tally of bv_pass2_dma_kernel
                // (bv_pass2.hip): class bytes by one v_perm per dword, class << 8 | value by one v_perm per cell, "< 0x200"
                // the whole predicate.
                const uint32_t L = c_x;
                uint32_t c0, c1, c2, c3;
                if (tag) {
                    // (w0 is not loaded: the cells' calls are the tags of their rank words)
                    if (j + 1u == n_slots) {
                        if ((uint32_t)lane >= last2) { w2 = bv_u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u}; w3 = w2; }  // (not loaded: stale bytes)
                        else if (tail && (uint32_t)lane == last2 - 1u) {
                            w2.x = bv_p2t_mask_tail(w2.x, tail); w2.y = bv_p2t_mask_tail(w2.y, tail - 2);
                            w2.z = bv_p2t_mask_tail(w2.z, tail - 4); w2.w = bv_p2t_mask_tail(w2.w, tail - 6);
                            w3.x = bv_p2t_mask_tail(w3.x, tail - 8); w3.y = bv_p2t_mask_tail(w3.y, tail - 10);
                            w3.z = bv_p2t_mask_tail(w3.z, tail - 12); w3.w = bv_p2t_mask_tail(w3.w, tail - 14);
                        }
                    }
                    c0 = bv_p2t_class4(L, w2.x, w2.y); c1 = bv_p2t_class4(L, w2.z, w2.w);
                    c2 = bv_p2t_class4(L, w3.x, w3.y); c3 = bv_p2t_class4(L, w3.z, w3.w);
                } else {
                    if (j + 1u == n_slots) {
                        w0.x = (w0.x & keep2[0]) | (N4 & ~keep2[0]); w0.y = (w0.y & keep2[1]) | (N4 & ~keep2[1]);
                        w0.z = (w0.z & keep2[2]) | (N4 & ~keep2[2]); w0.w = (w0.w & keep2[3]) | (N4 & ~keep2[3]);
                        if ((uint32_t)lane >= last2) { w2 = bv_u32x4{0u, 0u, 0u, 0u}; w3 = w2; }  // (not loaded: stale bytes)
                    }
                    c0 = __builtin_amdgcn_perm(L, L, w0.x) ^ 0x80808080u; c1 = __builtin_amdgcn_perm(L, L, w0.y) ^ 0x80808080u;
                    c2 = __builtin_amdgcn_perm(L, L, w0.z) ^ 0x80808080u; c3 = __builtin_amdgcn_perm(L, L, w0.w) ^ 0x80808080u;
                }
                {
                // ranks that do not fit the 256-rank window: remembered, the row is then re-done by the window sweeps
                hi_acc |= (w2.x | w2.y | w2.z | w2.w | w3.x | w3.y | w3.z | w3.w) & hi_mask;
                uint32_t x[16];
                x[0] = bv_p2d_xm<0>(c0, w1.x); x[1] = bv_p2d_xm<1>(c0, w1.x); x[2] = bv_p2d_xm<2>(c0, w1.x); x[3] = bv_p2d_xm<3>(c0, w1.x);
                x[4] = bv_p2d_xm<0>(c1, w1.y); x[5] = bv_p2d_xm<1>(c1, w1.y); x[6] = bv_p2d_xm<2>(c1, w1.y); x[7] = bv_p2d_xm<3>(c1, w1.y);
                x[8] = bv_p2d_xm<0>(c2, w1.z); x[9] = bv_p2d_xm<1>(c2, w1.z); x[10] = bv_p2d_xm<2>(c2, w1.z); x[11] = bv_p2d_xm<3>(c2, w1.z);
                x[12] = bv_p2d_xm<0>(c3, w1.w); x[13] = bv_p2d_xm<1>(c3, w1.w); x[14] = bv_p2d_xm<2>(c3, w1.w); x[15] = bv_p2d_xm<3>(c3, w1.w);
                uint32_t y[16];
                y[0] = bv_p2d_xr<0, 0>(c0, w2.x); y[1] = bv_p2d_xr<1, 1>(c0, w2.x); y[2] = bv_p2d_xr<2, 0>(c0, w2.y); y[3] = bv_p2d_xr<3, 1>(c0, w2.y);
                y[4] = bv_p2d_xr<0, 0>(c1, w2.z); y[5] = bv_p2d_xr<1, 1>(c1, w2.z); y[6] = bv_p2d_xr<2, 0>(c1, w2.w); y[7] = bv_p2d_xr<3, 1>(c1, w2.w);
                y[8] = bv_p2d_xr<0, 0>(c2, w3.x); y[9] = bv_p2d_xr<1, 1>(c2, w3.x); y[10] = bv_p2d_xr<2, 0>(c2, w3.y); y[11] = bv_p2d_xr<3, 1>(c2, w3.y);
                y[12] = bv_p2d_xr<0, 0>(c3, w3.z); y[13] = bv_p2d_xr<1, 1>(c3, w3.z); y[14] = bv_p2d_xr<2, 0>(c3, w3.w); y[15] = bv_p2d_xr<3, 1>(c3, w3.w);
                // (a deep row -- an eighth of its cells are REF / ALT reads --: the dominant mapq's lanes are counted, not added one by
                // one; every other row: both histograms under ONE predicate -- the class byte is the same in x and y)
                if (((c_y & 0xFFFFu) + (c_y >> 16)) * 8u >= a.n_samples && !(a.flags & BV_FLAG_NO_DOM)) {
                    bv_lds_add16_dom<2>(x, hist, one, 0x200u, dom);
                    bv_lds_add16<2>(y, hist + 512, one, 0x200u);
                } else bv_lds_add16x2<2>(x, y, hist, hist + 512, one, 0x200u);
                }
                BV_PH(4);
            }
        }
        bv_lrt_sync<0>();

        // ---- the previous pass-1 row's stores are older than the (at most) K slots in flight: wait for exactly them, publish it
        if (prev_kind != 0u) {
            if (inflight == (uint32_t)BV_F_K) asm volatile(BV_F_W_ALL ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bv_f_publish(a, sh, B0, prev_site, prev_kind, lane);
            prev_kind = 0;
            // every pass-1 row of this wave below its next one is out: the row of this epilogue, or the one already drawn, or
            // -- none drawn -- whatever the cursor hands out next (the end of the pass-1 rows sets the mark to "all")
            // (a cursor past the workgroup's own range: everything of that range this wave had is out)
            const uint32_t cur_now = bv_f_lds_read_u(&sh.ctl[BV_FC_CURSOR]);
            uint32_t mark = is_p1 ? site : (((st & N_HAVE) && n_kind == BV_FK_P1) ? n_site : (cur_now < B1 - B0 ? B0 + cur_now : 0xFFFFFFF0u));
            if (lane == 0 && !(st & P1_FIN)) sh.pub[wave] = mark;
        }

        if (is_p1) {
        // ---- the row's totals (LDS operations of one wave execute in order: the adds above are done)
        uint32_t c[4][2], facc[4], racc[4];
        bool bad = false;
        uint32_t q0_mask = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int qr = 0; qr < 2; ++qr) {
                const int q = (qr << 6) | lane;
                const uint32_t f = hist[(b << 7) | q], r = hist[((b | 4) << 7) | q];
                c[b][qr] = f + r;
                if (qr == 0) { facc[b] = f; racc[b] = r; } else { facc[b] += f; racc[b] += r; }
                if (qr == 1) bad |= (c[b][qr] != 0u) && (q >= BV_NQ_VALID);
            }
            if (__builtin_amdgcn_readfirstlane((int)c[b][0]) != 0) q0_mask |= 1u << b;
        }
        uint32_t fwd[4], rev[4];
        {
            const uint32_t vv[8] = {facc[0], facc[1], facc[2], facc[3], racc[0], racc[1], racc[2], racc[3]};
            uint32_t t[8];
            bv_wave_sum8_u32(vv, t, lane);
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

        // ---- candidate or not (as bv_p1s_stream_kernel): not a candidate = nothing covered, or exactly one active base
        // (basetype.cpp:135-139), it is the reference base, none of its calls has phred 0, and the all-sites strand table is shallow
        bool is_cand = false;
        uint32_t n_active = 0;
        if (total != 0u && !(a.flags & BV_FLAG_TALLY_ONLY)) {
            const int bsel = lane & 3;
            const bool act = (double)bv_sel4u(depth, bsel) / (int)total >= a.min_af;  // basetype.cpp:137, one base per lane
            const uint32_t act_mask = (uint32_t)(__ballot(act) & 0xFull);
            n_active = (uint32_t)__popc(act_mask);
            int ref = (int)c_x;
            if (ref > 4) ref = 4;
            const bool one_ref = act_mask != 0u && (act_mask & (act_mask - 1u)) == 0u && ref < 4 && act_mask == (1u << ref);
            is_cand = !one_ref || (q0_mask & act_mask) != 0u;
            if (!is_cand) {
                // Fisher tables of (ref_fwd, ref_rev, alt_fwd, alt_rev): imax - imin + 1 (kfunc.c:253-257)
                const uint32_t rf = bv_sel4u(fwd, ref), rr = bv_sel4u(rev, ref);
                const uint32_t af = fwd[0] + fwd[1] + fwd[2] + fwd[3] - rf;
                const int n1_ = (int)(rf + rr), n_1 = (int)(rf + af), n = (int)total;
                const int imax = n_1 < n1_ ? n_1 : n1_;
                int imin = n1_ + n_1 - n;
                if (imin < 0) imin = 0;
                is_cand = (imax - imin + 1) > BV_S_SIMPLE_MAX_TABLES;
            }
        }
        uint32_t nb = 0, kind = 1u, n_stores = 1u;
        if (is_cand) {
            // every non-empty (base, phred < 128) bin, in (base, phred) order -- the order of bv_prologue_wave
            unsigned long long m[8];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
#pragma unroll
                for (int qr = 0; qr < 2; ++qr) {
                    m[b * 2 + qr] = __ballot(c[b][qr] != 0u);
                    nb += (uint32_t)__popcll(m[b * 2 + qr]);
                }
            }
            // four per wave (bv_solver16.h) unless the site needs what only the wave solver has: the ordered replay of a
            // shallow site, the literal 0/0 arithmetic of phred-0 calls or of min_af <= 0, or more than 128 bins
            const bool is_easy = q0_mask == 0u && total > (uint32_t)BV_ORD_MAX && nb <= (uint32_t)BV_G16_MAX_BINS && a.min_af > 0.0 &&
                                 !(a.flags & BV_FLAG_WAVE_SOLVER);
            kind = is_easy ? (n_active <= 2u ? 2u : 3u) : 4u;
            uint32_t *dst = a.bins + (size_t)site * BV_S_BIN_STRIDE;
            uint32_t pos0 = 0;
            // (opaque to the optimiser: it would otherwise keep the eight bin codes of a lane as loop invariants -- in scratch
            // memory, with a scratch load and an s_waitcnt vmcnt(0), a drained ring, per candidate row)
            uint32_t lane16 = (uint32_t)lane << 16;
            asm volatile("" : "+v"(lane16));
            if (is_easy) {
                // through the LDS stage: two 64-lane stores whatever the number of bins (words past it are never read)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int qr = 0; qr < 2; ++qr) {
                        const unsigned long long mm = m[b * 2 + qr];
                        const uint32_t pos = pos0 + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                        if (c[b][qr] != 0u) stage[pos] = ((((uint32_t)b << 7) | ((uint32_t)qr << 6)) << 16) | lane16 | c[b][qr];
                        pos0 += (uint32_t)__popcll(mm);
                    }
                }
                bv_lrt_sync<0>();
                const uint32_t w0 = stage[lane], w1 = stage[64 + lane];
                dst[lane] = w0;
                dst[64 + lane] = w1;
                n_stores = 3u;
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
#pragma unroll
                    for (int qr = 0; qr < 2; ++qr) {
                        const unsigned long long mm = m[b * 2 + qr];
                        const uint32_t pos = pos0 + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                        if (c[b][qr] != 0u) dst[pos] = ((((uint32_t)b << 7) | ((uint32_t)qr << 6)) << 16) | lane16 | c[b][qr];
                        pos0 += (uint32_t)__popcll(mm);
                    }
                }
                n_stores = 0u;  // not counted: the waits that follow are the conservative ones
            }
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
        prev_site = site;
        prev_kind = (uint32_t)__builtin_amdgcn_readfirstlane((int)kind);
        wsel = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_stores);
        } else {
        // ---- a variant site's two rank sums (ref_vs_alt_ranksumtest on mapq and read-position rank, caller.cpp:1151-1154) from
        // the [class][256] histograms, as exact integers; their phred values are formed for up to 64 rows at a time
        wsel = 0u;
        const unsigned long long n12 = (unsigned long long)(c_y & 0xFFFFu) + (unsigned long long)(c_y >> 16);
        if (__ballot(hi_acc != 0u) != 0ull) {
            // a rank >= 256 somewhere in the row (long reads): re-done at once by the exact window sweeps (the ring drains -- rare)
            const uint8_t *r_bs = a.bs, *r_mq = a.mapq;
            const uint16_t *r_rp = a.rpr;
            if (a.ch != nullptr) {
                const BvChainC ch = bv_chain_const(a.ch);
                const uint32_t sg = bv_chain_seg(ch, site);
                r_bs = ch->bs[sg]; r_mq = ch->mapq[sg]; r_rp = ch->rpr[sg];
            }
            bv_f_p2_redo(r_bs, r_mq, r_rp, a.out, a.pitch, a.n_samples, site, c_z | (tag ? 0x80000000u : 0u), c_y, (uint32_t)(uintptr_t)(bv_lds_u32 *)hist);
        } else {
            uint32_t *hm = hist, *hr = hist + 512;
            unsigned long long below = 0, twoR = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) twoR += bv_ranksum_window(hm[w * 64 + lane], hm[256 + w * 64 + lane], n12, below, lane);
            stash.tw_m = ((uint32_t)lane == stash.n) ? twoR : stash.tw_m;
            below = 0; twoR = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) twoR += bv_ranksum_window(hr[w * 64 + lane], hr[256 + w * 64 + lane], n12, below, lane);
            stash.tw_r = ((uint32_t)lane == stash.n) ? twoR : stash.tw_r;
            stash.site = ((uint32_t)lane == stash.n) ? site : stash.site;
            stash.n12 = ((uint32_t)lane == stash.n) ? c_y : stash.n12;
            if (++stash.n == 64u) bv_f_stash_flush(a, stash, lane);
        }
        }
        // ---- hand the histogram back, zeroed
        {
            uint4 *h4 = reinterpret_cast<uint4 *>(hist);
#pragma unroll
            for (int i = 0; i < BV_S_HWORDS / 4 / BV_WAVE; ++i) h4[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
            if (lane < 2) h4[BV_S_HWORDS / 4 + lane] = make_uint4(0, 0, 0, 0);
        }
        bv_lrt_sync<0>();
        if (is_p1) BV_PH(5); else BV_PH(6);
        if (st & N_HAVE) { c_site = n_site; c_kind = n_kind; c_x = n_x; c_y = n_y; c_z = n_z; st &= ~N_HAVE; }
        else st &= ~C_HAVE;
        // the last pass-1 row of this wave is behind it: publish it, count the wave
        if ((st & CUR_DONE) && !(st & P1_FIN) && !((st & C_HAVE) && c_kind == BV_FK_P1) && !((st & N_HAVE) && n_kind == BV_FK_P1)) finish_p1();
    }
    // ---- no row in flight: the ring is idle
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (prev_kind != 0u) {  // (a pass-1 row whose successor was not drawn yet: the variant queue had priority and was emptied by another wave)
        bv_f_publish(a, sh, B0, prev_site, prev_kind, lane);
        if (prev_kind == 4u) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        prev_kind = 0;
        if (lane == 0 && !(st & (P1_FIN | CUR_DONE))) {
            const uint32_t cur_now = bv_f_lds_read_u(&sh.ctl[BV_FC_CURSOR]);
            sh.pub[wave] = cur_now < B1 - B0 ? B0 + cur_now : 0xFFFFFFF0u;
        }
    }
    if ((st & CUR_DONE) && !(st & P1_FIN)) finish_p1();
    if (P2RING && stash.n != 0u) bv_f_stash_flush(a, stash, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BV_PH(8);
    BV_PH_FLUSH(a.counters, lane);
    (void)c_z;
    return st & (P_DONE | CUR_DONE | P1_FIN);
}

// ------------------------------------------------------------------------------ pass-2 rows, tallied from REGISTERS
// Behind the last pass-1 row the launch is ~140 us of the variant sites' rank-sum rows and of the solver's last jobs, and that phase
// is issue-bound: two streaming waves per SIMD tallying (~170 instructions per 1,024 cells and wave) beside a solver wave that runs
// out of jobs 70-100 us in.  Here those rows come by plain non-temporal loads, two blocks of 1,024 cells in flight
// per wave -- no ring, no LDS-DMA: all a wave needs is a 4 KiB histogram -- so the SOLVER waves, out of jobs, tally them too (their
// group scratch is the histogram): half as many waves again on every SIMD for the launch's last stretch.  Same tally, same epilogue,
// same records as the ring's pass-2 slots (bv_f_stream_until_idle).  For the tagged rank layout (BV_SLAB_RPR_TAGGED: mapq + ranks,
// the class of a cell from its rank word); rows of the plain layout, and every row with BV_FLAG_P2_TAIL_DMA (A/B, tests), keep the ring.
// Returns when no variant row can be had right now, or -- candidates first -- when a candidate waits for a solver.
struct BvP2Blk {
    bv_u32x4 w1, w2, w3;  // 16 cells per lane: mapq, ranks 0-7, ranks 8-15 (tagged: no call plane)
};
// 16 bytes of a plane, non-temporal, as a GLOBAL load (a generic pointer makes a flat one, which the compiler can only wait for with
// vmcnt(0) and lgkmcnt(0): no second block in flight under the first one's tally)
__device__ __forceinline__ bv_u32x4 bv_f_ntload16(const uint8_t *base, uint32_t off) {
    typedef const __attribute__((address_space(1))) bv_u32x4 *gp;
    return __builtin_nontemporal_load((gp)(uintptr_t)(base + off));
}
__device__ __attribute__((noinline, not_tail_called)) void bv_f_p2_rows(uint32_t ka_lo_, uint32_t ka_hi_, uint32_t sh_lds_, uint32_t hist_lds_, uint32_t B0_) {
    const uint32_t sh_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_lds_);
    const uint32_t hist_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)hist_lds_);
    const uint32_t B0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)B0_);
    BvFusedShared &sh = *(BvFusedShared *)(__attribute__((address_space(3))) BvFusedShared *)(uintptr_t)sh_lds;
    uint32_t *hist = (uint32_t *)(__attribute__((address_space(3))) uint32_t *)(uintptr_t)hist_lds;  // [2][256] mapq, [2][256] ranks: zero on entry, zero on return
    BvP1ShortArgs a;
    {
        const uint64_t kp = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)ka_hi_) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)ka_lo_);
        const __attribute__((address_space(4))) BvP1ShortArgs *ka = (const __attribute__((address_space(4))) BvP1ShortArgs *)(uintptr_t)kp;
        a.bs = BV_F_GLOBAL(const uint8_t, ka->bs); a.pitch = ka->pitch; a.n_samples = ka->n_samples; a.flags = ka->flags;
        a.out = BV_F_GLOBAL(bv_site_result, ka->out); a.counters = BV_F_GLOBAL(uint32_t, ka->counters); a.ch = BV_F_GLOBAL(const BvChain, ka->ch);
        a.mapq = BV_F_GLOBAL(const uint8_t, ka->mapq); a.rpr = BV_F_GLOBAL(const uint16_t, ka->rpr); a.ovf = BV_F_GLOBAL(uint32_t, ka->ovf);
        a.rpr_tag = ka->rpr_tag;
    }
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t qvhead_lds = (uint32_t)(uintptr_t)(bv_lds_u32 *)&sh.ctl[BV_FC_QV_HEAD];
    const uint32_t ovhead_lds = (uint32_t)(uintptr_t)(bv_lds_u32 *)&sh.ctl[BV_FC_OV_HEAD];
    const uint32_t n_chunks = (a.n_samples + 15u) >> 4;
    const int tail = (int)(a.n_samples & 15u);
    const uint32_t n_blk = (n_chunks + 63u) >> 6;
    const uint32_t last2 = n_chunks - (n_blk - 1u) * 64u;  // chunks of a row's last block: 1 .. 64
    const uint32_t hi_mask = bv_rpr_hi_mask(1u);
    uint32_t one;
    asm volatile("v_mov_b32 %0, 1" : "=v"(one));
    BvFusedStash stash;
    stash.site = 0; stash.n12 = 0; stash.tw_m = 0; stash.tw_r = 0; stash.n = 0;
    // a row: the variant site's facts (class table, REF / ALT depths, the window sweeps' 2-bit table) and its planes
    struct Row {
        uint32_t site, L, n12w, lut;
        const uint8_t *pm, *pr, *r_bs, *r_mq;
        const uint16_t *r_rp;
    };
    // a variant site from the queue in LDS, else from the overflow list (as bv_f_stream_until_idle draws them); false: none right now
    auto pop = [&](Row &r) __attribute__((always_inline)) -> bool {
#pragma unroll 1
        for (;;) {
            const uint32_t h = bv_f_lds_read_u(&sh.ctl[BV_FC_QV_HEAD]);
            if (h != bv_f_lds_read_u(&sh.ctl[BV_FC_QV_TAIL])) {
                if (bv_f_lds_cas_wave(qvhead_lds, h, h + 1u) != h) continue;  // (another wave took it: look again)
                const uint32_t *e = sh.qv[h & (BV_F_QVCAP - 1u)];
                uint32_t site = BV_F_EMPTY;
                for (uint32_t spins = 0; (site = bv_f_lds_read_u(&e[0])) == BV_F_EMPTY && spins < BV_F_SPIN_MAX; ++spins) __builtin_amdgcn_s_sleep(4);
                r.L = bv_f_lds_read_u(&e[1]); r.n12w = bv_f_lds_read_u(&e[2]); r.lut = bv_f_lds_read_u(&e[3]);
                if (lane == 0) *(bv_lds_vu32 *)&e[0] = BV_F_EMPTY;
                if (site == BV_F_EMPTY) { if (lane == 0) atomicOr(&a.counters[BV_CTR_TIMEOUT], BV_TMO_TAKE_VARIANT); continue; }  // (timed out: flagged)
                r.site = site;
            } else {
                const uint32_t oh = bv_f_lds_read_u(&sh.ctl[BV_FC_OV_HEAD]);
                if (oh == bv_f_lds_read_u(&sh.ctl[BV_FC_OV_TAIL])) return false;
                if (bv_f_lds_cas_wave(ovhead_lds, oh, oh + 1u) != oh) continue;
                bv_u32x4 e;  // (through the L2: the entry was written by a solver wave of this workgroup)
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(e) : "v"(a.ovf + 4u * (size_t)(B0 + oh)) : "memory");
                r.site = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.x); r.L = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.y);
                r.n12w = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.z); r.lut = (uint32_t)__builtin_amdgcn_readfirstlane((int)e.w);
            }
            r.r_bs = a.bs; r.r_mq = a.mapq; r.r_rp = a.rpr;
            if (a.ch != nullptr) {
                const BvChainC ch = bv_chain_const(a.ch);
                const uint32_t sg = bv_chain_seg(ch, r.site);
                r.r_bs = ch->bs[sg]; r.r_mq = ch->mapq[sg]; r.r_rp = ch->rpr[sg];
            }
            const uint64_t off = (uint64_t)r.site * a.pitch;
            r.pm = bv_uniform_ptr(r.r_mq + off);
            r.pr = bv_uniform_ptr(reinterpret_cast<const uint8_t *>(r.r_rp) + 2u * off);
            return true;
        }
    };
    // The rows as ONE stream of blocks, a row on an even number of positions (an odd row's last position is read again and not
    // tallied), even positions in A, odd ones in B: position g of the current row -- or, from g = n_even on, position g - n_even of the
    // NEXT row, so that a row's first two blocks are in flight under the epilogue of the row before it.  Every load UNCONDITIONAL (a
    // block past the last row's end is that row's last block again; a lane past the end of a last block reads the block's last
    // valid chunk -- in bounds, and replaced by cells that count nowhere before the tally looks at it): behind a branch the compiler
    // waits for everything in flight, in straight-line code it counts.
    const uint32_t n_even = (n_blk + 1u) & ~1u;
    Row cur, nxt;
    bool have_nxt = false;
    auto load_pos = [&](uint32_t g, BvP2Blk &W) __attribute__((always_inline)) {
        const bool in_next = g >= n_even && have_nxt;
        uint32_t b = in_next ? g - n_even : g;
        b = b < n_blk ? b : n_blk - 1u;
        const uint8_t *pm = in_next ? nxt.pm : cur.pm, *pr = in_next ? nxt.pr : cur.pr;
        const uint32_t ln = (b + 1u == n_blk && (uint32_t)lane >= last2) ? last2 - 1u : (uint32_t)lane;
        W.w1 = bv_f_ntload16(pm + (size_t)b * 1024u, ln * 16u);
        W.w2 = bv_f_ntload16(pr + (size_t)b * 2048u, ln * 32u);
        W.w3 = bv_f_ntload16(pr + (size_t)b * 2048u, ln * 32u + 16u);
    };
    BvP2Blk A, B;
    bool have = pop(cur);
    if (have) {
        load_pos(0u, A);
        load_pos(1u, B);
    }
#pragma unroll 1
    while (have) {
        const uint32_t site = cur.site, L = cur.L, n12w = cur.n12w;
        const bool deep = ((n12w & 0xFFFFu) + (n12w >> 16)) * 8u >= a.n_samples && !(a.flags & BV_FLAG_NO_DOM);
        uint32_t hi_acc = 0, dom = BV_DOM_NONE;
        // the tally of a block, as the ring's pass-2 slots are tallied (bv_f_stream_until_idle)
        auto tally_blk = [&](uint32_t b, BvP2Blk &W) __attribute__((always_inline)) {
            const bool lastb = b + 1u == n_blk;
            if (lastb && (uint32_t)lane >= last2) {  // (not the lane's own cells: see load_pos)
                W.w1 = bv_u32x4{0u, 0u, 0u, 0u};
                W.w2 = bv_u32x4{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
                W.w3 = W.w2;
            }
            if (lastb && tail && (uint32_t)lane == last2 - 1u) {
                W.w2.x = bv_p2t_mask_tail(W.w2.x, tail); W.w2.y = bv_p2t_mask_tail(W.w2.y, tail - 2);
                W.w2.z = bv_p2t_mask_tail(W.w2.z, tail - 4); W.w2.w = bv_p2t_mask_tail(W.w2.w, tail - 6);
                W.w3.x = bv_p2t_mask_tail(W.w3.x, tail - 8); W.w3.y = bv_p2t_mask_tail(W.w3.y, tail - 10);
                W.w3.z = bv_p2t_mask_tail(W.w3.z, tail - 12); W.w3.w = bv_p2t_mask_tail(W.w3.w, tail - 14);
            }
            const uint32_t c0 = bv_p2t_class4(L, W.w2.x, W.w2.y), c1 = bv_p2t_class4(L, W.w2.z, W.w2.w);
            const uint32_t c2 = bv_p2t_class4(L, W.w3.x, W.w3.y), c3 = bv_p2t_class4(L, W.w3.z, W.w3.w);
            // ranks that do not fit the 256-rank window: remembered, the row is then re-done by the window sweeps
            hi_acc |= (W.w2.x | W.w2.y | W.w2.z | W.w2.w | W.w3.x | W.w3.y | W.w3.z | W.w3.w) & hi_mask;
            uint32_t x[16], y[16];
            x[0] = bv_p2d_xm<0>(c0, W.w1.x); x[1] = bv_p2d_xm<1>(c0, W.w1.x); x[2] = bv_p2d_xm<2>(c0, W.w1.x); x[3] = bv_p2d_xm<3>(c0, W.w1.x);
            x[4] = bv_p2d_xm<0>(c1, W.w1.y); x[5] = bv_p2d_xm<1>(c1, W.w1.y); x[6] = bv_p2d_xm<2>(c1, W.w1.y); x[7] = bv_p2d_xm<3>(c1, W.w1.y);
            x[8] = bv_p2d_xm<0>(c2, W.w1.z); x[9] = bv_p2d_xm<1>(c2, W.w1.z); x[10] = bv_p2d_xm<2>(c2, W.w1.z); x[11] = bv_p2d_xm<3>(c2, W.w1.z);
            x[12] = bv_p2d_xm<0>(c3, W.w1.w); x[13] = bv_p2d_xm<1>(c3, W.w1.w); x[14] = bv_p2d_xm<2>(c3, W.w1.w); x[15] = bv_p2d_xm<3>(c3, W.w1.w);
            y[0] = bv_p2d_xr<0, 0>(c0, W.w2.x); y[1] = bv_p2d_xr<1, 1>(c0, W.w2.x); y[2] = bv_p2d_xr<2, 0>(c0, W.w2.y); y[3] = bv_p2d_xr<3, 1>(c0, W.w2.y);
            y[4] = bv_p2d_xr<0, 0>(c1, W.w2.z); y[5] = bv_p2d_xr<1, 1>(c1, W.w2.z); y[6] = bv_p2d_xr<2, 0>(c1, W.w2.w); y[7] = bv_p2d_xr<3, 1>(c1, W.w2.w);
            y[8] = bv_p2d_xr<0, 0>(c2, W.w3.x); y[9] = bv_p2d_xr<1, 1>(c2, W.w3.x); y[10] = bv_p2d_xr<2, 0>(c2, W.w3.y); y[11] = bv_p2d_xr<3, 1>(c2, W.w3.y);
            y[12] = bv_p2d_xr<0, 0>(c3, W.w3.z); y[13] = bv_p2d_xr<1, 1>(c3, W.w3.z); y[14] = bv_p2d_xr<2, 0>(c3, W.w3.w); y[15] = bv_p2d_xr<3, 1>(c3, W.w3.w);
            if (deep) {
                bv_lds_add16_dom<2>(x, hist, one, 0x200u, dom);
                bv_lds_add16<2>(y, hist + 512, one, 0x200u);
            } else bv_lds_add16x2<2>(x, y, hist, hist + 512, one, 0x200u);
        };
#pragma unroll 1
        for (uint32_t b = 0; b < n_even; b += 2u) {
            tally_blk(b, A);
            if (b + 2u == n_even) {
                // the row's last pair: the next row is drawn here, its first two blocks are requested below -- unless a candidate waits:
                // the solver's jobs come first (see the kernel's loop), and the wave leaves behind this row's epilogue
                const bool cand = bv_f_lds_read_u(&sh.ctl[BV_FC_Q3_TAIL]) != bv_f_lds_read_u(&sh.ctl[BV_FC_Q3_HEAD]) ||
                                  bv_f_lds_read_u(&sh.ctl[BV_FC_Q2_TAIL]) != bv_f_lds_read_u(&sh.ctl[BV_FC_Q2_HEAD]);
                have_nxt = !cand && pop(nxt);
            }
            load_pos(b + 2u, A);
            if (b + 1u < n_blk) tally_blk(b + 1u, B);
            load_pos(b + 3u, B);
        }
        bv_lrt_sync<0>();
        // ---- the site's two rank sums (as in bv_f_stream_until_idle)
        const unsigned long long n12 = (unsigned long long)(n12w & 0xFFFFu) + (unsigned long long)(n12w >> 16);
        if (__ballot(hi_acc != 0u) != 0ull) {
            bv_f_p2_redo(cur.r_bs, cur.r_mq, cur.r_rp, a.out, a.pitch, a.n_samples, site, cur.lut | 0x80000000u, n12w, hist_lds);
        } else {
            uint32_t *hm = hist, *hr = hist + 512;
            unsigned long long below = 0, twoR = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) twoR += bv_ranksum_window(hm[w * 64 + lane], hm[256 + w * 64 + lane], n12, below, lane);
            stash.tw_m = ((uint32_t)lane == stash.n) ? twoR : stash.tw_m;
            below = 0; twoR = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) twoR += bv_ranksum_window(hr[w * 64 + lane], hr[256 + w * 64 + lane], n12, below, lane);
            stash.tw_r = ((uint32_t)lane == stash.n) ? twoR : stash.tw_r;
            stash.site = ((uint32_t)lane == stash.n) ? site : stash.site;
            stash.n12 = ((uint32_t)lane == stash.n) ? n12w : stash.n12;
            if (++stash.n == 64u) bv_f_stash_flush(a, stash, lane);
        }
        {
            uint4 *h4 = reinterpret_cast<uint4 *>(hist);
#pragma unroll
            for (int i = 0; i < BV_S_HWORDS / 4 / BV_WAVE; ++i) h4[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
        }
        bv_lrt_sync<0>();
        have = have_nxt;
        cur = nxt;
        have_nxt = false;
    }
    if (stash.n != 0u) bv_f_stash_flush(a, stash, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <bool FUSE2>
__global__ __launch_bounds__(BV_WAVE *BV_F_NW) void bv_p1s_fused_kernel(BvP1ShortArgs a) {
    __shared__ BvFusedShared sh;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // every workgroup owns a contiguous range of sites
    const uint32_t B0 = (uint32_t)((uint64_t)a.n_sites * blockIdx.x / gridDim.x), B1 = (uint32_t)((uint64_t)a.n_sites * (blockIdx.x + 1) / gridDim.x);
    // ---- set-up: histograms zeroed, queues empty, tables in LDS; nothing is in flight yet, so a plain barrier is fine
    if (wave < BV_F_NS) {
        uint4 *h4 = reinterpret_cast<uint4 *>(sh.hist[wave]);
#pragma unroll
        for (int i = 0; i < (BV_S_HWORDS + BV_S_OVF + 8) / 4 / BV_WAVE + 1; ++i)
            if (i * BV_WAVE + lane < (BV_S_HWORDS + BV_S_OVF + 8) / 4) h4[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
    }
    for (int i = tid; i < BV_F_QCAP; i += BV_WAVE * BV_F_NW) { sh.q3[i] = BV_F_EMPTY; sh.q2[i] = BV_F_EMPTY; }
    for (int i = tid; i < BV_F_QVCAP; i += BV_WAVE * BV_F_NW) sh.qv[i][0] = BV_F_EMPTY;
    for (int i = tid; i < BV_QBINS; i += BV_WAVE * BV_F_NW) {
        sh.tab_hit[i] = a.tables->hit[i];
        sh.tab_miss[i] = a.tables->miss[i];
        sh.tab_loghit[i] = a.tables->loghit[i];
        sh.tab_logmiss[i] = a.tables->logmiss[i];
    }
    if (tid < 16) sh.ctl[tid] = 0u;
    if (tid < BV_F_NS) sh.pub[tid] = B0;
#ifdef BV_TEAM_DEBUG
    if (tid == 0) {
        uint32_t *dbg_ = a.counters + BV_CTR_WORDS + (blockIdx.x < 512u ? blockIdx.x : 511u) * 8u;
        dbg_[0] = (uint32_t)__builtin_amdgcn_s_memrealtime(); dbg_[3] = 0u; dbg_[5] = 0u;
        dbg_[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
        if (blockIdx.x == 0) a.counters[BV_CTR_WORDS + 5150] = 4u;  // whose stamps these are
#ifdef BV_PHASE_DEBUG
        if (blockIdx.x == 0) for (int i = 0; i < 60; ++i) a.counters[BV_CTR_WORDS + 4200 + i] = 0u;
#endif
    }
#endif
    __syncthreads();

    const bool is_stream = wave < BV_F_NS;
    bool streaming = is_stream;
    uint32_t sst = 0;
    BvFusedSolver v;
    v.sa.ref_base = a.ref_base; v.sa.out = a.out; v.sa.var_list = a.var_list; v.sa.counters = a.counters;
    v.sa.min_af = a.min_af; v.sa.flags = a.flags;
    {
        // (wave-uniform values read through vector loads -- the tables are not provably unwritten -- go to scalar registers:
        // as vector registers they are live across the whole loop and end up in scratch memory)
        const uint64_t lf = (uint64_t)(uintptr_t)a.tables->lnfact;
        v.sa.lnfact.t = (decltype(v.sa.lnfact.t))(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(lf >> 32)) << 32) |
                                                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)lf));
        v.sa.lnfact.n = __builtin_amdgcn_readfirstlane((int)a.tables->lnfact_n);
    }
    v.sa.loghit = a.tables->loghit; v.sa.logmiss = a.tables->logmiss;
    v.sa.bs = a.bs; v.sa.q = a.q; v.sa.pitch = a.pitch; v.sa.n_samples = a.n_samples;
    v.n_vl = 0;
    v.fuse2 = FUSE2;
    // this wave's solver scratch: a streaming wave's ring (while no row is in flight), a solver wave's own
    if (is_stream) {
        v.grp = &sh.ring[wave].solve.ws.grp[0][0]; v.vl = sh.ring[wave].solve.vl; v.big = &sh.ring[wave].solve.ws;
    } else {
        v.grp = &sh.grp[wave - BV_F_NS][0][0]; v.vl = sh.vl[wave - BV_F_NS]; v.big = nullptr;
    }
    const uint32_t sh_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) BvFusedShared *)&sh;
    const uint64_t ka_ptr = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();  // the argument block `a`, as the streaming function reads it
    // A streaming wave streams while it has rows; whenever none is in flight (waiting for a variant row, or finished) it does
    // one unit of solver work with its ring as scratch.  ONE call site of the solver step for both kinds of waves (the solver
    // is most of the kernel's code).
#pragma unroll 1
    for (;;) {
        // A streaming wave past its pass-1 rows calls the streaming function only when a variant row waits (two LDS words
        // tell): until round 5 the function's prologue saved 60 callee-saved VGPRs to scratch memory per call, and waves that
        // polled through it wrote 316 MB per launch of 8,192 sites.  The saves are gone (see the function), the test stays: a
        // call still reads the argument block and sets the ring up.
        // (the tagged rank layout only: with the call plane to read as well the plain loads lose to the ring -- 0.533 against 0.519 ms)
        const bool p2_regs = FUSE2 && a.rpr_tag != 0u && !(a.flags & BV_FLAG_P2_TAIL_DMA);
        bool go = streaming;
        if (streaming && (sst & BV_FS_CUR_DONE) && (sst & BV_FS_P1_FIN)) {
            if (!FUSE2) { go = false; streaming = false; }
            else if (bv_f_lds_read_u(&sh.ctl[BV_FC_QV_TAIL]) == bv_f_lds_read_u(&sh.ctl[BV_FC_QV_HEAD]) &&
                     bv_f_lds_read_u(&sh.ctl[BV_FC_OV_TAIL]) == bv_f_lds_read_u(&sh.ctl[BV_FC_OV_HEAD])) {
                go = false;
                if (bv_f_no_row_ever(sh.ctl)) streaming = false;
            }
        }
        // (the lane number as a value the compiler cannot see through: what the solver derives from it -- lane & 15, masks,
        // scratch offsets, some forty values -- was hoisted out of this loop and then lived in scratch memory, a memory trip per use)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if (go) {
            // (the wave's variant sites since its last flush sit in its ring's LDS: out before rows stream through it again)
            if (v.n_vl) bv_f_flush_vl(a, v, ln);
            // past its pass-1 rows a wave tallies the variant sites' rows from registers (bv_f_p2_rows); until then -- and with
            // BV_FLAG_P2_TAIL_DMA always -- rows go through the ring
            if (p2_regs && (sst & BV_FS_CUR_DONE) && (sst & BV_FS_P1_FIN)) {
                // Candidates first.  Behind the last pass-1 row the launch is as long as the solver's last jobs (62-74 us each) plus the rows
                // of the variant sites they find; a candidate that waits for a solver wave to finish its job first adds up to a whole
                // job to that -- and keeps the solver waves from joining the tally of the rows.
                const uint32_t least = bv_f_lds_read_u(&sh.ctl[BV_FC_NDONE]) == (uint32_t)BV_F_NS ? 1u : BV_F_MIN_JOB;
                const bool cand = bv_f_lds_read_u(&sh.ctl[BV_FC_Q3_TAIL]) - bv_f_lds_read_u(&sh.ctl[BV_FC_Q3_HEAD]) >= least ||
                                  bv_f_lds_read_u(&sh.ctl[BV_FC_Q2_TAIL]) - bv_f_lds_read_u(&sh.ctl[BV_FC_Q2_HEAD]) >= least;
                if (!cand) bv_f_p2_rows((uint32_t)ka_ptr, (uint32_t)(ka_ptr >> 32), sh_lds, (uint32_t)(uintptr_t)(bv_lds_u32 *)sh.hist[wave], B0);
            } else if (FUSE2 && !((sst & BV_FS_CUR_DONE) && (sst & BV_FS_P1_FIN)))
                // (until a wave is past its pass-1 rows its ring carries nothing else, whatever streams the variant rows later: the
                // instance of the function without the pass-2 tally -- it returns when the cursor is exhausted and its rows are out)
                sst = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv_f_stream_until_idle<FUSE2, false>((uint32_t)ka_ptr, (uint32_t)(ka_ptr >> 32), sh_lds, (uint32_t)wave, B0, B1, sst));
            else
                sst = (uint32_t)__builtin_amdgcn_readfirstlane((int)bv_f_stream_until_idle<FUSE2, FUSE2>((uint32_t)ka_ptr, (uint32_t)(ka_ptr >> 32), sh_lds, (uint32_t)wave, B0, B1, sst));
            if (sst & BV_FS_P_DONE) streaming = false;
        }
        const int r = bv_f_solver_step(a, sh, v, B0, B1, ln);
        if (streaming) {
            if (r != 1) __builtin_amdgcn_s_sleep(8);
            continue;
        }
        if (p2_regs && !is_stream && r != 1) {
            // A solver wave with nothing to solve: once every streaming wave is past its pass-1 rows it tallies variant rows too -- from
            // registers, with its group scratch as the histogram (zeroed here, and left zero) -- and stays until no row can come any more.
            if (bv_f_lds_read_u(&sh.ctl[BV_FC_NDONE]) == (uint32_t)BV_F_NS) {
                if (bv_f_lds_read_u(&sh.ctl[BV_FC_QV_TAIL]) != bv_f_lds_read_u(&sh.ctl[BV_FC_QV_HEAD]) ||
                    bv_f_lds_read_u(&sh.ctl[BV_FC_OV_TAIL]) != bv_f_lds_read_u(&sh.ctl[BV_FC_OV_HEAD])) {
                    if (v.n_vl) bv_f_flush_vl(a, v, ln);
                    uint4 *h4 = reinterpret_cast<uint4 *>(v.grp);
#pragma unroll
                    for (int i = 0; i < BV_S_HWORDS / 4 / BV_WAVE; ++i) h4[i * BV_WAVE + ln] = make_uint4(0, 0, 0, 0);
                    bv_lrt_sync<0>();
                    bv_f_p2_rows((uint32_t)ka_ptr, (uint32_t)(ka_ptr >> 32), sh_lds, (uint32_t)(uintptr_t)(bv_lds_u32 *)v.grp, B0);
                    continue;
                }
                if (r == 2 && bv_f_no_row_ever(sh.ctl)) break;
            }
            __builtin_amdgcn_s_sleep(8);
            continue;
        }
        if (r == 2) break;
        if (r == 0) __builtin_amdgcn_s_sleep(8);
    }
    // (the solver step that reports "nothing left, ever" has flushed the wave's variant list)
#ifdef BV_TEAM_DEBUG
    if (lane == 0) atomicMax(&a.counters[BV_CTR_WORDS + (blockIdx.x < 512u ? blockIdx.x : 511u) * 8u + 3u], (uint32_t)__builtin_amdgcn_s_memrealtime());
#endif
}

// ------------------------------------------------------------------------------ launcher
bool bv_p1s_fused_takes(const BvP1ShortArgs &a) {
    // rows of at least BV_F_K slots (the prefetch then never runs more than one row ahead); chained launches too (the planes of
    // a row are looked up per segment where its address is formed; ref_base / out are the engine's contiguous copies)
    const uint32_t n_chunks = (a.n_samples + 15u) >> 4, n_slots = (n_chunks + 127u) >> 7;
    return n_slots >= (uint32_t)BV_F_K && a.n_samples <= BV_SHORT_ROW_MAX;
}
void bv_launch_p1s_fused(const BvP1ShortArgs &a, hipStream_t stream) {
    const uint32_t cu = a.n_cu ? a.n_cu : 256u;
    uint32_t grid = cu;
    const uint32_t need = (a.n_sites + BV_F_NS - 1) / BV_F_NS;  // at least one site per streaming wave
    if (grid > need) grid = need > 0 ? need : 1;
    const uint32_t cap = (a.flags >> 16) & 0xFFu;  // BV_FLAG_GRID_LIMIT
    if (cap && grid > cap) grid = cap;
    // with the rank planes (a.mapq, a.rpr): the variant sites' pass-2 rows are streamed by the same waves
    if (a.mapq != nullptr && a.rpr != nullptr) hipLaunchKernelGGL(bv_p1s_fused_kernel<true>, dim3(grid), dim3(BV_WAVE * BV_F_NW), 0, stream, a);
    else hipLaunchKernelGGL(bv_p1s_fused_kernel<false>, dim3(grid), dim3(BV_WAVE * BV_F_NW), 0, stream, a);
}
