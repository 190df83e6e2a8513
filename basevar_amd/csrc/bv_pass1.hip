// bv_pass1.hip -- pass 1 of the per-site basetype path: tally + solve, every site.
//
// Wave-specialised, persistent workgroups.  A workgroup is NTALLY tally waves + NSOLVE solver
// waves:
//
//   tally waves  stream one site (one slab row) after another, the row's 4 KiB blocks dealt
//                round-robin to the NTALLY waves: 16-byte coalesced, non-temporal loads of the
//                two byte planes along the sample axis (2 B/cell, each byte read exactly once),
//                software-pipelined (two sets of 4 KiB per plane, 16 KiB in flight per wave), covered
//                cells tallied with LDS atomics into a (strand, base, phred) histogram -- 8 KiB, one
//                of a small ring.  Sets that lie wholly inside the row take an unguarded fast path;
//                each 16-byte chunk is hand-scheduled (bv_tally_chunk).
//   solver wave  takes a finished histogram and runs the whole reference solver on it
//                (cites below), writes the site's 208-byte record, re-zeroes the histogram
//                and hands it back.
//
// The roles meet only through LDS sequence flags (published[] / filled[] / drained[]), never
// through s_barrier, so the HBM stream of site k+1 runs under the FP64 dependency chains of site k.
// (A first version that tallied and then solved with the same waves ran both phases in
// lock-step across the workgroups of a CU and reached 39 % of HBM peak; its tally alone
// ran at 75 %.  See DESIGN.md.)
//
// Reference functions realised here, all on the histogram:
//   BaseType ctor counts        src/basetype.cpp:45-71      -> depth[], total_depth
//   strand_bias (CVG flavour)   src/basetype.cpp:244-295    via caller.cpp:1236-1245
//   lrt(): EM / LRT / AF / QUAL src/basetype.cpp:130-199, src/algorithm.h:148-255
//   strand_bias (VCF flavour)   caller.cpp:1164
//   base-quality rank sum       src/basetype.cpp:201-242 via caller.cpp:1157
//   QD, CM_CAF                  caller.cpp:1122, 1160-1161
// HBM-bound by design; no inter-site reuse, so nothing to gain from an XCD-aware block
// remap (no two workgroups share a byte).  No MFMA: categorical tallies + small FP64 tables.
#include "bv_kernels.h"

#include "bv_solver.h"
#include "bv_tally.h"

#define BV_RING_EXTRA 2
template <int NBUF, int NSOLVE>
struct __attribute__((aligned(16))) BvPass1Shared {
    uint32_t hist[NBUF][BV_H2_WORDS];  // ring of histograms [(rev<<2)|base][phred byte]
    uint32_t published[NBUF];          // times a site has been assigned to the slot (lead tally wave)
    uint32_t filled[NBUF];             // tally-wave arrivals on the slot (NTALLY per fill)
    uint32_t drained[NBUF];            // times the slot has been solved and re-zeroed
    uint32_t site_of[NBUF];            // site held by the slot (0xFFFFFFFF = no more work)
    uint32_t swz_of[NBUF];             // 1: the slot's row is tallied with the dense-row swizzle (bv_tally_chunk<.., SWZ>); set with site_of
    uint32_t dense_hint;               // the solver's: the row it solved last was dense (a quarter of its cells covered)
    double tab_hit[BV_QBINS], tab_miss[BV_QBINS];  // LDS copy of BvTables
    BvSolverShared sv[NSOLVE];
};

// One wave streams one row, software-pipelined: two register sets of U chunks per plane, the
// loads of the next set are in flight (16 KiB per wave) while the current set is tallied.
#define BV_TALLY_U 4
struct BvChunkSet {
    bv_u32x4 vb[BV_TALLY_U], vq[BV_TALLY_U];
};
// FULL: every chunk of the set lies wholly inside the row (wave-uniform fact) -- unguarded loads off a
// scalar base (row pointer + set offset in SGPRs, lane offset in one VGPR, the chunk's 1 KiB step in
// the instruction's immediate) and no tail masking.  !FULL: the row's last, partial set.
template <bool FULL>
__device__ __forceinline__ void bv_chunks_load(BvChunkSet &c, const uint8_t *bs_row, const uint8_t *q_row, uint32_t base,
                                               uint32_t n_chunks, int lane) {
    const uint8_t *pb = bs_row + (size_t)base * 16u, *pq = q_row + (size_t)base * 16u;  // uniform
    const uint32_t voff = (uint32_t)lane * 16u;
#pragma unroll
    for (int u = 0; u < BV_TALLY_U; ++u) {
        const uint32_t off = voff + (uint32_t)u * (BV_WAVE * 16u);
        if (FULL || base + u * BV_WAVE + lane < n_chunks) {
            c.vb[u] = __builtin_nontemporal_load(reinterpret_cast<const bv_u32x4 *>(pb + off));
            c.vq[u] = __builtin_nontemporal_load(reinterpret_cast<const bv_u32x4 *>(pq + off));
        } else {
            c.vb[u] = bv_u32x4{0x08080808u, 0x08080808u, 0x08080808u, 0x08080808u};
            c.vq[u] = bv_u32x4{0u, 0u, 0u, 0u};
        }
    }
}
template <bool FULL, bool SWZ>
__device__ __forceinline__ void bv_chunks_tally(BvChunkSet &c, uint32_t base, uint32_t n_chunks, int tail, int lane,
                                                uint32_t *hist, uint32_t one) {
#pragma unroll
    for (int u = 0; u < BV_TALLY_U; ++u) {
        if (!FULL) {
            uint32_t idx = base + u * BV_WAVE + lane;
            if (tail && idx == n_chunks - 1) {
                c.vb[u].x = bv_mask_tail_dword(c.vb[u].x, tail);
                c.vb[u].y = bv_mask_tail_dword(c.vb[u].y, tail - 4);
                c.vb[u].z = bv_mask_tail_dword(c.vb[u].z, tail - 8);
                c.vb[u].w = bv_mask_tail_dword(c.vb[u].w, tail - 12);
            }
        }
        bv_tally_chunk<2, SWZ>(c.vb[u], c.vq[u], hist, one);
    }
}
// set at `base`: nothing to do past the row's end; FULL form when the set ends inside the row
#define BV_SET_LOAD(C, BASE)                                                                  \
    do {                                                                                      \
        const uint32_t b_ = (BASE);                                                           \
        if (b_ + BLK <= n_full) bv_chunks_load<true>(C, bs_row, q_row, b_, n_chunks, lane);   \
        else if (b_ < n_chunks) bv_chunks_load<false>(C, bs_row, q_row, b_, n_chunks, lane);  \
    } while (0)
#define BV_SET_TALLY(C, BASE)                                                                 \
    do {                                                                                      \
        const uint32_t b_ = (BASE);                                                           \
        if (b_ + BLK <= n_full) bv_chunks_tally<true, SWZ>(C, b_, n_chunks, tail, lane, hist, one);\
        else if (b_ < n_chunks) bv_chunks_tally<false, SWZ>(C, b_, n_chunks, tail, lane, hist, one);\
    } while (0)
// PRE: the row's first set is already in (or on its way into) A -- see bv_row_preload.
template <int NTALLY, bool PRE = false, bool SWZ = false>
__device__ __forceinline__ void bv_tally_row_wave(const uint8_t *bs_row, const uint8_t *q_row, uint32_t n_samples,
                                                  uint32_t *hist, int t, int lane, BvChunkSet *pre = nullptr) {
    const uint32_t n_chunks = (n_samples + 15u) >> 4, n_full = n_samples >> 4;
    const int tail = (int)(n_samples & 15u);
    constexpr uint32_t BLK = BV_WAVE * BV_TALLY_U;  // chunks per wave-iteration (4 KiB of each plane)
    constexpr uint32_t STRIDE = BLK * NTALLY;       // the row's blocks go round-robin over the tally waves
    uint32_t one;
    asm volatile("v_mov_b32 %0, 1" : "=v"(one));  // opaque: not re-materialised per cell
    BvChunkSet A, B;
    uint32_t base = (uint32_t)t * BLK;
    if (PRE) A = *pre;
    else BV_SET_LOAD(A, base);
    for (; base < n_chunks; base += 2 * STRIDE) {
        BV_SET_LOAD(B, base + STRIDE);
        BV_SET_TALLY(A, base);
        BV_SET_LOAD(A, base + 2 * STRIDE);
        BV_SET_TALLY(B, base + STRIDE);
    }
}
// issue the loads of a row's first set (the caller keeps the registers alive across other work)
__device__ __forceinline__ void bv_row_preload(BvChunkSet &A, const uint8_t *bs_row, const uint8_t *q_row,
                                               uint32_t n_samples, int lane) {
    const uint32_t n_chunks = (n_samples + 15u) >> 4, n_full = n_samples >> 4;
    constexpr uint32_t BLK = BV_WAVE * BV_TALLY_U;
    BV_SET_LOAD(A, 0u);
}
#undef BV_SET_LOAD
#undef BV_SET_TALLY

// ------------------------------------------------------------------------------ kernel
// Every spin is bounded (~1 s): a protocol bug must end the kernel with counters[3] set
// (reported by bv_engine_wait) instead of hanging the GPU.
#define BV_SPIN_SLEEP 16 /* x64 cycles between polls of a hand-off flag (4, 16, 64 measured: 4.09, 4.05, 4.04 ms) */
__device__ __forceinline__ void bv_wait_flag(const uint32_t *flag, uint32_t want, uint32_t *err) {
    uint32_t spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != want) {
        __builtin_amdgcn_s_sleep(BV_SPIN_SLEEP);
        if (++spins > (1u << 22)) {  // ~2 s
            atomicOr(err, BV_TMO_RING);  // (or, not add: sixteen waves giving up must not wrap the word to 0)
            break;
        }
    }
}
__device__ __forceinline__ void bv_set_flag(uint32_t *flag, uint32_t v) {
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// one arrival per WAVE (lane 0 only; the wave's earlier LDS atomics are ahead of it in the
// in-order LDS queue)
__device__ __forceinline__ void bv_add_flag(uint32_t *flag, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ------------------------------------------------------------------------------ the tail of a launch: team solves
// When a workgroup has streamed its last row, its tally waves used to exit while the solver wave worked through what was
// left in the ring -- a whole solve (8 EM runs, two Fisher tests, QUAL, a rank sum: 40-70 us on one wave at 10^5
// samples) with the HBM pipes empty.  On a 131,072-site batch that is 1-2 % of the launch; on a single 8,192-site batch it
// was 23 % (0.363 ms against 0.280 ms for the tally alone).  In the TEAM form of the kernel the tally waves stay: waves
// 0 .. NTALLY-2 join the solver wave for the LRT (bv_lrt's team mode: the EM runs of a level are independent,
// basetype.cpp:151-169), the last tally wave runs the site's Fisher tests (the CVG one needs nothing from the LRT; the
// VCF one starts the moment the ALT set is known) while the solver wave does QUAL and the base-quality rank sum.  Every
// run is the same one-wave arithmetic whoever executes it, so records are byte-identical to the plain form
// (tests/test_gpu_parity.py::test_team_tail_...).  Hand-offs are LDS flags with bounded spins, like the rest of the kernel.
struct __attribute__((aligned(16))) BvTeam {
    BvLrtTeamShared lrt;  // LRT scratch + counting barrier of the team
    uint32_t helpers;     // tally waves that have finished streaming and wait for team jobs
    uint32_t last_site;   // the last site this workgroup streams (set by the lead tally wave once it knows)
    uint32_t gen;         // team jobs handed out so far (the job's number, from 1)
    uint32_t site, buf;   // the job; site == 0xFFFFFFFF: no more jobs
    uint32_t done;        // helper completions, NTALLY per job
    uint32_t fv_go, f_done;  // == job number once the VCF table is posted / the Fisher results are
    uint32_t vtab[4], ntab;
    uint32_t f_flags;
    double c_fs, c_sor, v_fs, v_sor;
};
#define BV_TEAM_NO_SITE 0xFFFFFFFEu
#define BV_TEAM_MAX_SITES 65536

template <bool TEAM>
__device__ __forceinline__ BvTeam *bv_team_lds() {
    __shared__ BvTeam tm;
    return &tm;
}
template <>
__device__ __forceinline__ BvTeam *bv_team_lds<false>() {
    return nullptr;
}

// bv_site_solve's work policy on the solver wave of a team job (role 0)
template <int NLRT>
struct BvTeamWork {
    BvTeam *tm;
    uint32_t job;
    uint32_t *err;
    mutable bool posted = false;
    __device__ __forceinline__ void post(const uint32_t v[4], int ntab, int lane) const {
        if (lane == 0) {
            tm->vtab[0] = v[0]; tm->vtab[1] = v[1]; tm->vtab[2] = v[2]; tm->vtab[3] = v[3];
            tm->ntab = (uint32_t)ntab;
        }
        bv_set_flag(&tm->fv_go, job);
        posted = true;
    }
    __device__ __forceinline__ void lrt(const BvBins &B, const uint32_t depth[4], uint32_t total, int nspec, int ref,
                                        double min_af, BvLrtShared *, int lane, BvLrtOut &L, uint32_t q0_mask) const {
        bv_lrt<NLRT>(B, depth, total, 0 | (1 << 3) | (2 << 6) | (3 << 9), nspec, ref, min_af, &tm->lrt.lrt, 0, lane, L, q0_mask);
    }
    __device__ __forceinline__ void tables_known(const uint32_t v[4], int ntab, int lane) const { post(v, ntab, lane); }
    __device__ __forceinline__ void strand_bias(const uint32_t[4], const uint32_t v[4], int ntab, int lane, const BvLnTab &,
                                                double &c_fs, double &c_sor, double &v_fs, double &v_sor,
                                                uint32_t &flags) const {
        if (!posted) post(v, ntab, lane);  // not a variant site: only the CVG table
        bv_wait_flag(&tm->f_done, job, err);
        c_fs = tm->c_fs; c_sor = tm->c_sor;
        if (ntab == 2) { v_fs = tm->v_fs; v_sor = tm->v_sor; }
        flags |= tm->f_flags;
    }
};

// Which sites a team takes: deep ones (the shallow-site replay, bv_em_ordered, is one wave's job) of a normal launch.
__device__ __forceinline__ bool bv_team_takes(uint32_t flags, const BvSiteSums &S) {
    const uint32_t total = S.fwd[0] + S.fwd[1] + S.fwd[2] + S.fwd[3] + S.rev[0] + S.rev[1] + S.rev[2] + S.rev[3];
    return total > (uint32_t)BV_ORD_MAX && !(flags & (BV_FLAG_TALLY_ONLY | BV_FLAG_SKIP_LRT | BV_FLAG_SKIP_FISHER));
}

// A tally wave's part in one team job.  role 1 .. NLRT-1: LRT; role NLRT: the Fisher tests.
template <int NLRT>
__device__ __forceinline__ void bv_team_help(const BvSolveArgs &a, uint32_t site, int role, uint32_t job, uint32_t *hist,
                                             uint32_t *bin_code, uint32_t *bin_cnt, BvTeam *tm, const double *tab_hit,
                                             const double *tab_miss, int lane) {
    BvSiteSums S;
    S.q0_mask = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        if (hist[b << 8] + hist[(b | 4) << 8]) S.q0_mask |= 1u << b;
    // (the bins are re-derived by every team wave: the same values land in the same words, no hand-off needed)
    bv_prologue_wave<false>(hist, bin_code, bin_cnt, lane, S.fwd, S.rev, &S.nb, &S.badq);
    bv_lrt_sync<0>();
    uint32_t depth[4], total = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        depth[b] = S.fwd[b] + S.rev[b];
        total += depth[b];
    }
    int ref = a.ref_base[site];
    if (ref > 4) ref = 4;
    if (role < NLRT) {
        BvBins B;
        B.code = bin_code; B.cnt = bin_cnt; B.skip_mask = 0u; B.hit = tab_hit; B.miss = tab_miss;
        B.loghit = a.loghit; B.logmiss = a.logmiss;
        B.nb = (int)S.nb;
        B.ord = nullptr; B.n_ord = 0;
        BvLrtOut L;
        bv_lrt<NLRT>(B, depth, total, 0 | (1 << 3) | (2 << 6) | (3 << 9), 4, ref, a.min_af, &tm->lrt.lrt, role, lane, L, S.q0_mask);
    } else {
        uint32_t t[4] = {0, 0, 0, 0};  // CVG table: alt = every non-ref base (caller.cpp:1236-1245)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b == ref) { t[0] += S.fwd[b]; t[1] += S.rev[b]; } else { t[2] += S.fwd[b]; t[3] += S.rev[b]; }
        }
        uint32_t fl = 0;
#pragma unroll 1
        for (int k = 0; k < 2; ++k) {
            if (k) {
                bv_wait_flag(&tm->fv_go, job, tm->lrt.err);
                if (tm->ntab != 2u) break;
                t[0] = tm->vtab[0]; t[1] = tm->vtab[1]; t[2] = tm->vtab[2]; t[3] = tm->vtab[3];
            }
            double fs, sor;
            bv_strand_bias_wave(t[0], t[1], t[2], t[3], lane, a.lnfact, &fs, &sor, &fl);
            if (lane == 0) {
                if (k) { tm->v_fs = fs; tm->v_sor = sor; } else { tm->c_fs = fs; tm->c_sor = sor; }
            }
        }
        if (lane == 0) tm->f_flags = fl;
        bv_set_flag(&tm->f_done, job);
    }
    bv_add_flag(&tm->done, lane);
}

// -DBV_TEAM_DEBUG (tools/experiments/r3_team_debug.sh): every workgroup of the team form stamps s_memrealtime (100 MHz) at six
// points of its life into the engine's spare counter blocks, the solver wave counts team jobs / solo solves and their cycles;
// bv_engine_wait prints the distribution.  This is how the start and the tail of a launch were taken apart.
#ifdef BV_TEAM_DEBUG
#define BV_TEAM_STAMP_INIT()                                                                                               \
    uint32_t *dbg_ = a.counters + BV_CTR_WORDS + (blockIdx.x < 640u ? blockIdx.x : 639u) * 8u; /* counter blocks 1.. are free */ \
    if (TEAM && tid == 0 && blockIdx.x == 0) a.counters[BV_CTR_WORDS + 5150] = 2u; /* whose stamps these are */                       \
    if (TEAM && tid == 0) dbg_[7] = __builtin_amdgcn_s_getreg((31 << 11) | 20) /* XCC_ID */
#define BV_TEAM_STAMP(COND, SLOT)                                                   \
    do {                                                                            \
        if (TEAM && (COND)) dbg_[SLOT] = (uint32_t)__builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define BV_TEAM_STAMP_INIT() do { } while (0)
#define BV_TEAM_STAMP(COND, SLOT) do { } while (0)
#endif

// MODE 0: the plain form.  MODE 1: the team form (its ticket rules + the helpers).  MODE 2: the team form's ticket rules and
// start-up alone (first ticket = workgroup index, no ticket reserved while the solver is behind, phred tables fetched by the
// solver wave) -- for measuring them apart from the helpers.
template <int NTALLY, int NSOLVE, bool CHAIN = false, int MODE = 0>
__global__ __launch_bounds__(BV_WAVE *(NTALLY + NSOLVE), 4) void bv_pass1_kernel(BvPass1Args a) {
    constexpr bool TEAM = MODE == 1, TK = MODE != 0;
    static_assert(!TEAM || NSOLVE == 1, "the team form has one solver wave");
    constexpr int NT = BV_WAVE * (NTALLY + NSOLVE);
    constexpr int NBUF = NSOLVE + BV_RING_EXTRA;  // the tally may run BV_RING_EXTRA sites ahead of a slow (variant-site) solve
    __shared__ BvPass1Shared<NBUF, NSOLVE> sh;
    BvTeam *const tm = bv_team_lds<TEAM>();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    BV_TEAM_STAMP_INIT();
    BV_TEAM_STAMP(tid == 0, 0);  // entry
    // one-time set-up by the whole workgroup: zero the ring, copy the phred tables
    {
        uint4 *h4 = reinterpret_cast<uint4 *>(&sh.hist[0][0]);
        for (int i = tid; i < NBUF * BV_H2_WORDS / 4; i += NT) h4[i] = make_uint4(0, 0, 0, 0);
        if (!TK) {
            for (int i = tid; i < BV_QBINS; i += NT) {
                sh.tab_hit[i] = a.tables->hit[i];
                sh.tab_miss[i] = a.tables->miss[i];
            }
        }
        if (tid < NBUF) {
            sh.published[tid] = 0u;
            sh.filled[tid] = 0u;
            sh.drained[tid] = 0u;
            sh.swz_of[tid] = 0u;
        }
        if (tid == 0) sh.dense_hint = 0u;
        if (TEAM && tid == 0) {
            tm->helpers = 0u; tm->last_site = BV_TEAM_NO_SITE; tm->gen = 0u; tm->done = 0u; tm->fv_go = 0u; tm->f_done = 0u;
            tm->lrt.bar = 0u; tm->lrt.err = &a.counters[BV_CTR_TIMEOUT];
        }
    }
    __syncthreads();
    BV_TEAM_STAMP(tid == 0, 1);  // start barrier passed

    BvSolveArgs sa;  // the solver wave's; in the team form the tally waves fill theirs when they turn helpers
#define BV_FILL_SOLVE_ARGS()                                                                              \
    do {                                                                                                  \
        sa.ref_base = a.ref_base; sa.out = a.out; sa.var_list = a.var_list; sa.counters = a.counters;     \
        sa.min_af = a.min_af; sa.flags = a.flags;                                                         \
        sa.lnfact.t = a.tables->lnfact; sa.lnfact.n = (int)a.tables->lnfact_n;                            \
        sa.loghit = a.tables->loghit; sa.logmiss = a.tables->logmiss;                                     \
        sa.bs = a.bs; sa.q = a.q; sa.pitch = a.pitch; sa.n_samples = a.n_samples;                         \
    } while (0)

    // Sites are handed out by a global ticket counter (a.counters[BV_CTR_TICKET], zeroed by the engine before
    // the launch): a workgroup that drew cheap hom-ref sites simply draws more, so the persistent
    // grid drains evenly whatever the batch size.  Which workgroup solves a site has no influence
    // on the site's record, so results stay bit-reproducible.
    if (wave < NTALLY) {
        // ------------------------------------------------ tally waves (wave 0 leads)
        // (s_setprio(1) / s_setprio(3) for this role were measured, twice: 0.5-2 % slower; priorities stay equal.)
        // Tickets are drawn `chunk` sites at a time: 1 for long rows (a reserved site is tens of microseconds of
        // work, so reserving more would lengthen the drain), 4 for short rows (one atomic per site on a single
        // address saturates at ~85 M/s).
        const uint32_t chunk = a.n_samples > 16384u ? 1u : 4u;
        // Team form: (1) a workgroup's FIRST ticket is its own index -- no atomic: the persistent grid starts all at once and
        // 1,024 draws on one address take 12 us (measured: first rows began 3-30 us after the launch); the counter hands
        // out the tickets from gridDim.x * chunk on.  (2) The next ticket is drawn ahead, under the current row's stream,
        // only while the ring has room for it: a workgroup whose solver is a full ring behind would otherwise sit on a
        // reserved site that any other workgroup could have started at once (measured on 8,192-site launches: the last
        // tally ended 85 us after the first workgroup had run out of work).
        const uint32_t t0 = TK ? gridDim.x * chunk : 0u;
        uint32_t next = TK ? blockIdx.x * chunk : 0u, cur = 0, end = 0;
        bool have = true;  // `next` holds a ticket
        if (!TK && wave == 0 && lane == 0) next = atomicAdd(&a.counters[BV_CTR_TICKET], chunk);
        for (uint32_t k = 0;; ++k) {
            const uint32_t buf = k % NBUF, gen = k / NBUF;
            uint32_t site;
            if (wave == 0) {
                if (cur == end) {
                    if (TK && !have) {  // nothing was reserved: wait for room, then draw
                        bv_wait_flag(&sh.drained[buf], gen, &a.counters[BV_CTR_TIMEOUT]);
                        if (lane == 0) next = t0 + atomicAdd(&a.counters[BV_CTR_TICKET], chunk);
                    }
                    cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)next);
                    end = cur + chunk;
                    if (TK) {
                        if (cur < a.n_sites) {
                            const uint32_t nk = k + chunk;  // the slot of the next draw's first row
                            // (not under the very first row either: the grid starts in step, the burst of draws would return
                            // -- in order -- ahead of every workgroup's first loads)
                            have = k != 0 && __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&sh.drained[nk % NBUF], __ATOMIC_RELAXED,
                                                                                                  __HIP_MEMORY_SCOPE_WORKGROUP)) == (int)(nk / NBUF);
                            if (have && lane == 0) next = t0 + atomicAdd(&a.counters[BV_CTR_TICKET], chunk);
                        }
                    } else if (cur < a.n_sites && lane == 0)
                        next = atomicAdd(&a.counters[BV_CTR_TICKET], chunk);  // in flight under these rows' stream
                }
                site = cur < a.n_sites ? cur : 0xFFFFFFFFu;
                ++cur;
#ifdef BV_TEAM_DEBUG
                const unsigned long long tw_ = __builtin_readcyclecounter();
#endif
                bv_wait_flag(&sh.drained[buf], gen, &a.counters[BV_CTR_TIMEOUT]);  // the solver has re-zeroed this slot
#ifdef BV_TEAM_DEBUG  /* cycles the tally waves of this workgroup stood still because the ring was full */
                if (TEAM && lane == 0) atomicAdd(&a.counters[BV_CTR_EASY], (uint32_t)((__builtin_readcyclecounter() - tw_) >> 6));
#endif
                // dense rows (the solver's last row had a quarter of its cells covered: coverage is a property of the cohort) take
                // the bank swizzle -- the lead wave decides for the slot, the other tally waves and the solver follow it
                const uint32_t swz_row = (a.flags & BV_FLAG_NO_DOM) ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&sh.dense_hint, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (lane == 0) { sh.site_of[buf] = site; sh.swz_of[buf] = swz_row; }
                // (BV_FLAG_FAULT_LOST_HANDOFF, tests: workgroup 0 never publishes its first slot -- the other tally waves must give
                // up after their bounded wait)
                if (!((a.flags & BV_FLAG_FAULT_LOST_HANDOFF) && blockIdx.x == 0u && k == 0u))
                    bv_set_flag(&sh.published[buf], gen + 1u);
            } else {
                bv_wait_flag(&sh.published[buf], gen + 1u, &a.counters[BV_CTR_TIMEOUT]);
                site = sh.site_of[buf];
            }
            if (site == 0xFFFFFFFFu) {
                // end of work: this slot is the first solver's marker; the leader adds the others
                bv_add_flag(&sh.filled[buf], lane);
                if (wave == 0) {
                    for (int j = 1; j < NSOLVE; ++j) {
                        const uint32_t kk = k + j, b2 = kk % NBUF, g2 = kk / NBUF;
                        bv_wait_flag(&sh.drained[b2], g2, &a.counters[BV_CTR_TIMEOUT]);
                        if (lane == 0) sh.site_of[b2] = 0xFFFFFFFFu;
                        bv_set_flag(&sh.filled[b2], (g2 + 1u) * NTALLY);
                    }
                }
                break;
            }
            const uint8_t *pb = a.bs, *pq = a.q;
            if (CHAIN && a.ch != nullptr) {  // a chained launch: the segment's (biased) planes
                const BvChainC ch = bv_chain_const(a.ch);
                const uint32_t sg = bv_chain_seg(ch, (uint32_t)__builtin_amdgcn_readfirstlane((int)site));
                pb = ch->bs[sg]; pq = ch->q[sg];
            }
            BV_TEAM_STAMP(wave == 0 && lane == 0 && k == 0, 2);  // first row begins
            if (__builtin_amdgcn_readfirstlane((int)sh.swz_of[buf]) != 0)
                bv_tally_row_wave<NTALLY, false, true>(pb + (size_t)site * a.pitch, pq + (size_t)site * a.pitch, a.n_samples, sh.hist[buf], wave, lane);
            else
                bv_tally_row_wave<NTALLY>(pb + (size_t)site * a.pitch, pq + (size_t)site * a.pitch, a.n_samples, sh.hist[buf], wave, lane);
            if (TEAM && wave == 0) {
                // Was that this workgroup's last row?  (The next ticket was drawn a row ago and has long arrived.)  The solver
                // wave then waits the microsecond it takes the tally waves to report as helpers instead of starting alone.
                if (cur == end && !have && cur < a.n_sites) {
                    // nothing reserved (first row, or the ring was full when this row began): draw now if there is room
                    const uint32_t nk = k + 1u;
                    if (__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&sh.drained[nk % NBUF], __ATOMIC_RELAXED,
                                                                              __HIP_MEMORY_SCOPE_WORKGROUP)) == (int)(nk / NBUF)) {
                        if (lane == 0) next = t0 + atomicAdd(&a.counters[BV_CTR_TICKET], chunk);
                        have = true;
                    }
                }
                bool last = cur >= a.n_sites;
                if (!last && cur == end && have) last = (uint32_t)__builtin_amdgcn_readfirstlane((int)next) >= a.n_sites;
                if (last && lane == 0) tm->last_site = site;
            }
            bv_add_flag(&sh.filled[buf], lane);  // release: this wave's ds_add of the row are done
        }
        if (TEAM) {
            // ------------------------------------------------ the tail: help the solver wave with what is left in the ring
            BV_TEAM_STAMP(wave == 0 && lane == 0, 3);  // tally waves done
            BV_FILL_SOLVE_ARGS();  // (here, not at the start: the table pointers are loads the first row must not wait for)
            bv_add_flag(&tm->helpers, lane);
            for (uint32_t job = 1;; ++job) {
                bv_wait_flag(&tm->gen, job, &a.counters[BV_CTR_TIMEOUT]);
                const uint32_t site = tm->site, buf = tm->buf;
                if (site == 0xFFFFFFFFu) break;
                if (CHAIN && a.ch != nullptr) {
                    const BvChainC ch = bv_chain_const(a.ch);
                    const uint32_t sg = bv_chain_seg(ch, (uint32_t)__builtin_amdgcn_readfirstlane((int)site));
                    sa.ref_base = ch->ref_base[sg];
                }
                bv_team_help<NTALLY>(sa, site, wave + 1, job, sh.hist[buf], sh.sv[0].bin_code, sh.sv[0].bin_cnt, tm,
                                     sh.tab_hit, sh.tab_miss, lane);
            }
        }
    } else {
        // ------------------------------------------------ solver waves
        const int s = wave - NTALLY;
        BV_FILL_SOLVE_ARGS();
#undef BV_FILL_SOLVE_ARGS
        if (TK) {
            // The phred tables are the solver's (and, later, its helpers'): fetched here, under the first row's stream.  Behind
            // the workgroup's start barrier they held every tally wave back by one loaded-memory latency (measured: 2 us on
            // the XCD that starts first, 5-12 us on the other seven -- 1,024 workgroups already have 50 MB of row loads queued).
            for (int i = lane; i < BV_QBINS; i += BV_WAVE) {
                sh.tab_hit[i] = a.tables->hit[i];
                sh.tab_miss[i] = a.tables->miss[i];
            }
            bv_lrt_sync<0>();
            BV_TEAM_STAMP(lane == 0, 4);  // phred tables in LDS
        }
        uint32_t jobs = 0;  // team jobs handed out
        for (uint32_t k = (uint32_t)s;; k += NSOLVE) {
            const uint32_t buf = k % NBUF, gen = k / NBUF;
            bv_wait_flag(&sh.filled[buf], (gen + 1u) * NTALLY, &a.counters[BV_CTR_TIMEOUT]);
            const uint32_t site = sh.site_of[buf];
            if (site == 0xFFFFFFFFu) {
                BV_TEAM_STAMP(lane == 0, 5);  // solver wave done
                if (TEAM) {  // release the helpers
                    if (lane == 0) tm->site = 0xFFFFFFFFu;
                    bv_set_flag(&tm->gen, jobs + 1u);
                }
                break;
            }
            if (CHAIN && a.ch != nullptr) {
                const BvChainC ch = bv_chain_const(a.ch);
                const uint32_t sg = bv_chain_seg(ch, (uint32_t)__builtin_amdgcn_readfirstlane((int)site));
                sa.ref_base = ch->ref_base[sg]; sa.out = ch->out[sg]; sa.bs = ch->bs[sg]; sa.q = ch->q[sg];
            }
            // a row tallied with the dense-row swizzle: back into its plain order before anybody reads it
            if (__builtin_amdgcn_readfirstlane((int)sh.swz_of[buf]) != 0) bv_hist_unswizzle<8, 8>(sh.hist[buf], lane);
            if (TEAM) {
                uint32_t *hist = sh.hist[buf], *bin_code = sh.sv[s].bin_code, *bin_cnt = sh.sv[s].bin_cnt;
                BvSolverScratch *sv = &sh.sv[s].sc;
                constexpr int REC_WORDS = (int)(sizeof(bv_site_result) / 4);
                if (lane < REC_WORDS) reinterpret_cast<uint32_t *>(&sv->res)[lane] = 0u;
                BvSiteSums S;
                S.q0_mask = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (hist[b << 8] + hist[(b | 4) << 8]) S.q0_mask |= 1u << b;
                bv_prologue_wave<false>(hist, bin_code, bin_cnt, lane, S.fwd, S.rev, &S.nb, &S.badq);
                bv_lrt_sync<0>();
                BvHqFromHist hq{hist};
                bool team = bv_team_takes(sa.flags, S);
                if (team) {
                    team = __hip_atomic_load(&tm->helpers, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == (uint32_t)NTALLY;
                    if (!team && tm->last_site == site) {  // they are on their way
                        bv_wait_flag(&tm->helpers, (uint32_t)NTALLY, &a.counters[BV_CTR_TIMEOUT]);
                        team = true;
                    }
                }
#ifdef BV_TEAM_DEBUG
                const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
                if (team) {
                    ++jobs;
                    if (lane == 0) { tm->site = site; tm->buf = buf; tm->lrt.bar = 0u; }
                    bv_set_flag(&tm->gen, jobs);
                    BvTeamWork<NTALLY> work;
                    work.tm = tm; work.job = jobs; work.err = &a.counters[BV_CTR_TIMEOUT];
                    bv_site_solve<false, BvHqFromHist, false, BvTeamWork<NTALLY>>(sa, site, S, bin_code, bin_cnt, hq, sv, sh.tab_hit,
                                                                                 sh.tab_miss, lane, work);
                    bv_wait_flag(&tm->done, jobs * (uint32_t)NTALLY, &a.counters[BV_CTR_TIMEOUT]);  // nobody reads the slot any more
                } else {
                    bv_site_solve<false>(sa, site, S, bin_code, bin_cnt, hq, sv, sh.tab_hit, sh.tab_miss, lane);
                }
#ifdef BV_TEAM_DEBUG
                if (lane == 0) {  // (long rows leave the short-row list counters free)
                    const uint32_t dt = (uint32_t)((__builtin_readcyclecounter() - t0_) >> 6);
                    atomicAdd(&a.counters[team ? BV_CTR_CANDS : BV_CTR_EASY3], 1u);
                    atomicAdd(&a.counters[(team ? BV_CTR_CANDS : BV_CTR_EASY3) + 1], dt);
                }
#endif
            } else {
                bv_solve_site_wave<false>(sa, site, (BV_LDS uint32_t *)sh.hist[buf], (BV_LDS uint32_t *)sh.sv[s].bin_code,
                                          (BV_LDS uint32_t *)sh.sv[s].bin_cnt, (BV_LDS BvSolverScratch *)&sh.sv[s].sc,
                                          (BV_LDS const double *)sh.tab_hit, (BV_LDS const double *)sh.tab_miss, lane);
            }
            // what the next rows are likely to be: this one's depth (the record is still staged in the solver's scratch)
            if (lane == 0) sh.dense_hint = (sh.sv[s].sc.res.total_depth >= (a.n_samples >> 2)) ? 1u : 0u;
            // hand the slot back, zeroed
            uint4 *h4 = reinterpret_cast<uint4 *>(sh.hist[buf]);
#pragma unroll
            for (int i = 0; i < BV_H2_WORDS / 4 / BV_WAVE; ++i) h4[i * BV_WAVE + lane] = make_uint4(0, 0, 0, 0);
            bv_set_flag(&sh.drained[buf], gen + 1u);
        }
    }
}

// (Rows of at most 49,152 samples never come here: bv_pass1_short.hip / bv_pass1_fused.hip.  Round 1's form for them -- four
// independent waves per workgroup that each tally and solve their own site -- is in docs/history/.)
template <int NTALLY, int MODE>
static void bv_launch_pass1_cfg(const BvPass1Args &a, hipStream_t stream) {
    // Persistent grid: as many workgroups as stay resident (VGPR-limited to 16 waves per CU), never more than there are sites.
    constexpr int NSOLVE = 1;
    constexpr uint32_t by_vgpr = 16u / (NTALLY + NSOLVE);
    constexpr uint32_t by_lds = (uint32_t)((160u * 1024u) / sizeof(BvPass1Shared<NSOLVE + BV_RING_EXTRA, NSOLVE>));
    uint32_t grid = (a.n_cu ? a.n_cu : 256u) * (by_vgpr < by_lds ? by_vgpr : by_lds);
    if (grid > a.n_sites) grid = a.n_sites;
    // (the chained form is an instantiation of its own: the headline kernel keeps its register allocation)
    const dim3 block(64 * (NTALLY + NSOLVE));
    if (a.ch != nullptr) hipLaunchKernelGGL((bv_pass1_kernel<NTALLY, NSOLVE, true, MODE>), dim3(grid), block, 0, stream, a);
    else hipLaunchKernelGGL((bv_pass1_kernel<NTALLY, NSOLVE, false, MODE>), dim3(grid), block, 0, stream, a);
}

void bv_launch_pass1(const BvPass1Args &a, hipStream_t stream) {
    // Three tally waves + one solver wave per workgroup, four workgroups per CU.  (Measured and dropped, round 2: <3,2> 15-30 %
    // slower, <2,1> 4 %, <7,1> and <5,1> 6 %.)  Several tally waves share a row: short per-site latency, short tail.  Up to
    // BV_TEAM_MAX_SITES per launch the team form is used: the last solves of a workgroup are spread over its idle tally waves
    // (8,192 sites: 0.351 -> 0.327 ms, 65,536: ~1 %).  Larger launches take the team form's ticket rules and start-up without
    // its helpers (MODE 2): 14 interleaved A/B runs at 131,072 sites x 100 k samples, pass 1 4.17 -> 4.12 ms (-1.1 %); with the
    // helpers too (MODE 1) it measured 0.5 % slower there.
    // (BV_FLAG_LONG_ROW_FORM(2), tests: the form without helpers whatever the launch size -- an independent realisation of the tail)
    if (a.n_sites <= (uint32_t)BV_TEAM_MAX_SITES && ((a.flags >> 8) & 0xFu) != 2u) bv_launch_pass1_cfg<3, 1>(a, stream);
    else bv_launch_pass1_cfg<3, 2>(a, stream);
}
