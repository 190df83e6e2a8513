// bv_engine.hip -- host side of the C ABI declared in include/basevar_amd.h.
//
// One engine = one HIP stream + the small device scratch the two passes share (phred
// tables, variant-site list, counters) + HIP events that time each pass on the stream it
// runs on.  No oracle, no CPU arithmetic path: if no HIP device or no gfx950 code object is
// usable, creation fails loudly with BV_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <link.h>
#include <sched.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "bv_kernels.h"

void bv_launch_synth(const bv_synth_params &p, uint32_t n_sites, uint32_t n_samples, uint64_t pitch, uint8_t *bs,
                     uint8_t *q, uint8_t *mapq, uint16_t *rpr, uint8_t *ref_base, hipStream_t stream);

namespace {
std::mutex g_err_mu;
std::string g_err;  // errors raised without an engine (create failures)

void set_global_error(const std::string &m) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = m;
}

// ---- the host libm's log() table (BvTables::hostlog, bv_log_host in bv_device.h) ----------------------------------
// The reference's EM takes log() of per-sample marginals with the host's libm (algorithm.h:243) and compares /
// truncates sums of them; at tie-prone shallow sites the last bit of log() decides the call.  The device replays those
// sites with the host's own algorithm: its data table is looked up in the libm this process has loaded, and it is used
// only if the restated algorithm agrees with log() bit for bit on every probe below.  Anything else (another libm,
// another ifunc variant) leaves the table off and the device library's log in place -- correct, ulps from the host's.
struct HostLogFind {
    const double *tab;
};

int host_log_phdr_cb(struct dl_phdr_info *info, size_t, void *user) {
    // the C math library only: file name "libm.so*" or "libm-<version>.so" (not libmagma, libmpi, ...)
    if (!info->dlpi_name) return 0;
    const char *slash = std::strrchr(info->dlpi_name, '/');
    const char *fname = slash ? slash + 1 : info->dlpi_name;
    if (std::strncmp(fname, "libm.so", 7) != 0 && std::strncmp(fname, "libm-", 5) != 0) return 0;
    const double ln2hi = 0x1.62e42fefa3800p-1, ln2lo = 0x1.ef35793c76730p-45;  // the table opens with ln 2, split
    const size_t need = sizeof(double) * BV_HOSTLOG_N;
    for (int i = 0; i < info->dlpi_phnum; ++i) {
        const ElfW(Phdr) &ph = info->dlpi_phdr[i];
        if (ph.p_type != PT_LOAD || !(ph.p_flags & PF_R) || ph.p_memsz < need) continue;
        const char *base = reinterpret_cast<const char *>(info->dlpi_addr + ph.p_vaddr);
        for (size_t o = 0; o + need <= ph.p_memsz; o += 8) {
            double d[4];
            std::memcpy(d, base + o, sizeof d);
            if (d[0] == ln2hi && d[1] == ln2lo && d[2] < -0.49 && d[2] > -0.51 && d[3] > 0.33 && d[3] < 0.34) {
                static_cast<HostLogFind *>(user)->tab = reinterpret_cast<const double *>(base + o);
                return 1;
            }
        }
    }
    return 0;
}

inline uint64_t f64_bits(double x) {
    uint64_t u;
    std::memcpy(&u, &x, 8);
    return u;
}
inline double bits_f64(uint64_t u) {
    double x;
    std::memcpy(&x, &u, 8);
    return x;
}

// bv_log_host, on the host (same operations in the same order; std::fma is a true fused multiply-add)
double host_log_restated(double x, const double *T) {
    const double *A = T + 2, *B = T + 7, *tab = T + 18;
    uint64_t ix = f64_bits(x);
    const uint64_t LO = 0x3fee000000000000ull, HI = 0x3ff1090000000000ull;
    if (ix - LO < HI - LO) {
        if (ix == 0x3ff0000000000000ull) return 0.;
        const double r = x - 1.0, r2 = r * r, r3 = r * r2;
        const double pA = std::fma(r2, B[3], std::fma(r, B[2], B[1]));
        const double pB = std::fma(r2, B[6], std::fma(r, B[5], B[4]));
        const double pC = std::fma(r3, B[10], std::fma(r2, B[9], std::fma(r, B[8], B[7])));
        const double p = std::fma(std::fma(pC, r3, pB), r3, pA);
        const double t = std::fma(r, 0x1p27, r), rhi = std::fma(-0x1p27, r, t), rlo = r - rhi;
        const double s = rhi * rhi;
        const double hi = std::fma(s, B[0], r);
        const double lo = std::fma(s, B[0], r - hi);
        const double lo2 = std::fma(B[0] * rlo, rhi + r, lo);
        return std::fma(p, r3, lo2) + hi;
    }
    const uint32_t top = (uint32_t)(ix >> 48);
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        if ((ix << 1) == 0) return -HUGE_VAL;
        if (ix == 0x7ff0000000000000ull) return x;
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return std::nan("");
        ix = f64_bits(x * 0x1p52);
        ix -= 52ull << 52;
    }
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const uint32_t i = (uint32_t)(tmp >> 45) & 127u;
    const int k = (int)((int64_t)tmp >> 52);
    const double z = bits_f64(ix - (tmp & (0xfffull << 52)));
    const double invc = tab[2 * i], logc = tab[2 * i + 1], kd = (double)k;
    const double r = std::fma(z, invc, -1.0);
    const double w = std::fma(kd, T[0], logc), hi = r + w;
    const double lo = std::fma(kd, T[1], (w - hi) + r);
    const double r2 = r * r, r3 = r * r2;
    const double p = std::fma(std::fma(r, A[4], A[3]), r2, std::fma(r, A[2], A[1]));
    return std::fma(r3, p, std::fma(r2, A[0], lo)) + hi;
}

// fills dst[0..BV_HOSTLOG_N) + the "usable" flag behind it; true when the host's log() is reproduced exactly
bool load_host_log_table_once(double *dst);
// (searched and verified once per process: every engine gets a copy of the result)
bool load_host_log_table(double *dst) {
    static std::once_flag once;
    static double cached[BV_HOSTLOG_N + 2];
    static bool ok = false;
    std::call_once(once, [] { ok = load_host_log_table_once(cached); });
    std::memcpy(dst, cached, sizeof(cached));
    return ok;
}
bool load_host_log_table_once(double *dst) {
    std::memset(dst, 0, sizeof(double) * (BV_HOSTLOG_N + 2));
    HostLogFind find{nullptr};
    dl_iterate_phdr(host_log_phdr_cb, &find);
    if (!find.tab) return false;
    std::memcpy(dst, find.tab, sizeof(double) * BV_HOSTLOG_N);
    // probes: the marginals the EM sees are mixtures of (1 - eps_q) and eps_q / 3 -- those values, fractions of them,
    // a dense sweep of the near-1 branch, every table cell's two edges, and a spread of exponents down to subnormals
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto next = [&]() {
        rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
        return rng;
    };
    auto same = [&](double x) {
        volatile double vx = x;  // keep the compiler from folding log() of a constant
        const double a = host_log_restated(x, dst), b = std::log(vx);
        return f64_bits(a) == f64_bits(b) || (a != a && b != b);
    };
    bool ok = true;
    for (int qv = 0; qv < BV_QBINS && ok; ++qv) {
        const double eps = std::exp((double)qv * -0.23025850929940458);
        ok = ok && same(1.0 - eps) && same(eps / 3);
        for (int k = 1; k < 64 && ok; ++k) ok = same((1.0 - eps) * k / 64.0) && same(eps / 3 * k / 64.0) && same((1.0 - eps) * k / 64.0 + eps / 3 * (64 - k) / 64.0);
    }
    for (int n = 0; n < 200000 && ok; ++n) {
        const double u = (double)(next() >> 11) * 0x1p-53;                    // (0, 1)
        ok = same(u) && same(0.93 + 0.14 * u) && same(std::ldexp(0.5 + 0.5 * u, -(int)(next() % 1070)));
    }
    for (uint64_t i = 0; i < 128 && ok; ++i)
        for (int d = -2; d <= 2 && ok; ++d) ok = same(bits_f64(0x3fe6000000000000ull + (i << 45) + (uint64_t)(int64_t)d));
    ok = ok && same(5e-324) && same(2.0) && same(1e300) && same(0.0) && same(-1.0) && same(HUGE_VAL);
    dst[BV_HOSTLOG_N] = ok ? 1.0 : 0.0;
    return ok;
}
}  // namespace

struct bv_engine {
    bv_engine_config cfg;
    hipStream_t stream = nullptr;      // engine-owned stream
    hipStream_t last_stream = nullptr; // stream of the last submit
    std::vector<hipStream_t> used_streams;  // every stream that carried work since the last bv_engine_wait
    hipEvent_t ev_host = nullptr;      // BV_FLAG_HOST_ORDERED: what the copy stream waits for before it reads host planes
    hipEvent_t ev_done = nullptr;      // end of the last submit: a submit on ANOTHER stream waits for it (shared scratch)
    bool ev_done_set = false;
    bool done_pending = false;         // the last submit's end is not recorded in ev_done yet (flush_done)
    hipStream_t done_stream = nullptr;
    uint32_t n_cu = 256;               // hipDeviceProp_t::multiProcessorCount
    int host_log_exact = 0;            // 1: the device replays shallow sites with the host libm's log(), verified bit-exact
    uint8_t *d_gid = nullptr;          // engine-owned copy of group_id, padded to 16 bytes with BV_NO_GROUP
    size_t d_gid_bytes = 0;
    BvTables *d_tables = nullptr;
    double *d_lnfact = nullptr;
    uint32_t *d_var_list = nullptr;
    // Counter blocks (BV_CTR_* words each, bv_kernels.h): a launch that is cut into chunks (short rows, launch_passes)
    // gives every chunk a block of its own; everything else uses block 0.
    static constexpr uint32_t kCtrBlocks = 8;
    uint32_t *d_counters = nullptr;    // [kCtrBlocks][BV_CTR_WORDS]
    uint32_t *h_counters = nullptr;    // pinned host mirror
    uint32_t last_blocks = 1;          // blocks the last launch used (their VARIANTS words add up to its variant count)
    uint32_t last_ctr_base = 0;        // ... starting at this block
    // Submits take the counter blocks in turn (launch i: block i % kCtrBlocks): all blocks' per-launch lines are zeroed by ONE
    // 2-D fill every kCtrBlocks launches, and the host mirror is filled by bv_engine_wait, not by a copy behind every submit.
    // (Measured: the 23 KB device-to-host copy behind each submit kept the next submit's first kernel waiting ~10 us --
    // 100 k sites x 10 k samples 157.7 -> 160.4 M sites/s without it, 8,192-site batches 51.9 -> 55.7 M.)
    uint32_t ctr_rot = 0;
    bool ctr_mirror_stale = false;     // the device counters are ahead of h_counters
    static constexpr int kRing = 256;
    hipEvent_t ring[kRing][4] = {};    // per-submit events: start, end of pass 1, end of pass 2, [3] end of the streaming kernel of pass 1
    bool ring_one_kernel[kRing] = {};  // pass 1 was ONE kernel (long rows): [3] was not recorded, its time is [0] -> [1]
    int ring_head = 0, ring_count = 0; // pending (not yet accumulated) triplets
    int last_slot = -1;
    uint32_t n_launches = 0;           // launches since creation (BV_FLAG_SPARSE_TIMING times every eighth)
    uint32_t last_form = 0;            // BV_FORM_* bits of the last launch (bv_engine_last_launch_form)
    double acc1_ms = 0., acc2_ms = 0., acc_stream_ms = 0.;
    // short rows (bv_pass1_short.hip): HBM scratch between the streaming kernel and the solve kernel
    BvSiteSummary *d_summ = nullptr;
    uint32_t *d_bins = nullptr, *d_cand_list = nullptr, *d_easy_list = nullptr, *d_easy3_list = nullptr, *d_ovf = nullptr;
    uint32_t short_sites = 0;          // sites the short-row scratch holds
    uint32_t *d_gitems = nullptr;      // pop-group calls handed from the pass-2 tally kernels to bv_p2g_solve16_kernel
    uint32_t gitem_cap = 0;            // items (of BV_P2G_ITEM_WORDS words) d_gitems holds
    uint8_t *d_gidp = nullptr;         // group ids prepared for bv_p2g_stream_kernel (bv_launch_gid_prepare)
    // more than BV_GROUPS_PER_ROUND pop-groups: pass 2 runs once per round of groups, on the round's own view of the group plane
    // (groups of other rounds read as "no group") and into records of its own, which are then moved to their columns of `gout`
    uint8_t *d_gid_round = nullptr;
    size_t d_gid_round_bytes = 0;
    bv_group_result *d_gout_round = nullptr;
    size_t d_gout_round_bytes = 0;
    BvChain *d_chain = nullptr;        // segment tables of chained launches (bv_engine_submit_many)
    uint8_t *d_ref_cat = nullptr;      // chained short-row launches: reference bases / records of all segments, contiguous
    bv_site_result *d_out_cat = nullptr;
    unsigned chain_next = 0;
    size_t d_gidp_bytes = 0;
    uint32_t acc_n = 0;
    bool submitted = false;
    // Host buffers (BV_MEM_HOST slabs, tiles, record buffers) go through a ring of device staging buffers filled by a
    // copy stream of their own: the PCIe copy of submit / tile k+1 runs under the kernels of k, and a buffer is reused
    // only after the work that read it has finished (events).
    static constexpr int kStage = 4;
    struct StageSlot {
        void *buf = nullptr;
        size_t bytes = 0;
        hipEvent_t copied = nullptr, freed = nullptr;
        hipStream_t cs = nullptr;  // the copy stream that fills this slot
        bool used = false;
    };
    StageSlot sring[kStage];
    unsigned sring_next = 0;
    bv_site_result *stage_out = nullptr;
    bv_group_result *stage_gout = nullptr;
    bv_site_result *host_out = nullptr;
    bv_group_result *host_gout = nullptr;
    size_t host_out_bytes = 0, host_gout_bytes = 0;
    // sample-axis tile mode
    uint32_t *tile_state = nullptr;
    uint32_t *tile_maxr = nullptr;
    size_t tile_state_bytes = 0, tile_maxr_bytes = 0;
    uint32_t tile_sites = 0, tile_groups = 0, tile_stride = 0, tile_samples_total = 0, tile_samples_seen = 0;
    uint32_t tile_rank_win = 1024, tile_hg_off = 0, tile_ord_off = 0;
    uint32_t *tile_ovf = nullptr;      // pool of read-position ranks beyond the window (bv_tiles.hip), kOvfCap entries
    static constexpr uint32_t kOvfCap = 1u << 22;
    bool tile_ranks = false, tile_open = false;
    uint32_t tile_layout = 0;          // bv_slab.layout of the job's tiles (every tile of a job has the first one's)
    bool tile_layout_set = false;
    bool j_filled = false;             // joined rows: the columns not yet delivered hold "uncovered" (a packed tile scatters into them)
    hipStream_t copy_stream[2] = {nullptr, nullptr};  // alternate: the set-up of one copy hides under the transfer of the other
    // joined-rows realisation of the tile mode: resident planes [tile_sites][j_pitch]
    bool tile_join = false;
    uint8_t *j_buf = nullptr;
    size_t j_bytes = 0, j_pitch = 0, j_o_q = 0, j_o_mq = 0, j_o_rp = 0, j_o_gid = 0;
    // BV_FLAG_LANES: device-resident submits alternate between two child engines (streams and scratch of their own), so that
    // the solve kernels of one submit run under the streaming kernels of the next; the parent runs no kernels then
    static constexpr int kMaxLanes = 4;
    bv_engine *lane[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
    int n_lanes = 2;                   // (BASEVAR_AMD_LANES: tuning runs)
    unsigned lane_next = 0;
    int last_lane = -1;
    bool is_lane = false;
    hipEvent_t ev_entry = nullptr;     // what a lane waits for: the caller's stream at the time of the submit
    // bv_engine_tiles_add_many: descriptor tables, a ring of pinned host + device buffers
    static constexpr int kDescRing = 4;
    BvTileScatterPlane *h_desc[kDescRing] = {}, *d_desc[kDescRing] = {};
    hipEvent_t ev_desc[kDescRing] = {};
    bool desc_used[kDescRing] = {};
    unsigned desc_next = 0;
    mutable std::mutex mu;
    std::string err;
};

namespace {
int fail(bv_engine *e, int code, const std::string &msg) {
    if (e) {
        std::lock_guard<std::mutex> lk(e->mu);
        e->err = msg;
    } else {
        set_global_error(msg);
    }
    return code;
}
#define BV_HIP(e, call)                                                                         \
    do {                                                                                        \
        hipError_t _s = (call);                                                                 \
        if (_s != hipSuccess)                                                                   \
            return fail((e), BV_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_s));    \
    } while (0)
}  // namespace

namespace {
// fold every completed pending triplet into the accumulators; `block` waits for them
int drain_timings(bv_engine *e, bool block) {
    while (e->ring_count > 0) {
        int slot = (e->ring_head - e->ring_count + bv_engine::kRing * 2) % bv_engine::kRing;
        hipEvent_t *t = e->ring[slot];
        if (block) {
            BV_HIP(e, hipEventSynchronize(t[2]));
        } else if (hipEventQuery(t[2]) != hipSuccess) {
            break;
        }
        float a = 0.f, b = 0.f, c = 0.f;
        BV_HIP(e, hipEventElapsedTime(&a, t[0], t[1]));
        BV_HIP(e, hipEventElapsedTime(&b, t[1], t[2]));
        if (e->ring_one_kernel[slot]) c = a;
        else BV_HIP(e, hipEventElapsedTime(&c, t[0], t[3]));
        e->acc1_ms += a;
        e->acc2_ms += b;
        e->acc_stream_ms += c;
        e->acc_n += 1;
        e->ring_count -= 1;
    }
    return BV_OK;
}
}  // namespace


namespace {
// The engine's scratch (variant list, counters, staging) is shared by its submits, so work of one engine is
// serialised even when the caller alternates streams: a submit on a stream other than the previous one first
// waits for the end of the previous submit.  Every stream used is remembered for bv_engine_wait.
int flush_done(bv_engine *e) {
    if (e->done_pending) {
        BV_HIP(e, hipEventRecord(e->ev_done, e->done_stream));
        e->done_pending = false;
        e->ev_done_set = true;
    }
    return BV_OK;
}
int use_stream(bv_engine *e, hipStream_t st) {
    bool seen = false;
    for (hipStream_t u : e->used_streams) seen |= (u == st);
    if (!seen) e->used_streams.push_back(st);
    if (st != e->last_stream) {
        int rc = flush_done(e);
        if (rc != BV_OK) return rc;
        if (e->ev_done_set) BV_HIP(e, hipStreamWaitEvent(st, e->ev_done, 0));
    }
    e->last_stream = st;
    return BV_OK;
}
// The end of a submit is marked lazily: the event is recorded only when somebody needs it -- a submit on another stream,
// bv_engine_join -- and then on the stream that carried the submit (whatever the caller queued there since is waited for too:
// conservative, never wrong).  A host that keeps to one stream pays for no event at all (an event record behind every submit
// was one more packet in front of the next submit's first kernel).
int mark_done(bv_engine *e, hipStream_t st) {
    e->done_pending = true;
    e->done_stream = st;
    return BV_OK;
}
// group ids as the kernels read them: 16-byte chunks up to round_up(n_samples, 16) -- the ABI promises only
// [n_samples] bytes of any alignment, so the engine keeps its own padded copy (n_samples bytes per submit)
int stage_group_ids(bv_engine *e, const uint8_t *gid, uint32_t n_samples, bool host, hipStream_t st, const uint8_t **out) {
    const size_t need = (((size_t)n_samples + 255) & ~(size_t)255) + 256;
    if (need > e->d_gid_bytes) {
        if (e->d_gid) BV_HIP(e, hipFree(e->d_gid));
        e->d_gid = nullptr; e->d_gid_bytes = 0;
        BV_HIP(e, hipMalloc(&e->d_gid, need));
        e->d_gid_bytes = need;
    }
    BV_HIP(e, hipMemsetAsync(e->d_gid, 0xFF, need, st));
    BV_HIP(e, hipMemcpyAsync(e->d_gid, gid, n_samples, host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
    *out = e->d_gid;
    return BV_OK;
}
// ---- staging ring
int stage_acquire(bv_engine *e, size_t bytes, bv_engine::StageSlot **out) {
    for (auto &cs : e->copy_stream)
        if (!cs) BV_HIP(e, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    const unsigned k = e->sring_next++;
    bv_engine::StageSlot &sl = e->sring[k % bv_engine::kStage];
    sl.cs = e->copy_stream[k & 1u];
    if (!sl.copied) {
        BV_HIP(e, hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
        BV_HIP(e, hipEventCreateWithFlags(&sl.freed, hipEventDisableTiming));
    }
    if (bytes > sl.bytes) {
        if (sl.buf) BV_HIP(e, hipFree(sl.buf));  // synchronises with work that still uses it
        sl.buf = nullptr; sl.bytes = 0; sl.used = false;
        BV_HIP(e, hipMalloc(&sl.buf, bytes));
        sl.bytes = bytes;
    }
    // the buffer is free again once the work that read it last has run
    if (sl.used) BV_HIP(e, hipStreamWaitEvent(sl.cs, sl.freed, 0));
    *out = &sl;
    return BV_OK;
}
// BV_FLAG_HOST_ORDERED: the copies of host planes wait for everything queued on the caller's stream so far (a caller that
// fills its pinned planes with asynchronous work on that stream).  Default: host planes are complete when the call is made
// (include/basevar_amd.h) and the copy runs ahead, under the kernels of earlier submits.
int stage_order(bv_engine *e, bv_engine::StageSlot *sl, hipStream_t st) {
    if (!(e->cfg.flags & BV_FLAG_HOST_ORDERED)) return BV_OK;
    if (!e->ev_host) BV_HIP(e, hipEventCreateWithFlags(&e->ev_host, hipEventDisableTiming));
    BV_HIP(e, hipEventRecord(e->ev_host, st));
    BV_HIP(e, hipStreamWaitEvent(sl->cs, e->ev_host, 0));
    return BV_OK;
}
int stage_publish(bv_engine *e, bv_engine::StageSlot *sl, hipStream_t st) {  // the copies are queued: `st` may read after them
    BV_HIP(e, hipEventRecord(sl->copied, sl->cs));
    BV_HIP(e, hipStreamWaitEvent(st, sl->copied, 0));
    return BV_OK;
}
int stage_release(bv_engine *e, bv_engine::StageSlot *sl, hipStream_t st) {  // everything queued on `st` so far is the last reader
    BV_HIP(e, hipEventRecord(sl->freed, st));
    sl->used = true;
    return BV_OK;
}
// Host planes that lie in ONE allocation, one after the other (a packed tile, see bv_tile_packed_layout), go over the
// link as one copy: the per-copy cost of five 2-3 MB copies per tile was a quarter of the tile's transfer time.
struct HostSpan {
    const uint8_t *lo = nullptr, *hi = nullptr;
    void add(const void *p, size_t n) {
        if (!p || !n) return;
        const uint8_t *a = static_cast<const uint8_t *>(p);
        if (!lo || a < lo) lo = a;
        if (!hi || a + n > hi) hi = a + n;
    }
    size_t bytes() const { return (size_t)(hi - lo); }
};
struct HostPlane {
    const void *src;     // host pointer (may be NULL: plane absent)
    size_t bytes;
    const uint8_t *dev;  // out: where the plane lives in the staging buffer
};
// Queue the host->device copies of `n` planes into a fresh staging slot (+ `extra` bytes of device scratch behind them).
int stage_host_planes(bv_engine *e, HostPlane *pl, int n, size_t extra, bv_engine::StageSlot **slot_out, uint8_t **extra_dev, hipStream_t st) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    HostSpan sp;
    size_t sum = 0;
    for (int i = 0; i < n; ++i) {
        if (!pl[i].src || !pl[i].bytes) continue;
        sp.add(pl[i].src, pl[i].bytes);
        sum += up(pl[i].bytes);
    }
    // ONE copy only for the exact layout of bv_tile_packed_layout: the planes in order, each at the 256-aligned end of the one
    // before it, in one allocation -- then every byte of [lo, hi) is the caller's.  (Planes that merely lie close together
    // are copied one by one: the bytes between them are not ours to read.)
    bool one_copy = sp.lo != nullptr && (reinterpret_cast<uintptr_t>(sp.lo) & 15u) == 0;
    {
        size_t at = 0;
        for (int i = 0; i < n && one_copy; ++i) {
            if (!pl[i].src || !pl[i].bytes) continue;
            one_copy = static_cast<const uint8_t *>(pl[i].src) == sp.lo + at;
            at += up(pl[i].bytes);
        }
    }
    const size_t planes_bytes = one_copy ? up(sp.bytes()) : sum;
    bv_engine::StageSlot *sl = nullptr;
    int rc = stage_acquire(e, planes_bytes + up(extra), &sl);
    if (rc != BV_OK) return rc;
    rc = stage_order(e, sl, st);
    if (rc != BV_OK) return rc;
    uint8_t *base = static_cast<uint8_t *>(sl->buf);
    if (one_copy) {
        BV_HIP(e, hipMemcpyAsync(base, sp.lo, sp.bytes(), hipMemcpyHostToDevice, sl->cs));
        for (int i = 0; i < n; ++i)
            pl[i].dev = (pl[i].src && pl[i].bytes) ? base + (static_cast<const uint8_t *>(pl[i].src) - sp.lo) : nullptr;
    } else {
        size_t off = 0;
        for (int i = 0; i < n; ++i) {
            pl[i].dev = nullptr;
            if (!pl[i].src || !pl[i].bytes) continue;
            BV_HIP(e, hipMemcpyAsync(base + off, pl[i].src, pl[i].bytes, hipMemcpyHostToDevice, sl->cs));
            pl[i].dev = base + off;
            off += up(pl[i].bytes);
        }
    }
    *slot_out = sl;
    if (extra_dev) *extra_dev = base + planes_bytes;
    return BV_OK;
}
}  // namespace

extern "C" {

const char *bv_version(void) { return "basevar_amd 0.2 abi2 gfx950"; }

double bv_min_af(uint32_t n_samples, float user_min_af) {
    // src/basetype_caller.cpp:122: min_af = std::min(float(100)/input_bf.size(), min_af)
    float a = float(100) / n_samples;
    float m = (a < user_min_af) ? a : user_min_af;
    return (double)m;
}

int bv_engine_create(const bv_engine_config *cfg, bv_engine **out) {
    if (!cfg || !out) return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_create: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, BV_ERR_NO_DEVICE, "bv_engine_create: no HIP device visible (the engine has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_create: device ordinal out of range");
    if (cfg->max_sites == 0) return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_create: max_sites == 0");
    // the fault-injection bit (basevar_amd_diag.h) makes every launch stall ~2 s and fail: refused unless the process asks for it
    if ((cfg->flags & BV_FLAG_FAULT_LOST_HANDOFF) && !std::getenv("BASEVAR_AMD_FAULT_INJECT"))
        return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_create: BV_FLAG_FAULT_LOST_HANDOFF is a test-only fault injection; set BASEVAR_AMD_FAULT_INJECT=1 to allow it");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess)
        return fail(nullptr, BV_ERR_NO_DEVICE, "bv_engine_create: cannot query device");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, BV_ERR_NO_DEVICE,
                    std::string("bv_engine_create: kernels are built for gfx950 only, device is ") + prop.gcnArchName);

    bv_engine *e = new bv_engine();
    e->cfg = *cfg;
    auto bail = [&](int code) {
        std::string m = e->err;
        bv_engine_destroy(e);
        set_global_error(m);
        return code;
    };
#define BV_TRY(call)                                                                                  \
    do {                                                                                              \
        hipError_t _s = (call);                                                                       \
        if (_s != hipSuccess) {                                                                       \
            e->err = std::string(#call) + ": " + hipGetErrorString(_s);                               \
            return bail(BV_ERR_HIP);                                                                  \
        }                                                                                             \
    } while (0)
    BV_TRY(hipSetDevice(cfg->device));
    e->n_cu = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    BV_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    BV_TRY(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
    for (auto &tri : e->ring)
        for (auto &ev : tri) BV_TRY(hipEventCreate(&ev));
    BV_TRY(hipMalloc(&e->d_tables, sizeof(BvTables)));
    BV_TRY(hipMalloc(&e->d_var_list, sizeof(uint32_t) * (size_t)cfg->max_sites));
    BV_TRY(hipMalloc(&e->d_counters, sizeof(uint32_t) * BV_CTR_WORDS * bv_engine::kCtrBlocks));
    BV_TRY(hipHostMalloc(&e->h_counters, sizeof(uint32_t) * BV_CTR_WORDS * bv_engine::kCtrBlocks));
    std::memset(e->h_counters, 0, sizeof(uint32_t) * BV_CTR_WORDS * bv_engine::kCtrBlocks);
    BV_TRY(hipMemset(e->d_counters, 0, sizeof(uint32_t) * BV_CTR_WORDS * bv_engine::kCtrBlocks));

    // eps table with the host libm, exactly the reference's expression (basetype.cpp:47-48, :63)
    BvTables t;
    const double MLN10TO10 = -0.23025850929940458;  // basetype.h:20
    for (int qv = 0; qv < BV_QBINS; ++qv) {
        double epsilon = exp((double)qv * MLN10TO10);
        t.hit[qv] = 1.0 - epsilon;
        t.miss[qv] = epsilon / 3;
        // log of the two likelihood values with the host libm (algorithm.h:243 takes log of exactly these when a
        // subset holds one base); log(0) = -inf at phred 0 is never read (such sites take the iterative path)
        t.loghit[qv] = log(t.hit[qv]);
        t.logmiss[qv] = log(t.miss[qv]);
    }
    e->host_log_exact = load_host_log_table(t.hostlog) ? 1 : 0;
    if (!e->host_log_exact) {
        // not silent: said once per process on stderr, kept as the global message (bv_last_error(NULL)) of this successful
        // create, and flagged per record (BV_SITE_LOG_APPROX) on the sites it concerns
        static const char *note =
            "basevar_amd: note: the host libm's log() could not be reproduced on the device (no glibc __log_data table found, or it "
            "failed verification); sites of <= 64 covered samples use the device library's log(): values within 1e-6, exact ties "
            "between allele subsets undecided (records carry BV_SITE_LOG_APPROX)";
        static std::once_flag said;
        std::call_once(said, [] { if (!std::getenv("BASEVAR_AMD_QUIET")) std::fprintf(stderr, "%s\n", note); });
        set_global_error(note);
    }
    // log-factorials for the Fisher test with the host libm -- kfunc.c:197-201 calls lgamma(n + 1) --
    // for every depth a site of this engine can reach (deeper tables fall back to a series on the device)
    {
        size_t nfact = (size_t)(cfg->max_samples ? cfg->max_samples : 1u << 20) + 2;
        if (nfact < (1u << 16)) nfact = 1u << 16;
        if (nfact > (1u << 23)) nfact = 1u << 23;
        std::vector<double> lf(nfact);
        for (size_t k = 0; k < nfact; ++k) {
            int sign;
            lf[k] = lgamma_r((double)k + 1.0, &sign);
        }
        BV_TRY(hipMalloc(&e->d_lnfact, sizeof(double) * nfact));
        BV_TRY(hipMemcpy(e->d_lnfact, lf.data(), sizeof(double) * nfact, hipMemcpyHostToDevice));
        t.lnfact = e->d_lnfact;
        t.lnfact_n = (uint32_t)nfact;
        t.pad_ = 0;
    }
    BV_TRY(hipMemcpy(e->d_tables, &t, sizeof(t), hipMemcpyHostToDevice));
#undef BV_TRY
    if (cfg->flags & BV_FLAG_LANES) {
        // the two lanes now, not at their first submit: a process has few hardware queues and streams are dealt to them in
        // the order they are created -- created late, the lanes landed on queues already carrying other streams
        if (const char *nl = std::getenv("BASEVAR_AMD_LANES")) e->n_lanes = std::max(2, std::min((int)bv_engine::kMaxLanes, std::atoi(nl)));
        for (int k = 0; k < e->n_lanes; ++k) {
            bv_engine_config c = *cfg;
            c.flags &= ~BV_FLAG_LANES;
            const int rc = bv_engine_create(&c, &e->lane[k]);
            if (rc != BV_OK) {
                const std::string m = bv_last_error(nullptr);
                bv_engine_destroy(e);
                set_global_error("bv_engine_create: lane engine: " + m);
                return rc;
            }
            e->lane[k]->is_lane = true;
        }
    }
    *out = e;
    return BV_OK;
}

int bv_engine_destroy(bv_engine *e) {
    if (!e) return BV_OK;
    (void)hipSetDevice(e->cfg.device);
    for (bv_engine *&l : e->lane) {
        if (l) (void)bv_engine_destroy(l);
        l = nullptr;
    }
    if (e->ev_entry) (void)hipEventDestroy(e->ev_entry);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (hipStream_t st : e->used_streams) (void)hipStreamSynchronize(st);
    for (auto &tri : e->ring)
        for (auto &ev : tri)
            if (ev) (void)hipEventDestroy(ev);
    if (e->d_tables) (void)hipFree(e->d_tables);
    if (e->d_lnfact) (void)hipFree(e->d_lnfact);
    if (e->d_var_list) (void)hipFree(e->d_var_list);
    if (e->d_counters) (void)hipFree(e->d_counters);
    if (e->d_gid) (void)hipFree(e->d_gid);
    if (e->d_summ) (void)hipFree(e->d_summ);
    if (e->d_bins) (void)hipFree(e->d_bins);
    if (e->d_cand_list) (void)hipFree(e->d_cand_list);
    if (e->d_easy_list) (void)hipFree(e->d_easy_list);
    if (e->d_easy3_list) (void)hipFree(e->d_easy3_list);
    if (e->d_ovf) (void)hipFree(e->d_ovf);
    if (e->d_gitems) (void)hipFree(e->d_gitems);
    if (e->d_gidp) (void)hipFree(e->d_gidp);
    if (e->d_chain) (void)hipFree(e->d_chain);
    if (e->d_ref_cat) (void)hipFree(e->d_ref_cat);
    if (e->d_out_cat) (void)hipFree(e->d_out_cat);
    if (e->ev_done) (void)hipEventDestroy(e->ev_done);
    if (e->ev_host) (void)hipEventDestroy(e->ev_host);
    if (e->h_counters) (void)hipHostFree(e->h_counters);
    for (auto &sl : e->sring) {
        if (sl.buf) (void)hipFree(sl.buf);
        if (sl.copied) (void)hipEventDestroy(sl.copied);
        if (sl.freed) (void)hipEventDestroy(sl.freed);
    }
    for (int i = 0; i < bv_engine::kDescRing; ++i) {
        if (e->h_desc[i]) (void)hipHostFree(e->h_desc[i]);
        if (e->d_desc[i]) (void)hipFree(e->d_desc[i]);
        if (e->ev_desc[i]) (void)hipEventDestroy(e->ev_desc[i]);
    }
    if (e->d_gid_round) (void)hipFree(e->d_gid_round);
    if (e->d_gout_round) (void)hipFree(e->d_gout_round);
    if (e->tile_state) (void)hipFree(e->tile_state);
    if (e->tile_maxr) (void)hipFree(e->tile_maxr);
    if (e->tile_ovf) (void)hipFree(e->tile_ovf);
    if (e->j_buf) (void)hipFree(e->j_buf);
    for (hipStream_t cs : e->copy_stream)
        if (cs) (void)hipStreamSynchronize(cs);
    for (hipStream_t cs : e->copy_stream)
        if (cs) (void)hipStreamDestroy(cs);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return BV_OK;
}

// The two passes over device-resident planes + the copies back (records to a host caller, counters).
//
// (Round 3 also ran short-row batches as a software pipeline of chunks over two streams -- the solve kernels of chunk c under the
// streaming kernel of chunk c + 1.  Measured a loss at every size (beside a streaming kernel the solve kernels get one
// workgroup per CU and run 3 x longer, the streaming kernel slows by 50-70 %): removed; docs/history/DESIGN_round3.md 4.2b.)
static int launch_passes(bv_engine *e, const uint8_t *bs, const uint8_t *q, const uint8_t *mq, const uint16_t *rp,
                         const uint8_t *refb, const uint8_t *gid, size_t P, uint32_t n_sites, uint32_t n_samples, uint32_t n_groups,
                         bv_site_result *dout, bv_group_result *dgout, hipStream_t st, const BvChain *chain = nullptr /* device */,
                         bool chain_cat = false /* chained short rows: refb / dout are contiguous copies */, uint32_t layout = 0 /* BV_SLAB_* */) {
    const uint32_t rpr_tag = (layout & BV_SLAB_RPR_TAGGED) && rp != nullptr ? 1u : 0u;
    const size_t S = n_sites, G = n_groups;
    // The group kernels hold one (base, phred) histogram per group in LDS and are built for at most BV_GROUPS_PER_ROUND of them;
    // the reference takes any number of groups (a std::map, basetype_caller.cpp:372-410).  More groups run as ROUNDS of pass 2:
    // round r sees groups [r x 32, r x 32 + 32) only (bv_launch_gid_round) and writes [S][32] records of its own, which one 2-D
    // copy moves to their columns of the caller's [S][G] array.  Only round 0 forms the rank sums.
    const size_t Gr = G < BV_GROUPS_PER_ROUND ? G : (size_t)BV_GROUPS_PER_ROUND, n_rounds = G ? (G + Gr - 1) / Gr : 1;
    if (n_rounds > 1 && chain != nullptr) return fail(e, BV_ERR_INVALID_ARG, "launch_passes: more than 32 pop-groups do not chain");
    if (G && chain == nullptr) BV_HIP(e, hipMemsetAsync(dgout, 0, S * G * sizeof(bv_group_result), st));  // (chained: per segment, by the caller)
    if (n_rounds > 1) {
        const size_t gb = (((size_t)n_samples + 255) & ~(size_t)255) + 256, ob = S * Gr * sizeof(bv_group_result);
        if (gb > e->d_gid_round_bytes) {
            if (e->d_gid_round) BV_HIP(e, hipFree(e->d_gid_round));
            e->d_gid_round = nullptr; e->d_gid_round_bytes = 0;
            BV_HIP(e, hipMalloc(&e->d_gid_round, gb));
            e->d_gid_round_bytes = gb;
        }
        if (ob > e->d_gout_round_bytes) {
            if (e->d_gout_round) BV_HIP(e, hipFree(e->d_gout_round));
            e->d_gout_round = nullptr; e->d_gout_round_bytes = 0;
            BV_HIP(e, hipMalloc(&e->d_gout_round, ob));
            e->d_gout_round_bytes = ob;
        }
    }

    // Per-pass timing: four event records per launch (start, end of the streaming kernel, end of pass 1, end of pass 2).  They are
    // not free -- each is a packet the next kernel queues behind: ~15 us per launch together (measured: 100 k sites x 10 k samples
    // 158.4 -> 162.4 M sites/s without them, 8,192-site batches 56.7 -> 63.0 M) -- so BV_FLAG_SPARSE_TIMING records them for one
    // launch in eight; the averages of bv_engine_timing_get then rest on those launches.
    const bool timed = !(e->cfg.flags & BV_FLAG_SPARSE_TIMING) || (e->n_launches % 8u) == 0u;
    e->n_launches += 1;
    hipEvent_t *ev = nullptr;
    if (timed) {
        if (e->ring_count == bv_engine::kRing) {
            int rc = drain_timings(e, true);  // ring full: fold the oldest submits first
            if (rc != BV_OK) return rc;
        }
        const int slot = e->ring_head;
        e->ring_head = (e->ring_head + 1) % bv_engine::kRing;
        e->ring_count += 1;
        e->last_slot = slot;
        ev = e->ring[slot];
        e->ring_one_kernel[slot] = false;
    }
    // Rows of at most BV_SHORT_ROW_MAX samples take the short-row forms of pass 1 (bv_pass1_fused.hip; bv_pass1_short.hip)
    const bool two_kernel = n_samples <= BV_SHORT_ROW_MAX;

    // ---- pass-2 arguments common to every chunk; scratch of the pop-group calls
    BvPass2Args a2;
    a2.bs = bs; a2.q = q; a2.mapq = mq; a2.rpr = rp; a2.ref_base = refb; a2.group_id = gid; a2.pitch = P;
    a2.n_sites = n_sites; a2.n_samples = n_samples; a2.n_groups = n_groups;
    a2.min_af = e->cfg.min_af; a2.tables = e->d_tables; a2.out = dout; a2.gout = dgout;
    a2.var_list = e->d_var_list; a2.counters = e->d_counters; a2.n_cu = e->n_cu; a2.flags = e->cfg.flags;
    a2.gitems = nullptr; a2.gitem_cap = 0; a2.gidp = nullptr;
    a2.ch = chain;
    a2.ch_cat = chain_cat ? 1u : 0u;
    a2.rpr_tag = rpr_tag;
    if (G && gid && dgout && !(e->cfg.flags & BV_FLAG_GROUP_INLINE)) {
        // scratch for the group calls of the variant sites (1.5 KiB per site x group), grown on demand and capped at 8 GiB:
        // the variant sites past the cap keep the one-wave-per-group solver inside the tally kernel.  An allocation that
        // fails is retried at half the size (the inline path takes what the scratch cannot).
        const uint64_t want64 = (uint64_t)S * Gr, most = (8192ull << 20) / (sizeof(uint32_t) * BV_P2G_ITEM_WORDS);
        uint32_t want = (uint32_t)(want64 < most ? want64 : most);
        if (want > e->gitem_cap) {
            if (e->d_gitems) BV_HIP(e, hipFree(e->d_gitems));
            e->d_gitems = nullptr; e->gitem_cap = 0;
            for (;;) {
                if (hipMalloc(&e->d_gitems, sizeof(uint32_t) * BV_P2G_ITEM_WORDS * (size_t)want) == hipSuccess) {
                    e->gitem_cap = want;
                    break;
                }
                (void)hipGetLastError();
                e->d_gitems = nullptr;
                if (want < 1024u) break;  // no scratch at all: every group is solved inside the tally kernel
                want /= 2u;
            }
        }
        a2.gitems = e->d_gitems; a2.gitem_cap = e->gitem_cap;
        // short rows: the group plane as the streaming group tally wants it
        const size_t n16 = ((size_t)n_samples + 15) & ~(size_t)15;
        if (n16 > e->d_gidp_bytes) {
            if (e->d_gidp) BV_HIP(e, hipFree(e->d_gidp));
            e->d_gidp = nullptr; e->d_gidp_bytes = 0;
            BV_HIP(e, hipMalloc(&e->d_gidp, n16 + 256));
            e->d_gidp_bytes = n16;
        }
        a2.gidp = e->d_gidp;
        // the group plane as the perm-form tallies read it (short rows: bv_p2g_stream_kernel; long rows: bv_p2_fast_sweep)
        // (several rounds of groups: prepared per round, below)
        if (n_rounds == 1) {
            bv_launch_gid_prepare(gid, e->d_gidp, (uint32_t)n16, n_groups, st);
            BV_HIP(e, hipGetLastError());
        }
    }
    if (n_rounds > 1) a2.n_groups = (uint32_t)Gr;  // (what the kernel-selection predicates below see)

    e->last_blocks = 1;
#ifdef BV_TEAM_DEBUG
    const bool rotate = false;  // (the instrumented builds keep their stamps in the blocks behind the first)
#else
    const bool rotate = true;
#endif
    uint32_t cb = 0;  // this launch's first counter block
    if (rotate) {
        cb = e->ctr_rot % bv_engine::kCtrBlocks;
        if (cb == 0)  // a new round: the per-launch lines of every block (not the sticky error counters behind them) in one fill
            BV_HIP(e, hipMemset2DAsync(e->d_counters, sizeof(uint32_t) * BV_CTR_WORDS, 0, sizeof(uint32_t) * BV_CTR_PER_LAUNCH * BV_CTR_STRIDE,
                                       bv_engine::kCtrBlocks, st));
        e->ctr_rot += 1;
    } else {
        e->ctr_rot = 0;  // (the next rotating launch starts a round of its own)
        BV_HIP(e, hipMemsetAsync(e->d_counters, 0, sizeof(uint32_t) * BV_CTR_PER_LAUNCH * BV_CTR_STRIDE, st));
    }
    e->last_ctr_base = cb;
    a2.counters = e->d_counters + (size_t)cb * BV_CTR_WORDS;
    bool pass2_fused = false;  // pass 1's kernel has streamed the pass-2 rows too (bv_pass1_fused.hip)
    e->last_form = two_kernel ? BV_FORM_SHORT_ROWS : 0u;
    if (ev) BV_HIP(e, hipEventRecord(ev[0], st));
    if (two_kernel) {
        if (n_sites > e->short_sites) {
            // scratch between the kernels, grown to the largest short-row submit seen: 48 B + 2 KiB + 12 B per site
            if (e->d_summ) BV_HIP(e, hipFree(e->d_summ));
            if (e->d_bins) BV_HIP(e, hipFree(e->d_bins));
            if (e->d_cand_list) BV_HIP(e, hipFree(e->d_cand_list));
            if (e->d_easy_list) BV_HIP(e, hipFree(e->d_easy_list));
            if (e->d_easy3_list) BV_HIP(e, hipFree(e->d_easy3_list));
            if (e->d_ovf) BV_HIP(e, hipFree(e->d_ovf));
            e->d_ovf = nullptr;
            e->d_summ = nullptr; e->d_bins = nullptr; e->d_cand_list = nullptr; e->d_easy_list = nullptr; e->d_easy3_list = nullptr; e->short_sites = 0;
            BV_HIP(e, hipMalloc(&e->d_summ, sizeof(BvSiteSummary) * (size_t)n_sites));
            BV_HIP(e, hipMalloc(&e->d_bins, sizeof(uint32_t) * BV_S_BIN_STRIDE * (size_t)n_sites));
            BV_HIP(e, hipMalloc(&e->d_cand_list, sizeof(uint32_t) * (size_t)n_sites));
            BV_HIP(e, hipMalloc(&e->d_easy_list, sizeof(uint32_t) * (size_t)n_sites));
            BV_HIP(e, hipMalloc(&e->d_easy3_list, sizeof(uint32_t) * (size_t)n_sites));
            BV_HIP(e, hipMalloc(&e->d_ovf, sizeof(uint32_t) * 4 * (size_t)n_sites));
            e->short_sites = n_sites;
        }
        BvP1ShortArgs s1;
        s1.bs = bs; s1.q = q; s1.ref_base = refb; s1.pitch = P; s1.n_sites = n_sites; s1.n_samples = n_samples;
        s1.flags = e->cfg.flags; s1.n_cu = e->n_cu; s1.min_af = e->cfg.min_af; s1.tables = e->d_tables; s1.out = dout;
        s1.var_list = e->d_var_list; s1.counters = e->d_counters + (size_t)cb * BV_CTR_WORDS;
        s1.summ = e->d_summ; s1.bins = e->d_bins;
        s1.cand_list = e->d_cand_list; s1.easy_list = e->d_easy_list; s1.easy3_list = e->d_easy3_list;
        s1.ovf = e->d_ovf;
        s1.ch = chain;
        s1.mapq = nullptr; s1.rpr = nullptr; s1.rpr_tag = rpr_tag;
        // Rows of at least three 4 KiB slots: pass 1 as ONE persistent kernel (bv_pass1_fused.hip: solver waves beside the
        // streaming waves of every workgroup), which streams the variant sites' rank-sum rows (pass 2) too -- with pop-groups
        // where their tallies stream on their own (<= 7 groups): the launch that follows then carries the group kernels only.
        // Shorter rows, and BV_FLAG_SHORT_ROW_FORM(9) (tests: an independent realisation): a streaming kernel, a solve
        // kernel, pass 2 a launch of its own (bv_pass1_short.hip); BV_FLAG_SHORT_ROW_FORM(10): the fused kernel for pass 1 only.
        const uint32_t form = (e->cfg.flags >> 12) & 0xFu;
        if (form != 9u && bv_p1s_fused_takes(s1)) {
            if (ev) e->ring_one_kernel[e->last_slot] = true;
            // (with pop-groups of any number: the launch behind then carries the group tallies only -- the rank sums of a variant row
            // cost this kernel 0.12 ms per 100 k sites, the workgroup-per-row group kernel 0.16-0.2)
            if (form != 10u && mq != nullptr && rp != nullptr && !(e->cfg.flags & BV_FLAG_PASS2_SWEEP)) {
                s1.mapq = mq; s1.rpr = rp;
                pass2_fused = true;
            }
            bv_launch_p1s_fused(s1, st);
            BV_HIP(e, hipGetLastError());
            e->last_form |= BV_FORM_ONE_KERNEL | (pass2_fused ? BV_FORM_PASS2_FUSED : 0u);
        } else {
            bv_launch_p1s_stream(s1, st);
            BV_HIP(e, hipGetLastError());
            if (ev) BV_HIP(e, hipEventRecord(ev[3], st));
            bv_launch_p1s_solve(s1, st);
            BV_HIP(e, hipGetLastError());
        }
        if (ev) BV_HIP(e, hipEventRecord(ev[1], st));
    } else {
        if (ev) e->ring_one_kernel[e->last_slot] = true;
        BvPass1Args a1;
        a1.bs = bs; a1.q = q; a1.ref_base = refb; a1.pitch = P; a1.n_sites = n_sites;
        a1.n_samples = n_samples; a1.flags = e->cfg.flags; a1.min_af = e->cfg.min_af; a1.tables = e->d_tables; a1.out = dout;
        a1.var_list = e->d_var_list; a1.counters = e->d_counters + (size_t)cb * BV_CTR_WORDS; a1.n_cu = e->n_cu;
        a1.ch = chain;
        bv_launch_pass1(a1, st);
        BV_HIP(e, hipGetLastError());
        e->last_form |= BV_FORM_ONE_KERNEL;
        if (ev) BV_HIP(e, hipEventRecord(ev[1], st));  // (one kernel: no separate event for "the streaming kernel")
    }

    for (size_t r = 0; r < n_rounds; ++r) {
        const size_t g_lo = r * Gr, g_n = n_rounds > 1 ? std::min(Gr, G - g_lo) : G;  // this round's groups
        if (n_rounds > 1) {
            const size_t n16 = ((size_t)n_samples + 15) & ~(size_t)15;
            bv_launch_gid_round(gid, e->d_gid_round, (uint32_t)n16, (uint32_t)g_lo, (uint32_t)g_n, st);
            BV_HIP(e, hipGetLastError());
            bv_launch_gid_prepare(e->d_gid_round, e->d_gidp, (uint32_t)n16, (uint32_t)g_n, st);
            BV_HIP(e, hipGetLastError());
            BV_HIP(e, hipMemsetAsync(e->d_gout_round, 0, S * g_n * sizeof(bv_group_result), st));
        }
        BvPass2Args ac = a2;
        if (n_rounds > 1) { ac.group_id = e->d_gid_round; ac.n_groups = (uint32_t)g_n; ac.gout = e->d_gout_round; }
        if (pass2_fused || r > 0) {  // the rank sums are formed already (by pass 1's kernel / by round 0): what is left is the pop-groups
            if (G == 0) continue;
            ac.mapq = nullptr; ac.rpr = nullptr;
        }
        bv_launch_pass2(ac, st);
        BV_HIP(e, hipGetLastError());
        bv_launch_p2g_solve16(ac, st);
        BV_HIP(e, hipGetLastError());
        if (n_rounds > 1)  // the round's records -> columns [g_lo, g_lo + g_n) of every site's groups
            BV_HIP(e, hipMemcpy2DAsync(dgout + g_lo, G * sizeof(bv_group_result), e->d_gout_round, g_n * sizeof(bv_group_result),
                                       g_n * sizeof(bv_group_result), S, hipMemcpyDeviceToDevice, st));
    }
    if (ev) BV_HIP(e, hipEventRecord(ev[2], st));

    if (rotate) e->ctr_mirror_stale = true;  // mirrored by bv_engine_wait
    else BV_HIP(e, hipMemcpyAsync(e->h_counters, e->d_counters, sizeof(uint32_t) * BV_CTR_WORDS * bv_engine::kCtrBlocks, hipMemcpyDeviceToHost, st));
    if (e->host_out) {
        BV_HIP(e, hipMemcpyAsync(e->host_out, e->stage_out, e->host_out_bytes, hipMemcpyDeviceToHost, st));
        if (e->host_gout && e->host_gout_bytes)
            BV_HIP(e, hipMemcpyAsync(e->host_gout, e->stage_gout, e->host_gout_bytes, hipMemcpyDeviceToHost, st));
    }
    e->submitted = true;
    return mark_done(e, st);
}

int bv_engine_submit(bv_engine *e, const bv_slab *slab, bv_site_result *out, bv_group_result *gout, void *stream_) {
    if (!e) return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_submit: null engine");
    if (!slab || !out) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit: null slab/out");
    if (slab->n_sites == 0) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit: n_sites == 0");
    if (slab->n_sites > e->cfg.max_sites)
        return fail(e, BV_ERR_TOO_LARGE, "bv_engine_submit: n_sites exceeds cfg.max_sites");
    if (slab->n_samples == 0 || slab->pitch < slab->n_samples || (slab->pitch & 15ull))
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit: pitch must be >= n_samples and a multiple of 16");
    if (!slab->base_strand || !slab->qual || !slab->ref_base)
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit: base_strand, qual and ref_base planes are required");
    if ((slab->mapq == nullptr) != (slab->rpr == nullptr))
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit: mapq and rpr planes must be given together");
    if (slab->n_groups > BV_MAX_GROUPS)
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit: n_groups exceeds BV_MAX_GROUPS");
    if (slab->n_groups > 0 && (!slab->group_id || !gout))
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit: n_groups > 0 needs group_id and gout");
    auto misaligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; };
    if (misaligned(slab->base_strand) || misaligned(slab->qual) || misaligned(slab->mapq) || misaligned(slab->rpr))
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit: planes must be 16-byte aligned");
    if (slab->mem_kind != BV_MEM_HOST && misaligned(out))
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit: device record buffers must be 16-byte aligned");
    if ((slab->layout & ~BV_SLAB_RPR_TAGGED) || slab->reserved_)
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit: unknown bv_slab.layout bits (built against another BV_ABI_VERSION?)");

    BV_HIP(e, hipSetDevice(e->cfg.device));
    if ((e->cfg.flags & BV_FLAG_LANES) && !e->is_lane && slab->mem_kind != BV_MEM_HOST) {
        // two lanes: this submit goes to the child engine whose turn it is, on that child's own stream, ordered behind what
        // the caller's stream holds now; the caller's stream gets nothing back (bv_engine_join / bv_engine_wait)
        const int k = (int)(e->lane_next++ % (unsigned)e->n_lanes);
        if (!e->lane[k]) {
            bv_engine_config c = e->cfg;
            c.flags &= ~BV_FLAG_LANES;
            int rc = bv_engine_create(&c, &e->lane[k]);
            if (rc != BV_OK) return fail(e, rc, std::string("bv_engine_submit: lane engine: ") + bv_last_error(nullptr));
            e->lane[k]->is_lane = true;
        }
        bv_engine *l = e->lane[k];
        // (only if that stream holds unfinished work: recording an event on an idle stream and waiting for it on another cost
        // 0.5-2 ms per submit on this stack -- measured, round 3 -- against ~10 us when the marker follows real work)
        // NULL means the engine's own stream here too (include/basevar_amd.h): planes written on bv_engine_stream(e) just
        // before a NULL-stream submit are ordered like those of any other stream.  (A stream that is being captured
        // answers the query with an error: treated as "holds work", the marker is then part of the capture.)
        hipStream_t src = stream_ ? (hipStream_t)stream_ : e->stream;
        if (hipStreamQuery(src) != hipSuccess) {
            (void)hipGetLastError();  // hipErrorNotReady is the answer, not an error
            if (!e->ev_entry) BV_HIP(e, hipEventCreateWithFlags(&e->ev_entry, hipEventDisableTiming));
            BV_HIP(e, hipEventRecord(e->ev_entry, src));
            BV_HIP(e, hipStreamWaitEvent(l->stream, e->ev_entry, 0));
        }
        const int rc = bv_engine_submit(l, slab, out, gout, nullptr);
        if (rc != BV_OK) return fail(e, rc, bv_last_error(l));
        e->last_lane = k;
        e->submitted = true;
        return BV_OK;
    }
    e->last_lane = -1;
    hipStream_t st = stream_ ? (hipStream_t)stream_ : e->stream;
    {
        int rc = use_stream(e, st);
        if (rc != BV_OK) return rc;
    }

    const uint8_t *bs = slab->base_strand, *q = slab->qual, *mq = slab->mapq, *refb = slab->ref_base,
                  *gid = slab->group_id;
    const uint16_t *rp = slab->rpr;
    bv_site_result *dout = out;
    bv_group_result *dgout = gout;
    const size_t S = slab->n_sites, P = slab->pitch, G = slab->n_groups;
    bv_engine::StageSlot *slot = nullptr;
    if (slab->mem_kind == BV_MEM_HOST) {
        // host planes -> a staging slot, copied by the copy stream (under the kernels of the previous submit)
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        HostPlane pl[5] = {{bs, S * P, nullptr}, {q, S * P, nullptr}, {mq, mq ? S * P : 0, nullptr},
                           {rp, rp ? S * P * 2 : 0, nullptr}, {refb, S, nullptr}};
        const size_t out_b = up(S * sizeof(bv_site_result)), gout_b = up(S * G * sizeof(bv_group_result));
        uint8_t *extra = nullptr;
        int rc = stage_host_planes(e, pl, 5, out_b + gout_b, &slot, &extra, st);
        if (rc != BV_OK) return rc;
        rc = stage_publish(e, slot, st);
        if (rc != BV_OK) return rc;
        bs = pl[0].dev; q = pl[1].dev; mq = pl[2].dev;
        rp = reinterpret_cast<const uint16_t *>(pl[3].dev);
        refb = pl[4].dev;
        dout = reinterpret_cast<bv_site_result *>(extra);
        dgout = G ? reinterpret_cast<bv_group_result *>(extra + out_b) : nullptr;
        e->stage_out = dout; e->stage_gout = dgout;
        e->host_out = out; e->host_gout = gout;
        e->host_out_bytes = S * sizeof(bv_site_result);
        e->host_gout_bytes = S * G * sizeof(bv_group_result);
    } else {
        e->host_out = nullptr; e->host_gout = nullptr;
    }
    if (G) {
        int rc = stage_group_ids(e, slab->group_id, slab->n_samples, slab->mem_kind == BV_MEM_HOST, st, &gid);
        if (rc != BV_OK) return rc;
    }

    int rc = launch_passes(e, bs, q, mq, rp, refb, gid, P, slab->n_sites, slab->n_samples, slab->n_groups, dout, dgout, st, nullptr, false, slab->layout);
    if (rc == BV_OK && slot) rc = stage_release(e, slot, st);  // planes read, records copied back: the slot may be refilled
    return rc;
}

// Several device-resident slabs of one row length, ONE launch per pass (BvChain): what a host with a few small batches
// ready should call -- the tail of every batch but the last hides under the next batch's stream.  Falls back to one
// submit per slab whenever the chained kernels do not apply (short rows, pop-groups, host memory, a forced kernel shape).
int bv_engine_submit_many(bv_engine *e, uint32_t n_slabs, const bv_slab *slabs, bv_site_result *const *outs, void *stream_) {
    return bv_engine_submit_many_g(e, n_slabs, slabs, outs, nullptr, stream_);
}

// The same with pop-groups: gouts[k] = slab k's [n_sites][n_groups] records (NULL array: no slab may have groups).  A queue
// with groups chains when every slab names the SAME group_id array and group count (one cohort).
int bv_engine_submit_many_g(bv_engine *e, uint32_t n_slabs, const bv_slab *slabs, bv_site_result *const *outs,
                            bv_group_result *const *gouts, void *stream_) {
    if (!e) return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_submit_many: null engine");
    if (!slabs || !outs || n_slabs == 0) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit_many: null / empty argument");
    bool chainable = n_slabs > 1;
    uint64_t total = 0;
    auto misaligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; };
    // every slab is checked before anything is launched (the checks of bv_engine_submit)
    for (uint32_t k = 0; k < n_slabs; ++k) {
        const bv_slab &s = slabs[k];
        if (!outs[k]) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit_many: null record buffer");
        if (s.n_sites == 0 || s.n_samples == 0 || s.pitch < s.n_samples || (s.pitch & 15ull) || !s.base_strand || !s.qual || !s.ref_base ||
            (s.mapq == nullptr) != (s.rpr == nullptr) || misaligned(s.base_strand) || misaligned(s.qual) || misaligned(s.mapq) ||
            misaligned(s.rpr) || (s.mem_kind != BV_MEM_HOST && misaligned(outs[k])) || s.n_groups > BV_MAX_GROUPS || (s.layout & ~BV_SLAB_RPR_TAGGED) || s.reserved_)
            return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit_many: a slab fails the checks of bv_engine_submit");
        if (s.n_groups > 0 && (!s.group_id || !gouts || !gouts[k]))
            return fail(e, BV_ERR_INVALID_ARG, "bv_engine_submit_many: a slab with pop-groups needs group_id and its gouts[k] (bv_engine_submit_many_g)");
        if (s.n_sites > e->cfg.max_sites) return fail(e, BV_ERR_TOO_LARGE, "bv_engine_submit_many: a slab exceeds cfg.max_sites");
        // (host memory and the diagnostic kernel choices take kernels that know no chain; a queue is one cohort: one row
        // length, one pitch, one set of planes, one group assignment)
        chainable = chainable && s.mem_kind != BV_MEM_HOST && !(e->cfg.flags & (BV_FLAG_PASS2_SWEEP | BV_FLAG_GROUP_INLINE)) &&
                    s.n_samples == slabs[0].n_samples && s.pitch == slabs[0].pitch && (s.mapq == nullptr) == (slabs[0].mapq == nullptr) &&
                    s.n_groups == slabs[0].n_groups && (s.n_groups == 0 || s.group_id == slabs[0].group_id) && s.layout == slabs[0].layout;
        total += s.n_sites;
    }
    const uint32_t G = slabs[0].n_groups;
    // pop-groups chain only when every (site, group) of a launch has an item in the scratch (no inline solves: their kernels
    // would need the segment look-up too) -- checked per launch below through the 8 GiB cap of launch_passes
    if (chainable && G && (uint64_t)std::min<uint64_t>(total, e->cfg.max_sites) * G * sizeof(uint32_t) * BV_P2G_ITEM_WORDS > (8192ull << 20)) chainable = false;
    if (G > BV_GROUPS_PER_ROUND) chainable = false;  // several rounds of groups (launch_passes): slab by slab
    if (!chainable) {
        for (uint32_t k = 0; k < n_slabs; ++k) {
            int rc = bv_engine_submit(e, &slabs[k], outs[k], slabs[k].n_groups ? gouts[k] : nullptr, stream_);
            if (rc != BV_OK) return rc;
        }
        return BV_OK;
    }
    if (total > e->cfg.max_sites) return fail(e, BV_ERR_TOO_LARGE, "bv_engine_submit_many: the slabs together exceed cfg.max_sites");
    BV_HIP(e, hipSetDevice(e->cfg.device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : e->stream;
    {
        int rc = use_stream(e, st);
        if (rc != BV_OK) return rc;
    }
    e->host_out = nullptr; e->host_gout = nullptr;
    e->last_lane = -1;
    const size_t P = slabs[0].pitch;
    const bool ranks = slabs[0].mapq != nullptr;
    const uint8_t *gid = nullptr;
    if (G) {
        int rc = stage_group_ids(e, slabs[0].group_id, slabs[0].n_samples, false, st, &gid);
        if (rc != BV_OK) return rc;
    }
    // at most BV_MAX_CHAIN slabs per launch
    for (uint32_t k0 = 0; k0 < n_slabs; k0 += BV_MAX_CHAIN) {
        const uint32_t nk = n_slabs - k0 < (uint32_t)BV_MAX_CHAIN ? n_slabs - k0 : (uint32_t)BV_MAX_CHAIN;
        BvChain ch{};
        ch.n = nk;
        uint32_t first = 0;
        for (uint32_t i = 0; i < nk; ++i) {
            const bv_slab &s = slabs[k0 + i];
            const size_t bias = (size_t)first * P;
            ch.first[i] = first;
            ch.bs[i] = s.base_strand - bias; ch.q[i] = s.qual - bias;
            ch.mapq[i] = ranks ? s.mapq - bias : nullptr; ch.rpr[i] = ranks ? s.rpr - bias : nullptr;
            ch.ref_base[i] = s.ref_base - first;
            ch.out[i] = outs[k0 + i] - first;
            ch.gout[i] = G ? gouts[k0 + i] - (size_t)first * G : nullptr;
            if (G) BV_HIP(e, hipMemsetAsync(gouts[k0 + i], 0, (size_t)s.n_sites * G * sizeof(bv_group_result), st));
            first += s.n_sites;
        }
        const bv_slab &s0 = slabs[k0];
        bv_group_result *g0 = G ? gouts[k0] : nullptr;
        if (nk == 1) {
            int rc = launch_passes(e, s0.base_strand, s0.qual, s0.mapq, s0.rpr, s0.ref_base, gid, P, first, s0.n_samples, G, outs[k0], g0, st, nullptr, false, s0.layout);
            if (rc != BV_OK) return rc;
            continue;
        }
        // the segment table lives in device memory (a ring of 16: a table is rewritten only 16 chained launches later)
        if (!e->d_chain) BV_HIP(e, hipMalloc(&e->d_chain, sizeof(BvChain) * 16));
        BvChain *d_ch = e->d_chain + (e->chain_next++ & 15u);
        BV_HIP(e, hipMemcpyAsync(d_ch, &ch, sizeof(BvChain), hipMemcpyHostToDevice, st));
        if (s0.n_samples > BV_SHORT_ROW_MAX) {
            // long rows: every kernel looks its segment up per site (planes, reference bases, records)
            int rc = launch_passes(e, s0.base_strand, s0.qual, s0.mapq, s0.rpr, s0.ref_base, gid, P, first, s0.n_samples, G, outs[k0], g0, st, d_ch, false, s0.layout);
            if (rc != BV_OK) return rc;
        } else {
            // short rows: the planes are looked up per row (wave-uniform places only); the per-site reference bases and records,
            // which the lane-per-site and four-per-wave kernels touch with one site per lane, go through contiguous copies
            // (the pop-group records are written once per (variant site, group): looked up where they are written)
            if (!e->d_ref_cat || !e->d_out_cat) {
                if (e->d_ref_cat) (void)hipFree(e->d_ref_cat);
                if (e->d_out_cat) (void)hipFree(e->d_out_cat);
                e->d_ref_cat = nullptr; e->d_out_cat = nullptr;
                BV_HIP(e, hipMalloc(&e->d_ref_cat, (size_t)e->cfg.max_sites + 256));
                BV_HIP(e, hipMalloc(&e->d_out_cat, sizeof(bv_site_result) * (size_t)e->cfg.max_sites));
            }
            bv_launch_chain_gather_ref(d_ch, first, e->d_ref_cat, st);
            BV_HIP(e, hipGetLastError());
            int rc = launch_passes(e, s0.base_strand, s0.qual, s0.mapq, s0.rpr, e->d_ref_cat, gid, P, first, s0.n_samples, G, e->d_out_cat, g0, st, d_ch, true, s0.layout);
            if (rc != BV_OK) return rc;
            bv_launch_chain_scatter_out(d_ch, first, e->d_out_cat, st);
            BV_HIP(e, hipGetLastError());
            rc = mark_done(e, st);  // the scatter is the end of this submit
            if (rc != BV_OK) return rc;
        }
    }
    return BV_OK;
}

// ---------------------------------------------------------------- sample-axis tile mode
int bv_engine_tiles_begin(bv_engine *e, uint32_t n_sites, uint32_t n_samples_total, uint32_t n_groups, int with_ranks) {
    if (!e) return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_tiles_begin: null engine");
    if (n_sites == 0 || n_samples_total == 0) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_begin: empty job");
    if (n_sites > e->cfg.max_sites) return fail(e, BV_ERR_TOO_LARGE, "bv_engine_tiles_begin: n_sites exceeds cfg.max_sites");
    if (n_groups > BV_MAX_GROUPS) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_begin: n_groups exceeds BV_MAX_GROUPS");
    BV_HIP(e, hipSetDevice(e->cfg.device));
    // The job's state is cleared ON THE ENGINE'S STREAM, ordered like a submit (use_stream / mark_done): a tile added on another
    // stream waits for it.  (Until round 6 these were hipMemset calls on the null stream, which nothing orders against the engine's
    // non-blocking stream: a tally kernel could meet the state of a fresh allocation -- once in the round-6 campaigns, 8,037
    // mismatching fields in one 4,096-site job, gone on the re-run.)
    hipStream_t st0 = e->stream;
    {
        int rc = use_stream(e, st0);
        if (rc != BV_OK) return rc;
    }
    e->tile_join = false;
    e->tile_layout = 0; e->tile_layout_set = false; e->j_filled = false;
    if (!(e->cfg.flags & BV_FLAG_TILE_STATE)) {
        // joined rows: [n_sites][pitch] planes resident in HBM, if they fit next to what is already there
        const size_t pitch = ((size_t)n_samples_total + 255) & ~(size_t)255, plane = (size_t)n_sites * pitch;
        const size_t o_q = plane, o_mq = 2 * plane, o_rp = o_mq + (with_ranks ? plane : 0), o_gid = o_rp + (with_ranks ? 2 * plane : 0),
                     need = o_gid + pitch;
        bool ok = need <= e->j_bytes;
        if (!ok) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= (free_b + e->j_bytes) / 10 * 9) {
                if (e->j_buf) BV_HIP(e, hipFree(e->j_buf));
                e->j_buf = nullptr;
                e->j_bytes = 0;
                if (hipMalloc(&e->j_buf, need) == hipSuccess) {
                    e->j_bytes = need;
                    ok = true;
                } else {
                    (void)hipGetLastError();
                    e->j_buf = nullptr;
                }
            }
        }
        if (ok) {
            e->j_pitch = pitch; e->j_o_q = o_q; e->j_o_mq = o_mq; e->j_o_rp = o_rp; e->j_o_gid = o_gid;
            // (tiles fill the columns from the left in the order they are added; what a job leaves unfilled is set to 'N' at
            // finish -- not the whole plane here: 8.6 GB of memset for 8 Ki sites x 1 M samples)
            BV_HIP(e, hipMemsetAsync(e->j_buf + o_gid, 0xFF, pitch, st0));   // no pop-group
            e->tile_sites = n_sites; e->tile_groups = n_groups; e->tile_stride = 0;
            e->tile_samples_total = n_samples_total; e->tile_samples_seen = 0;
            e->tile_ranks = with_ranks != 0;
            e->tile_join = true;
            e->tile_open = true;
            return mark_done(e, st0);
        }
    }
    if (n_groups > BV_GROUPS_PER_ROUND)
        return fail(e, BV_ERR_TOO_LARGE, "bv_engine_tiles_begin: more than 32 pop-groups need the joined-rows realisation (2 KiB of "
                                         "per-site state per group otherwise), and the joined planes of this job do not fit the device");
    // H1 2048 + Hm 1024 + Hr 4 x W + Hg 512/group (bv_tiles.hip); W = 1024 ranks, or what the caller announced
    const uint32_t rank_win = with_ranks > 1 ? (uint32_t)((with_ranks + 1023) / 1024 * 1024) : 1024u;
    const uint32_t hg_off = 3072u + 4u * rank_win, ord_off = hg_off + n_groups * 512u, stride = ord_off + (1u + n_groups) * BV_TS_ORD_WORDS;
    e->tile_rank_win = rank_win; e->tile_hg_off = hg_off; e->tile_ord_off = ord_off;
    if (!e->tile_ovf) BV_HIP(e, hipMalloc(&e->tile_ovf, sizeof(uint32_t) * (2u + 2u * (size_t)bv_engine::kOvfCap)));
    BV_HIP(e, hipMemsetAsync(e->tile_ovf, 0, 2 * sizeof(uint32_t), st0));
    const size_t bytes = (size_t)n_sites * stride * sizeof(uint32_t), mbytes = (size_t)n_sites * sizeof(uint32_t);
    if (bytes > e->tile_state_bytes) {
        if (e->tile_state) BV_HIP(e, hipFree(e->tile_state));
        e->tile_state = nullptr;
        e->tile_state_bytes = 0;
        BV_HIP(e, hipMalloc(&e->tile_state, bytes));
        e->tile_state_bytes = bytes;
    }
    if (mbytes > e->tile_maxr_bytes) {
        if (e->tile_maxr) BV_HIP(e, hipFree(e->tile_maxr));
        e->tile_maxr = nullptr;
        e->tile_maxr_bytes = 0;
        BV_HIP(e, hipMalloc(&e->tile_maxr, mbytes));
        e->tile_maxr_bytes = mbytes;
    }
    BV_HIP(e, hipMemsetAsync(e->tile_state, 0, bytes, st0));
    BV_HIP(e, hipMemsetAsync(e->tile_maxr, 0, mbytes, st0));
    e->tile_sites = n_sites; e->tile_groups = n_groups; e->tile_stride = stride;
    e->tile_samples_total = n_samples_total; e->tile_samples_seen = 0;
    e->tile_ranks = with_ranks != 0;
    e->tile_open = true;
    return mark_done(e, st0);
}

int bv_tile_packed_layout(uint32_t n_sites, uint32_t width, int with_ranks, int with_groups, uint64_t *pitch,
                          uint64_t offsets[5], uint64_t *total_bytes) {
    if (!n_sites || !width || !pitch || !offsets || !total_bytes)
        return fail(nullptr, BV_ERR_INVALID_ARG, "bv_tile_packed_layout: bad argument");
    auto up = [](uint64_t x) { return (x + 255) & ~(uint64_t)255; };
    const uint64_t P = ((uint64_t)width + 15) / 16 * 16, plane = up((uint64_t)n_sites * P);
    uint64_t at = 0;
    offsets[0] = at; at += plane;
    offsets[1] = at; at += plane;
    offsets[2] = with_ranks ? at : 0; at += with_ranks ? plane : 0;
    offsets[3] = with_ranks ? at : 0; at += with_ranks ? up((uint64_t)n_sites * P * 2) : 0;
    offsets[4] = with_groups ? at : 0; at += with_groups ? up(P) : 0;
    *pitch = P;
    *total_bytes = at;
    return BV_OK;
}

int bv_engine_tiles_add(bv_engine *e, const bv_slab *t, void *stream_) {
    if (!e || !t) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add: null argument");
    if (!e->tile_open) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add: call bv_engine_tiles_begin first");
    if (t->n_sites != e->tile_sites) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add: tile n_sites differs from the job's");
    if (t->n_samples == 0 || t->pitch < t->n_samples || (t->pitch & 15ull) || !t->base_strand || !t->qual)
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add: bad tile geometry or missing planes");
    if (e->tile_ranks && (!t->mapq || !t->rpr)) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add: job was opened with rank planes");
    if (e->tile_groups && !t->group_id) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add: job has groups, tile has no group_id");
    if ((uint64_t)e->tile_samples_seen + t->n_samples > e->tile_samples_total)
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add: more samples than announced");
    if ((t->layout & ~BV_SLAB_RPR_TAGGED) || t->reserved_ || (e->tile_layout_set && t->layout != e->tile_layout))
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add: every tile of a job must have the same bv_slab.layout (known bits only)");
    e->tile_layout = t->layout; e->tile_layout_set = true;
    BV_HIP(e, hipSetDevice(e->cfg.device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : e->stream;
    {
        int rc = use_stream(e, st);
        if (rc != BV_OK) return rc;
    }
    const uint8_t *bs = t->base_strand, *q = t->qual, *mq = e->tile_ranks ? t->mapq : nullptr, *gid = e->tile_groups ? t->group_id : nullptr;
    const uint16_t *rp = e->tile_ranks ? t->rpr : nullptr;
    const size_t S = t->n_sites, P = t->pitch;
    bv_engine::StageSlot *slot = nullptr;
    if (t->mem_kind == BV_MEM_HOST) {
        HostPlane pl[5] = {{bs, S * P, nullptr}, {q, S * P, nullptr}, {mq, mq ? S * P : 0, nullptr},
                           {rp, rp ? S * P * 2 : 0, nullptr}, {gid, gid ? (size_t)t->n_samples : 0, nullptr}};
        int rc = stage_host_planes(e, pl, 5, 0, &slot, nullptr, st);
        if (rc != BV_OK) return rc;
        rc = stage_publish(e, slot, st);
        if (rc != BV_OK) return rc;
        bs = pl[0].dev; q = pl[1].dev; mq = pl[2].dev;
        rp = reinterpret_cast<const uint16_t *>(pl[3].dev);
        gid = pl[4].dev;
    }
    if (e->tile_join) {
        const uint64_t lo = e->tile_samples_seen, JP = e->j_pitch;
        const uint32_t w = t->n_samples, rows = t->n_sites;
        BvTileScatterArgs sc;
        sc.max_rows = rows;
        sc.n_planes = 0;
        auto plane = [&](uint8_t *dst, const uint8_t *src, uint64_t scale, uint32_t n_rows) {
            BvTileScatterPlane &p = sc.plane[sc.n_planes++];
            p.dst = dst; p.src = src; p.dst_pitch = scale * JP; p.src_pitch = scale * P; p.col_off = scale * lo;
            p.width_bytes = (uint32_t)(scale * w); p.n_rows = n_rows;
        };
        plane(e->j_buf, bs, 1, rows);
        plane(e->j_buf + e->j_o_q, q, 1, rows);
        if (mq) plane(e->j_buf + e->j_o_mq, mq, 1, rows);
        if (rp) plane(e->j_buf + e->j_o_rp, reinterpret_cast<const uint8_t *>(rp), 2, rows);
        if (gid) plane(e->j_buf + e->j_o_gid, gid, 1, 1);  // one row: the tile's group ids, in the same launch
        bv_launch_tile_scatter(sc, st);
        BV_HIP(e, hipGetLastError());
        if (slot) {
            int rc = stage_release(e, slot, st);
            if (rc != BV_OK) return rc;
        }
        e->tile_samples_seen += t->n_samples;
        return mark_done(e, st);
    }
    BvTileArgs a;
    a.bs = bs; a.q = q; a.mapq = mq; a.rpr = rp; a.group_id = gid; a.pitch = P; a.n_sites = t->n_sites;
    a.width = t->n_samples; a.n_groups = e->tile_groups; a.stride = e->tile_stride; a.state = e->tile_state;
    a.rank_win = e->tile_rank_win; a.hg_off = e->tile_hg_off;
    a.maxr = e->tile_maxr;
    a.ord_off = e->tile_ord_off; a.col0 = e->tile_samples_seen; a.ovf = e->tile_ovf; a.ovf_cap = bv_engine::kOvfCap;
    a.rpr_tag = (e->tile_layout & BV_SLAB_RPR_TAGGED) ? 1u : 0u;
    bv_launch_tile_tally(a, st);
    BV_HIP(e, hipGetLastError());
    if (slot) {
        int rc = stage_release(e, slot, st);
        if (rc != BV_OK) return rc;
    }
    e->tile_samples_seen += t->n_samples;
    return mark_done(e, st);
}

// A tile as its covered cells only (include/basevar_amd.h): scattered into the joined planes, which hold "uncovered" wherever no
// tile has delivered yet (filled once, when the job's first packed tile arrives), or added to the per-site tallies entry by entry.
int bv_sparse_tile_packed_layout(uint32_t n_sites, uint32_t n_entries, uint32_t width, int with_ranks, int with_groups,
                                 uint64_t offsets[7], uint64_t *total_bytes) {
    if (!n_sites || !width || !offsets || !total_bytes) return fail(nullptr, BV_ERR_INVALID_ARG, "bv_sparse_tile_packed_layout: bad argument");
    auto up = [](uint64_t x) { return (x + 255) & ~(uint64_t)255; };
    const uint64_t E = n_entries ? n_entries : 1u;
    uint64_t at = 0;
    offsets[0] = at; at += up(4ull * ((uint64_t)n_sites + 1));
    offsets[1] = at; at += up(2 * E);
    offsets[2] = at; at += up(E);
    offsets[3] = at; at += up(E);
    offsets[4] = with_ranks ? at : 0; at += with_ranks ? up(E) : 0;
    offsets[5] = with_ranks ? at : 0; at += with_ranks ? up(2 * E) : 0;
    offsets[6] = with_groups ? at : 0; at += with_groups ? up(width) : 0;
    *total_bytes = at;
    return BV_OK;
}

int bv_engine_tiles_add_sparse(bv_engine *e, const bv_sparse_tile *t, void *stream_) {
    if (!e || !t) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_sparse: null argument");
    if (!e->tile_open) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_sparse: call bv_engine_tiles_begin first");
    if (t->n_sites != e->tile_sites) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_sparse: tile n_sites differs from the job's");
    if (t->n_samples == 0 || t->n_samples > 65536u || !t->row_start || (t->n_entries && (!t->sample || !t->base_strand || !t->qual)))
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_sparse: bad tile geometry (at most 65,536 samples per tile) or missing arrays");
    if (e->tile_ranks && t->n_entries && (!t->mapq || !t->rpr)) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_sparse: job was opened with rank planes");
    if (e->tile_groups && !t->group_id) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_sparse: job has groups, tile has no group_id");
    if ((uint64_t)e->tile_samples_seen + t->n_samples > e->tile_samples_total)
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_sparse: more samples than announced");
    if ((t->layout & ~BV_SLAB_RPR_TAGGED) || (e->tile_layout_set && t->layout != e->tile_layout))
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_sparse: every tile of a job must have the same layout (known bits only)");
    e->tile_layout = t->layout; e->tile_layout_set = true;
    BV_HIP(e, hipSetDevice(e->cfg.device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : e->stream;
    {
        int rc = use_stream(e, st);
        if (rc != BV_OK) return rc;
    }
    const size_t S = t->n_sites, E = t->n_entries;
    const bool ranks = e->tile_ranks;
    const uint32_t *row_start = t->row_start;
    const uint16_t *smp = t->sample, *rp = ranks ? t->rpr : nullptr;
    const uint8_t *bs = t->base_strand, *q = t->qual, *mq = ranks ? t->mapq : nullptr, *gid = e->tile_groups ? t->group_id : nullptr;
    bv_engine::StageSlot *slot = nullptr;
    if (t->mem_kind == BV_MEM_HOST) {
        HostPlane pl[7] = {{row_start, 4 * (S + 1), nullptr}, {smp, 2 * E, nullptr}, {bs, E, nullptr}, {q, E, nullptr}, {mq, mq ? E : 0, nullptr},
                           {rp, rp ? 2 * E : 0, nullptr}, {gid, gid ? (size_t)t->n_samples : 0, nullptr}};
        int rc = stage_host_planes(e, pl, 7, 0, &slot, nullptr, st);
        if (rc != BV_OK) return rc;
        rc = stage_publish(e, slot, st);
        if (rc != BV_OK) return rc;
        row_start = reinterpret_cast<const uint32_t *>(pl[0].dev); smp = reinterpret_cast<const uint16_t *>(pl[1].dev);
        bs = pl[2].dev; q = pl[3].dev; mq = pl[4].dev; rp = reinterpret_cast<const uint16_t *>(pl[5].dev); gid = pl[6].dev;
    }
    BvSparseTileArgs a{};
    a.row_start = row_start; a.sample = smp; a.call = bs; a.phred = q; a.mapq = E ? mq : nullptr; a.rank = rp;
    a.n_sites = t->n_sites; a.width = t->n_samples; a.n_entries = t->n_entries;
    a.rpr_tag = (e->tile_layout & BV_SLAB_RPR_TAGGED) ? 1u : 0u;
    a.col0 = e->tile_samples_seen;
    if (e->tile_join) {
        uint8_t *jb = e->j_buf, *jq = e->j_buf + e->j_o_q, *jm = ranks ? e->j_buf + e->j_o_mq : nullptr;
        uint16_t *jr = ranks ? reinterpret_cast<uint16_t *>(e->j_buf + e->j_o_rp) : nullptr;
        if (!e->j_filled) {
            // every column that has not been delivered yet: "nobody covered" (once per job; dense tiles that follow overwrite theirs)
            if (e->tile_samples_seen == 0) {
                bv_launch_tile_fill_uncovered(jb, jq, jm, jr, (uint64_t)e->tile_sites * e->j_pitch, a.rpr_tag, st);
                BV_HIP(e, hipGetLastError());
            } else {
                const size_t lo = e->tile_samples_seen, w = e->tile_samples_total - lo;
                BV_HIP(e, hipMemset2DAsync(jb + lo, e->j_pitch, 0x08, w, S, st));
                BV_HIP(e, hipMemset2DAsync(jq + lo, e->j_pitch, 0, w, S, st));
                if (ranks) {
                    BV_HIP(e, hipMemset2DAsync(jm + lo, e->j_pitch, 0, w, S, st));
                    BV_HIP(e, hipMemset2DAsync(reinterpret_cast<uint8_t *>(jr) + 2 * lo, 2 * e->j_pitch, a.rpr_tag ? 0x80 : 0, 2 * w, S, st));
                }
            }
            e->j_filled = true;
        }
        a.bs = jb; a.q = jq; a.mq = jm; a.rp = jr; a.pitch = e->j_pitch;
        if (E) {
            bv_launch_tile_sparse_scatter(a, st);
            BV_HIP(e, hipGetLastError());
        }
        if (gid) BV_HIP(e, hipMemcpyAsync(e->j_buf + e->j_o_gid + e->tile_samples_seen, gid, t->n_samples, hipMemcpyDeviceToDevice, st));
    } else {
        a.group_id = gid; a.n_groups = e->tile_groups; a.stride = e->tile_stride; a.rank_win = e->tile_rank_win; a.hg_off = e->tile_hg_off;
        a.ord_off = e->tile_ord_off; a.ovf_cap = bv_engine::kOvfCap; a.state = e->tile_state; a.maxr = e->tile_maxr; a.ovf = e->tile_ovf;
        if (E) {
            bv_launch_tile_sparse_tally(a, st);
            BV_HIP(e, hipGetLastError());
        }
    }
    if (slot) {
        int rc = stage_release(e, slot, st);
        if (rc != BV_OK) return rc;
    }
    e->tile_samples_seen += t->n_samples;
    return mark_done(e, st);
}

// Several tiles at once.  Device-resident tiles of a joined-rows job go to their columns in ONE launch per <= 256 tiles
// (descriptor table in device memory); anything else -- host tiles, the per-site-tally realisation -- is added tile by tile.
int bv_engine_tiles_add_many(bv_engine *e, uint32_t n_tiles, const bv_slab *tiles, void *stream_) {
    if (!e || !tiles || n_tiles == 0) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_many: null / empty argument");
    if (!e->tile_open) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_many: call bv_engine_tiles_begin first");
    bool one_launch = e->tile_join;
    uint64_t seen = e->tile_samples_seen;
    for (uint32_t k = 0; k < n_tiles; ++k) {  // the checks of bv_engine_tiles_add, for every tile before anything is queued
        const bv_slab &t = tiles[k];
        if (t.n_sites != e->tile_sites) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_many: tile n_sites differs from the job's");
        if (t.n_samples == 0 || t.pitch < t.n_samples || (t.pitch & 15ull) || !t.base_strand || !t.qual)
            return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_many: bad tile geometry or missing planes");
        if (e->tile_ranks && (!t.mapq || !t.rpr)) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_many: job was opened with rank planes");
        if (e->tile_groups && !t.group_id) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_many: job has groups, tile has no group_id");
        seen += t.n_samples;
        if (seen > e->tile_samples_total) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_many: more samples than announced");
        if ((t.layout & ~BV_SLAB_RPR_TAGGED) || t.reserved_ || t.layout != tiles[0].layout || (e->tile_layout_set && t.layout != e->tile_layout))
            return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_add_many: every tile of a job must have the same bv_slab.layout (known bits only)");
        one_launch = one_launch && t.mem_kind != BV_MEM_HOST;
    }
    if (!one_launch) {
        for (uint32_t k = 0; k < n_tiles; ++k) {
            int rc = bv_engine_tiles_add(e, &tiles[k], stream_);
            if (rc != BV_OK) return rc;
        }
        return BV_OK;
    }
    BV_HIP(e, hipSetDevice(e->cfg.device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : e->stream;
    {
        int rc = use_stream(e, st);
        if (rc != BV_OK) return rc;
    }
    e->tile_layout = tiles[0].layout; e->tile_layout_set = true;
    const uint64_t JP = e->j_pitch;
    for (uint32_t k0 = 0; k0 < n_tiles; k0 += BV_TILE_MANY_MAX) {
        const uint32_t nk = n_tiles - k0 < (uint32_t)BV_TILE_MANY_MAX ? n_tiles - k0 : (uint32_t)BV_TILE_MANY_MAX;
        const int slot = (int)(e->desc_next++ % bv_engine::kDescRing);
        constexpr size_t kBytes = sizeof(BvTileScatterPlane) * 5 * BV_TILE_MANY_MAX;
        if (!e->h_desc[slot]) {
            BV_HIP(e, hipHostMalloc(&e->h_desc[slot], kBytes));
            BV_HIP(e, hipMalloc(&e->d_desc[slot], kBytes));
            BV_HIP(e, hipEventCreateWithFlags(&e->ev_desc[slot], hipEventDisableTiming));
        }
        if (e->desc_used[slot]) BV_HIP(e, hipEventSynchronize(e->ev_desc[slot]));  // the copy that last read this pinned table
        // the usual job -- tiles of one width and pitch, every byte quantity a multiple of 8: whole destination rows are
        // gathered across the tiles (bv_tile_join_rows_kernel); anything else: one descriptor per tile and plane
        const bv_slab &t0 = tiles[k0];
        bool uniform = !((t0.n_samples | t0.pitch | e->tile_samples_seen | JP) & 7u);
        for (uint32_t i = 0; i < nk && uniform; ++i) {
            const bv_slab &t = tiles[k0 + i];
            uniform = t.n_samples == t0.n_samples && t.pitch == t0.pitch &&
                      !(((uintptr_t)t.base_strand | (uintptr_t)t.qual | (uintptr_t)t.mapq | (uintptr_t)t.rpr | (uintptr_t)t.group_id) & 7u);
        }
        if (uniform) {
            const uint8_t **tab = reinterpret_cast<const uint8_t **>(e->h_desc[slot]);  // [5][nk] pointers
            for (uint32_t i = 0; i < nk; ++i) {
                const bv_slab &t = tiles[k0 + i];
                tab[0 * nk + i] = t.base_strand; tab[1 * nk + i] = t.qual; tab[2 * nk + i] = t.mapq;
                tab[3 * nk + i] = reinterpret_cast<const uint8_t *>(t.rpr); tab[4 * nk + i] = t.group_id;
            }
            BV_HIP(e, hipMemcpyAsync(e->d_desc[slot], tab, sizeof(void *) * 5 * nk, hipMemcpyHostToDevice, st));
            BV_HIP(e, hipEventRecord(e->ev_desc[slot], st));
            e->desc_used[slot] = true;
            const uint8_t *const *dtab = reinterpret_cast<const uint8_t *const *>(e->d_desc[slot]);
            const uint64_t lo = e->tile_samples_seen;
            auto join = [&](uint8_t *dst, int k, uint64_t scale, uint32_t n_rows) {
                BvTileJoinArgs ja;
                ja.dst = dst; ja.srcs = dtab + (size_t)k * nk; ja.dst_pitch = scale * JP; ja.src_pitch = scale * t0.pitch; ja.col_off = scale * lo;
                ja.width_bytes = (uint32_t)(scale * t0.n_samples); ja.n_tiles = nk; ja.n_rows = n_rows;
                bv_launch_tile_join_rows(ja, st);
            };
            join(e->j_buf, 0, 1, t0.n_sites);
            join(e->j_buf + e->j_o_q, 1, 1, t0.n_sites);
            if (e->tile_ranks) {
                join(e->j_buf + e->j_o_mq, 2, 1, t0.n_sites);
                join(e->j_buf + e->j_o_rp, 3, 2, t0.n_sites);
            }
            if (e->tile_groups) join(e->j_buf + e->j_o_gid, 4, 1, 1);
            BV_HIP(e, hipGetLastError());
            e->tile_samples_seen += (uint64_t)nk * t0.n_samples;
            continue;
        }
        BvTileScatterPlane wide[5 * BV_TILE_MANY_MAX], narrow[5 * BV_TILE_MANY_MAX];
        uint32_t nw = 0, nn = 0;
        uint64_t uw = 0, un = 0;
        for (uint32_t i = 0; i < nk; ++i) {
            const bv_slab &t = tiles[k0 + i];
            const uint64_t lo = e->tile_samples_seen, P = t.pitch;
            auto plane = [&](uint8_t *dst, const void *src, uint64_t scale, uint32_t n_rows) {
                BvTileScatterPlane p;
                p.dst = dst; p.src = static_cast<const uint8_t *>(src); p.dst_pitch = scale * JP; p.src_pitch = scale * P; p.col_off = scale * lo;
                p.width_bytes = (uint32_t)(scale * t.n_samples); p.n_rows = n_rows;
                const bool w8 = !((p.dst_pitch | p.col_off | p.src_pitch | p.width_bytes | (uint64_t)(uintptr_t)p.dst | (uint64_t)(uintptr_t)p.src) & 7u);
                if (w8) { wide[nw++] = p; const uint64_t u = (uint64_t)(p.width_bytes / 8u) * n_rows; if (u > uw) uw = u; }
                else { narrow[nn++] = p; const uint64_t u = (uint64_t)p.width_bytes * n_rows; if (u > un) un = u; }
            };
            plane(e->j_buf, t.base_strand, 1, t.n_sites);
            plane(e->j_buf + e->j_o_q, t.qual, 1, t.n_sites);
            if (e->tile_ranks) {
                plane(e->j_buf + e->j_o_mq, t.mapq, 1, t.n_sites);
                plane(e->j_buf + e->j_o_rp, t.rpr, 2, t.n_sites);
            }
            if (e->tile_groups) plane(e->j_buf + e->j_o_gid, t.group_id, 1, 1);
            e->tile_samples_seen += t.n_samples;
        }
        std::memcpy(e->h_desc[slot], wide, sizeof(BvTileScatterPlane) * nw);
        std::memcpy(e->h_desc[slot] + nw, narrow, sizeof(BvTileScatterPlane) * nn);
        BV_HIP(e, hipMemcpyAsync(e->d_desc[slot], e->h_desc[slot], sizeof(BvTileScatterPlane) * (nw + nn), hipMemcpyHostToDevice, st));
        BV_HIP(e, hipEventRecord(e->ev_desc[slot], st));
        e->desc_used[slot] = true;
        bv_launch_tile_scatter_many(e->d_desc[slot], nw, nn, uw, un, st);
        BV_HIP(e, hipGetLastError());
    }
    return mark_done(e, st);
}

int bv_engine_tiles_finish(bv_engine *e, const uint8_t *ref_base, bv_site_result *out, bv_group_result *gout,
                           uint32_t mem_kind, void *stream_) {
    if (!e || !ref_base || !out) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_finish: null argument");
    if (!e->tile_open) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_finish: no open tile job");
    if (e->tile_groups && !gout) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_tiles_finish: job has groups, gout is NULL");
    BV_HIP(e, hipSetDevice(e->cfg.device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : e->stream;
    {
        int rc = use_stream(e, st);
        if (rc != BV_OK) return rc;
    }
    const size_t S = e->tile_sites, G = e->tile_groups;
    const uint8_t *dref = ref_base;
    bv_site_result *dout = out;
    bv_group_result *dgout = gout;
    e->host_out = nullptr; e->host_gout = nullptr;
    bv_engine::StageSlot *slot = nullptr;
    if (mem_kind == BV_MEM_HOST) {
        auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
        HostPlane pl[1] = {{ref_base, S, nullptr}};
        const size_t out_b = up(S * sizeof(bv_site_result)), gout_b = up(S * G * sizeof(bv_group_result));
        uint8_t *extra = nullptr;
        int rc = stage_host_planes(e, pl, 1, out_b + gout_b, &slot, &extra, st);
        if (rc != BV_OK) return rc;
        rc = stage_publish(e, slot, st);
        if (rc != BV_OK) return rc;
        dref = pl[0].dev;
        dout = reinterpret_cast<bv_site_result *>(extra);
        dgout = G ? reinterpret_cast<bv_group_result *>(extra + out_b) : nullptr;
        e->stage_out = dout; e->stage_gout = dgout;
        e->host_out = out; e->host_gout = gout;
        e->host_out_bytes = S * sizeof(bv_site_result);
        e->host_gout_bytes = S * G * sizeof(bv_group_result);
    }
    if (e->tile_join) {
        e->tile_open = false;
        if (e->tile_samples_seen < e->tile_samples_total && !e->j_filled)  // samples announced but never delivered: uncovered cells
        {
            BV_HIP(e, hipMemset2DAsync(e->j_buf + e->tile_samples_seen, e->j_pitch, 0x08, e->tile_samples_total - e->tile_samples_seen, S, st));
            // (tagged ranks: the same cells' rank words must say "no call" too -- 0x8080: the tag's bit 15, rank 128)
            if (e->tile_ranks && (e->tile_layout & BV_SLAB_RPR_TAGGED))
                BV_HIP(e, hipMemset2DAsync(e->j_buf + e->j_o_rp + 2 * (size_t)e->tile_samples_seen, 2 * e->j_pitch, 0x80,
                                           2 * (size_t)(e->tile_samples_total - e->tile_samples_seen), S, st));
        }
        int rc = launch_passes(e, e->j_buf, e->j_buf + e->j_o_q, e->tile_ranks ? e->j_buf + e->j_o_mq : nullptr,
                               e->tile_ranks ? reinterpret_cast<const uint16_t *>(e->j_buf + e->j_o_rp) : nullptr, dref,
                               G ? e->j_buf + e->j_o_gid : nullptr, e->j_pitch, e->tile_sites, e->tile_samples_total, e->tile_groups,
                               dout, dgout, st, nullptr, false, e->tile_layout);
        if (rc == BV_OK && slot) rc = stage_release(e, slot, st);
        return rc;
    }
    BV_HIP(e, hipMemsetAsync(e->d_counters, 0, sizeof(uint32_t) * BV_CTR_PER_LAUNCH * BV_CTR_STRIDE, st));
    if (G) BV_HIP(e, hipMemsetAsync(dgout, 0, S * G * sizeof(bv_group_result), st));
    BvTileFinishArgs f;
    f.state = e->tile_state; f.maxr = e->tile_maxr; f.ref_base = dref; f.n_sites = e->tile_sites; f.n_groups = e->tile_groups;
    f.stride = e->tile_stride; f.have_ranks = e->tile_ranks ? 1u : 0u; f.min_af = e->cfg.min_af; f.tables = e->d_tables;
    f.rank_win = e->tile_rank_win; f.hg_off = e->tile_hg_off;
    f.ord_off = e->tile_ord_off; f.ovf = e->tile_ovf; f.ovf_cap = bv_engine::kOvfCap;
    f.out = dout; f.gout = dgout; f.var_list = e->d_var_list; f.counters = e->d_counters;
    bv_launch_tile_finish(f, st);
    BV_HIP(e, hipGetLastError());
    e->last_blocks = 1; e->last_ctr_base = 0; e->ctr_rot = 0;
    BV_HIP(e, hipMemcpyAsync(e->h_counters, e->d_counters, sizeof(uint32_t) * BV_CTR_WORDS * bv_engine::kCtrBlocks, hipMemcpyDeviceToHost, st));
    if (e->host_out) {
        BV_HIP(e, hipMemcpyAsync(e->host_out, e->stage_out, e->host_out_bytes, hipMemcpyDeviceToHost, st));
        if (e->host_gout && e->host_gout_bytes)
            BV_HIP(e, hipMemcpyAsync(e->host_gout, e->stage_gout, e->host_gout_bytes, hipMemcpyDeviceToHost, st));
    }
    e->tile_open = false;
    e->submitted = true;
    e->last_slot = -1;  // no pass-1/pass-2 event triplet for this realisation: bv_engine_kernel_ms has nothing to report
    if (slot) {
        int rc = stage_release(e, slot, st);
        if (rc != BV_OK) return rc;
    }
    return mark_done(e, st);
}

void *bv_engine_stream(bv_engine *e) { return e ? (void *)e->stream : nullptr; }

#ifdef BV_TEAM_DEBUG
// bv_p1s_stream_kernel (short rows): when each wave finished its static range of sites, per XCD
static void bv_stream_debug_report(const uint32_t *h) {
    const uint32_t *d = h + BV_CTR_WORDS;
    uint32_t t0 = 0; bool any = false;
    for (int b = 0; b < 512; ++b)
        if (d[4096 + b] && (!any || (int32_t)(d[4096 + b] - t0) < 0)) { t0 = d[4096 + b]; any = true; }
    if (!any) return;
    std::vector<double> v;
    struct Acc { double sum = 0; uint32_t n = 0; };
    Acc by_xcc[8], by_wave[4], by_simd[4], by_cu[16], by_se[8], by_slot[16];
    for (int w = 0; w < 2048; ++w) {
        if (!d[w] || !d[4096 + w / 4]) continue;
        const double t = (double)(int32_t)(d[w] - t0) * 0.01;
        const uint32_t hw = d[2048 + w];
        v.push_back(t);
        auto add = [&](Acc &a) { a.sum += t; a.n++; };
        add(by_xcc[d[4608 + w / 4] & 7u]); add(by_wave[w & 3]); add(by_simd[(hw >> 4) & 3u]); add(by_cu[(hw >> 8) & 15u]); add(by_se[(hw >> 13) & 7u]);
        add(by_slot[hw & 15u]);
    }
    if (v.empty()) return;
    std::sort(v.begin(), v.end());
    const size_t n = v.size();
    fprintf(stderr, "[stream debug] wave done min %.1f p10 %.1f p25 %.1f p50 %.1f p75 %.1f p90 %.1f max %.1f us (%zu waves)\n", v[0], v[n / 10], v[n / 4],
            v[n / 2], v[n * 3 / 4], v[n * 9 / 10], v[n - 1], n);
    auto show = [&](const char *nm, Acc *a, int k) {
        fprintf(stderr, "[stream debug] mean by %-12s:", nm);
        for (int i = 0; i < k; ++i) if (a[i].n) fprintf(stderr, " %d:%.0f(%u)", i, a[i].sum / a[i].n, a[i].n);
        fprintf(stderr, "\n");
    };
    show("XCD", by_xcc, 8); show("wave of group", by_wave, 4); show("SIMD", by_simd, 4); show("CU id", by_cu, 16); show("SE/SH bits", by_se, 8);
    show("wave slot", by_slot, 16);
}
#endif

#ifdef BV_TEAM_DEBUG
// bv_p1s_fused_kernel: per workgroup, when its streaming waves were done, what was left for the solvers then, when it ended
static void bv_fused_debug_report(const uint32_t *h) {
    const uint32_t *d = h + BV_CTR_WORDS;
    uint32_t t0 = 0; bool any = false;
    for (int b = 0; b < 512; ++b)
        if (d[b * 8] && (!any || (int32_t)(d[b * 8] - t0) < 0)) { t0 = d[b * 8]; any = true; }
    if (!any) return;
    const char *nm[5] = {"entry", "first streaming wave past its pass-1 rows", "last streaming wave past its pass-1 rows", "workgroup done", "last solver job done"};
    const int col[5] = {0, 1, 2, 3, 5};
    for (int j = 0; j < 5; ++j) {
        std::vector<double> v; double sum[8] = {}; uint32_t cnt[8] = {};
        for (int b = 0; b < 512; ++b) {
            if (!d[b * 8] || !d[b * 8 + col[j]]) continue;
            const double t = (double)(int32_t)(d[b * 8 + col[j]] - t0) * 0.01;
            v.push_back(t); sum[d[b * 8 + 7] & 7u] += t; cnt[d[b * 8 + 7] & 7u]++;
        }
        if (v.empty()) continue;
        std::sort(v.begin(), v.end());
        const size_t n = v.size();
        fprintf(stderr, "[fused debug] %-42s min %6.1f p10 %6.1f p50 %6.1f p90 %6.1f max %6.1f us | mean per XCD:", nm[j], v[0], v[n / 10], v[n / 2],
                v[n * 9 / 10], v[n - 1]);
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", cnt[x] ? sum[x] / cnt[x] : 0.);
        fprintf(stderr, "\n");
    }
#ifdef BV_PHASE_DEBUG
    {
        const char *pn[12] = {"wait for the slot (vmcnt)", "slot -> registers (4 ds_read_b128)", "request the next slot: the 4 DMA pieces", "tally, pass-1 slot", "tally, pass-2 slot",
                              "row epilogue, pass 1", "row epilogue, pass 2", "slots", "leaving (drain)", "inside the streaming function", "request the next slot: draw a row", "slot requested -> found landed"};
        for (int part = 0; part < 2; ++part) {
            const uint32_t *c = d + 4220 + 12 * part;
            fprintf(stderr, "[fused phases] -- streaming waves, %s\n", part ? "past their last pass-1 row" : "while they have pass-1 rows");
            for (int i = 0; i < 12; ++i) {
                if (i == 7) fprintf(stderr, "[fused phases] %-38s %u\n", pn[i], c[i]);
                else if (i == 11) fprintf(stderr, "[fused phases] %-38s %12.0f cycles  (%.0f per timed slot, %u timed)\n", pn[i], 16.0 * c[i], d[4244 + part] ? 16.0 * c[i] / d[4244 + part] : 0., d[4244 + part]);
                else if (i != 9 || part == 0) fprintf(stderr, "[fused phases] %-38s %12.0f cycles  (%.0f per slot)\n", pn[i], 16.0 * c[i], c[7] ? 16.0 * c[i] / c[7] : 0.);
            }
        }
    }
#endif
#ifdef BV_PHASE_DEBUG
    for (int l = 0; l < 2; ++l) {
        const uint32_t *j = d + 4212 + 4 * l;
        fprintf(stderr, "[fused phases] 16-lane jobs %s: %u (of them from q3: %u), %.2f sites per job, mean %.0f cycles\n", l ? "after the last pass-1 row" : "while rows stream", j[1], j[3],
                j[1] ? (double)j[2] / j[1] : 0., j[1] ? 16.0 * j[0] / j[1] : 0.);
    }
#endif
#ifdef BV_PHASE_DEBUG
    {
        const char *jn[5] = {"the entry, the summary's and the bins' loads", "phase 1: LRT, the record's first version", "its stores complete, variant sites queued",
                             "phase 2's loads", "phase 2: rank sum, QUAL, strand-bias tests"};
        const uint32_t nj = d[4213] + d[4217];
        for (int i = 0; i < 5; ++i) fprintf(stderr, "[fused phases] a 16-lane job, %-48s %8.0f cycles\n", jn[i], nj ? 16.0 * d[4250 + i] / nj : 0.);
    }
#endif
    const char *qn[3] = {"q3 entries", "q2 entries", "variant rows (or blocks of 64)"};
    for (int j = 0; j < 3; ++j) {
        std::vector<uint32_t> v;
        for (int b = 0; b < 512; ++b) if (d[b * 8]) v.push_back(j == 0 ? (d[b * 8 + 4] & 0xFFFFu) : j == 1 ? (d[b * 8 + 4] >> 16) : d[b * 8 + 6]);
        std::sort(v.begin(), v.end());
        fprintf(stderr, "[fused debug] waiting when the last streaming wave was past its pass-1 rows, %-30s: min %u p50 %u p90 %u max %u\n", qn[j], v[0],
                v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    }
}
#endif
#ifdef BV_TEAM_DEBUG
// the stamps of bv_pass1_kernel's team form (see BV_TEAM_STAMP in bv_pass1.hip): distribution over the workgroups, and per XCD
static void bv_team_debug_report(const uint32_t *h) {
    fprintf(stderr, "[team debug] team jobs %u (mean %.0f cycles)  solo solves %u (mean %.0f cycles)\n", h[BV_CTR_CANDS],
            h[BV_CTR_CANDS] ? 64.0 * h[BV_CTR_CANDS + 1] / h[BV_CTR_CANDS] : 0., h[BV_CTR_EASY3],
            h[BV_CTR_EASY3] ? 64.0 * h[BV_CTR_EASY3 + 1] / h[BV_CTR_EASY3] : 0.);
    fprintf(stderr, "[team debug] tally waves waiting for a free ring slot: %.0f cycles per workgroup (sum over its rows)\n", 64.0 * h[BV_CTR_EASY] / 1024.0);
    const uint32_t *d = h + BV_CTR_WORDS;
    const char *nm[6] = {"entry", "start barrier passed", "first row begins", "tally waves done", "phred tables in LDS", "solver wave done"};
    uint32_t t0 = 0; bool any = false;
    for (int b = 0; b < 640; ++b)
        if (d[b * 8] && (!any || (int32_t)(d[b * 8] - t0) < 0)) { t0 = d[b * 8]; any = true; }
    if (!any) return;
    for (int j = 0; j < 6; ++j) {
        std::vector<double> v; double sum[8] = {}; uint32_t cnt[8] = {};
        for (int b = 0; b < 640; ++b) {
            if (!d[b * 8]) continue;
            const double t = (double)(int32_t)(d[b * 8 + j] - t0) * 0.01;
            v.push_back(t); sum[d[b * 8 + 7] & 7u] += t; cnt[d[b * 8 + 7] & 7u]++;
        }
        std::sort(v.begin(), v.end());
        const size_t n = v.size();
        fprintf(stderr, "[team debug] %-22s min %6.1f p10 %6.1f p50 %6.1f p90 %6.1f max %6.1f us | mean per XCD:", nm[j], v[0], v[n / 10], v[n / 2],
                v[n * 9 / 10], v[n - 1]);
        for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", cnt[x] ? sum[x] / cnt[x] : 0.);
        fprintf(stderr, "\n");
    }
}
#endif

int bv_engine_wait(bv_engine *e) {
    if (!e) return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_wait: null engine");
    if (!e->submitted) return BV_OK;
    BV_HIP(e, hipSetDevice(e->cfg.device));
    for (bv_engine *l : e->lane) {
        if (!l) continue;
        const int rc = bv_engine_wait(l);
        if (rc != BV_OK) return fail(e, rc, bv_last_error(l));
    }
    if (e->last_stream == nullptr && e->used_streams.empty()) return BV_OK;  // only lanes carried work
    // every stream that carried a submit since the last wait (submits of one engine are serialised through ev_done,
    // so the counters mirrored by the LAST submit are final once all of them have drained)
    for (hipStream_t st : e->used_streams) BV_HIP(e, hipStreamSynchronize(st));
    e->used_streams.clear();
    e->used_streams.push_back(e->last_stream);
    e->done_pending = false; e->ev_done_set = false;  // everything submitted so far has finished: nothing left to order behind
    if (e->ctr_mirror_stale) {  // (the streams are idle: a plain copy)
        BV_HIP(e, hipMemcpy(e->h_counters, e->d_counters, sizeof(uint32_t) * BV_CTR_WORDS * bv_engine::kCtrBlocks, hipMemcpyDeviceToHost));
        e->ctr_mirror_stale = false;
    }
    uint32_t timed_out = 0, zero_freq = 0;
    for (uint32_t b = 0; b < bv_engine::kCtrBlocks; ++b) {
#ifdef BV_TEAM_DEBUG
        if (b == 0 && e->h_counters[BV_CTR_WORDS + 5150] == 1u) bv_stream_debug_report(e->h_counters);
        else if (b == 0 && e->h_counters[BV_CTR_WORDS + 5150] == 2u) bv_team_debug_report(e->h_counters);
        else if (b == 0 && e->h_counters[BV_CTR_WORDS + 5150] == 4u) bv_fused_debug_report(e->h_counters);
#endif
        timed_out |= e->h_counters[(size_t)b * BV_CTR_WORDS + BV_CTR_TIMEOUT];  // bit-coded BV_TMO_* flags, ORed by the kernels: OR here too
        zero_freq += e->h_counters[(size_t)b * BV_CTR_WORDS + BV_CTR_ZEROFREQ];
    }
    if (timed_out != 0 || zero_freq != 0) {
        // the error counters are sticky on the device (they accumulate over submits): reported once, then cleared
        for (uint32_t b = 0; b < bv_engine::kCtrBlocks; ++b) {
            BV_HIP(e, hipMemsetAsync(e->d_counters + (size_t)b * BV_CTR_WORDS + BV_CTR_PER_LAUNCH * BV_CTR_STRIDE, 0,
                                     sizeof(uint32_t) * (BV_CTR_WORDS - BV_CTR_PER_LAUNCH * BV_CTR_STRIDE), e->last_stream));
            e->h_counters[(size_t)b * BV_CTR_WORDS + BV_CTR_TIMEOUT] = e->h_counters[(size_t)b * BV_CTR_WORDS + BV_CTR_ZEROFREQ] = 0;
        }
        BV_HIP(e, hipStreamSynchronize(e->last_stream));
    }
    if (timed_out != 0) {
        char buf[200];
        std::snprintf(buf, sizeof buf, "pass 1: an intra-workgroup hand-off timed out (internal error; results invalid; which: %#x, see BV_TMO_* in "
                                       "csrc/bv_kernels.h)", timed_out);
        return fail(e, BV_ERR_HIP, buf);
    }
    if (zero_freq > 0) {
        char buf[160];
        std::snprintf(buf, sizeof buf,
                      "The sum of frequence of active bases must always > 0 (%u site(s); see BV_SITE_ZERO_FREQ)", zero_freq);
        return fail(e, BV_ERR_SITE, buf);  // message of src/basetype.cpp:114
    }
    return BV_OK;
}

int bv_engine_kernel_ms(bv_engine *e, float *pass1_ms, float *pass2_ms) {
    if (!e || !e->submitted) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_kernel_ms: nothing submitted");
    if (e->last_lane >= 0) return bv_engine_kernel_ms(e->lane[e->last_lane], pass1_ms, pass2_ms);
    if (e->last_slot < 0)
        return fail(e, BV_ERR_INVALID_ARG, "bv_engine_kernel_ms: the last job recorded no pass timings (per-site-tally tile job)");
    float a = 0.f, b = 0.f;
    hipEvent_t *t = e->ring[e->last_slot];
    BV_HIP(e, hipEventSynchronize(t[2]));
    BV_HIP(e, hipEventElapsedTime(&a, t[0], t[1]));
    BV_HIP(e, hipEventElapsedTime(&b, t[1], t[2]));
    if (pass1_ms) *pass1_ms = a;
    if (pass2_ms) *pass2_ms = b;
    return BV_OK;
}

int bv_engine_timing_reset(bv_engine *e) {
    if (!e) return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_timing_reset: null engine");
    BV_HIP(e, hipSetDevice(e->cfg.device));
    int rc = drain_timings(e, true);
    if (rc != BV_OK) return rc;
    e->acc1_ms = e->acc2_ms = e->acc_stream_ms = 0.;
    e->acc_n = 0;
    for (bv_engine *l : e->lane)
        if (l && (rc = bv_engine_timing_reset(l)) != BV_OK) return rc;
    return BV_OK;
}

int bv_engine_timing_get(bv_engine *e, double *pass1_total_ms, double *pass2_total_ms, uint32_t *n_submits) {
    if (!e) return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_timing_get: null engine");
    BV_HIP(e, hipSetDevice(e->cfg.device));
    int rc = drain_timings(e, true);
    if (rc != BV_OK) return rc;
    double a1 = e->acc1_ms, a2 = e->acc2_ms;
    uint32_t an = e->acc_n;
    for (bv_engine *l : e->lane) {  // the lanes' submits are this engine's
        if (!l) continue;
        double x = 0, y = 0; uint32_t m = 0;
        if ((rc = bv_engine_timing_get(l, &x, &y, &m)) != BV_OK) return rc;
        a1 += x; a2 += y; an += m;
    }
    if (pass1_total_ms) *pass1_total_ms = a1;
    if (pass2_total_ms) *pass2_total_ms = a2;
    if (n_submits) *n_submits = an;
    return BV_OK;
}

int bv_engine_timing_get_ex(bv_engine *e, double *stream_total_ms, double *pass1_total_ms, double *pass2_total_ms,
                            uint32_t *n_submits) {
    if (!e) return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_timing_get_ex: null engine");
    BV_HIP(e, hipSetDevice(e->cfg.device));
    int rc = drain_timings(e, true);
    if (rc != BV_OK) return rc;
    double as = e->acc_stream_ms, a1 = e->acc1_ms, a2 = e->acc2_ms;
    uint32_t an = e->acc_n;
    for (bv_engine *l : e->lane) {
        if (!l) continue;
        double w = 0, x = 0, y = 0; uint32_t m = 0;
        if ((rc = bv_engine_timing_get_ex(l, &w, &x, &y, &m)) != BV_OK) return rc;
        as += w; a1 += x; a2 += y; an += m;
    }
    if (stream_total_ms) *stream_total_ms = as;
    if (pass1_total_ms) *pass1_total_ms = a1;
    if (pass2_total_ms) *pass2_total_ms = a2;
    if (n_submits) *n_submits = an;
    return BV_OK;
}

// Make `stream` wait for every submit issued so far.  Without BV_FLAG_LANES the submits already ran on the stream they
// were given; with it they run on the lanes' own streams and a consumer ordered on a stream (a gather of the records,
// a copy) calls this first.
int bv_engine_join(bv_engine *e, void *stream_) {
    if (!e) return fail(nullptr, BV_ERR_INVALID_ARG, "bv_engine_join: null engine");
    BV_HIP(e, hipSetDevice(e->cfg.device));
    hipStream_t st = stream_ ? (hipStream_t)stream_ : e->stream;
    for (bv_engine *l : e->lane) {
        if (!l) continue;
        int rc = flush_done(l);
        if (rc != BV_OK) return fail(e, rc, bv_last_error(l));
        if (l->ev_done_set) BV_HIP(e, hipStreamWaitEvent(st, l->ev_done, 0));
    }
    if (st != e->last_stream) {
        int rc = flush_done(e);
        if (rc != BV_OK) return rc;
        if (e->ev_done_set) BV_HIP(e, hipStreamWaitEvent(st, e->ev_done, 0));
    }
    return BV_OK;
}

int bv_host_log_probe(double *table) {
    double t[BV_HOSTLOG_N + 2];
    const bool ok = load_host_log_table(t);
    if (ok && table) std::memcpy(table, t, sizeof(double) * BV_HOSTLOG_N);
    return ok ? 1 : 0;
}

double bv_host_log_eval(const double *table, double x) { return host_log_restated(x, table); }

int bv_engine_host_log_exact(const bv_engine *e) { return e ? e->host_log_exact : 0; }

__global__ void bv_host_log_eval_kernel(const BvTables *tables, const double *x, double *y, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = bv_log_host(x[i], tables->hostlog);
}

int bv_engine_host_log_eval(bv_engine *e, const double *x, double *y, uint32_t n) {
    if (!e || !x || !y) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_host_log_eval: null argument");
    if (!e->host_log_exact) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_host_log_eval: the host log table was not verified on this host");
    if (n == 0) return BV_OK;
    BV_HIP(e, hipSetDevice(e->cfg.device));
    double *d = nullptr;
    BV_HIP(e, hipMalloc(&d, sizeof(double) * 2 * (size_t)n));
    hipError_t st = hipMemcpy(d, x, sizeof(double) * n, hipMemcpyHostToDevice);
    if (st == hipSuccess) {
        bv_host_log_eval_kernel<<<(n + 255u) / 256u, 256, 0, e->stream>>>(e->d_tables, d, d + n, n);
        st = hipGetLastError();
    }
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
    if (st == hipSuccess) st = hipMemcpy(y, d + n, sizeof(double) * n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (st != hipSuccess) return fail(e, BV_ERR_HIP, std::string("bv_engine_host_log_eval: ") + hipGetErrorString(st));
    return BV_OK;
}

int bv_engine_last_launch_form(bv_engine *e, uint32_t *form) {
    if (!e || !form) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_last_launch_form: null argument");
    if (e->last_lane >= 0) return bv_engine_last_launch_form(e->lane[e->last_lane], form);
    *form = e->last_form;
    return BV_OK;
}

int bv_engine_last_variant_count(bv_engine *e, uint32_t *n_variant) {
    if (!e || !n_variant) return fail(e, BV_ERR_INVALID_ARG, "bv_engine_last_variant_count: null argument");
    if (e->last_lane >= 0) return bv_engine_last_variant_count(e->lane[e->last_lane], n_variant);
    uint32_t n = 0;
    for (uint32_t b = 0; b < e->last_blocks; ++b) n += e->h_counters[(size_t)(e->last_ctr_base + b) * BV_CTR_WORDS + BV_CTR_VARIANTS];
    *n_variant = n;
    return BV_OK;
}

const char *bv_last_error(const bv_engine *e) {
    static thread_local std::string copy;
    if (e) {
        std::lock_guard<std::mutex> lk(e->mu);
        copy = e->err;
    } else {
        std::lock_guard<std::mutex> lk(g_err_mu);
        copy = g_err;
    }
    return copy.c_str();
}

// ---- NUMA placement of host buffers ---------------------------------------------------------------------------------
// The node a GPU's PCIe function hangs off, from sysfs (hipDeviceGetPCIBusId -> /sys/bus/pci/devices/<bdf>/numa_node).
int bv_device_numa_node(int device, char *pci_bdf, size_t pci_bdf_len) {
    char bdf[32] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    for (char *c = bdf; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
    if (pci_bdf && pci_bdf_len) std::snprintf(pci_bdf, pci_bdf_len, "%s", bdf);
    char path[128];
    std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE *f = std::fopen(path, "r")) {
        if (std::fscanf(f, "%d", &node) != 1) node = -1;
        std::fclose(f);
    }
    return node;
}

// The calling thread keeps those of its CPUs that belong to the GPU's node: memory it allocates and first touches from now on
// (pinned tiles, staging vectors) is node-local under the default policy.  Returns the node, or -1 with the mask unchanged
// (no node reported, no node CPU list, or none of the node's CPUs in the current mask).
int bv_bind_thread_to_device_node(int device) {
    const int node = bv_device_numa_node(device, nullptr, 0);
    if (node < 0) return -1;
    char path[128];
    std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = std::fopen(path, "r");
    if (!f) return -1;
    char buf[4096] = {0};
    const size_t got = std::fread(buf, 1, sizeof buf - 1, f);
    std::fclose(f);
    buf[got] = 0;
    cpu_set_t cur, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof cur, &cur) != 0) return -1;
    int n_set = 0;
    char *save = nullptr;  // (strtok_r: bv_call's per-GPU worker threads bind themselves concurrently)
    for (char *tok = strtok_r(buf, ",\n", &save); tok; tok = strtok_r(nullptr, ",\n", &save)) {  // "0-63,128-191"
        int lo = 0, hi = 0;
        const int k = std::sscanf(tok, "%d-%d", &lo, &hi);
        if (k < 1) continue;
        if (k == 1) hi = lo;
        for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &cur)) { CPU_SET(c, &want); ++n_set; }
    }
    if (n_set == 0) return -1;
    if (sched_setaffinity(0, sizeof want, &want) != 0) return -1;
    return node;
}

int bv_synth_fill(int device, const bv_synth_params *p, uint32_t n_sites, uint32_t n_samples, uint64_t pitch,
                  uint8_t *base_strand, uint8_t *qual, uint8_t *mapq, uint16_t *rpr, uint8_t *ref_base, void *stream) {
    if (!p || !base_strand || !qual || !ref_base || n_sites == 0 || n_samples == 0 || pitch < n_samples || (pitch & 15ull))
        return fail(nullptr, BV_ERR_INVALID_ARG, "bv_synth_fill: bad argument");
    BV_HIP(nullptr, hipSetDevice(device));
    bv_launch_synth(*p, n_sites, n_samples, pitch, base_strand, qual, mapq, rpr, ref_base, (hipStream_t)stream);
    BV_HIP(nullptr, hipGetLastError());
    return BV_OK;
}

}  // extern "C"
