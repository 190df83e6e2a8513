#!/usr/bin/env python3
"""bench.py -- genomic sites/s through the basetype engine on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[2], the configuration the metric and the north-star target
are quoted on): synthetic NIPT-scale pileup, 100,000 samples per site, generated on the
device.  The 1 M-site job does not fit HBM at once (5 B/cell = 500 GB),
so it is processed in HBM-resident batches (default 131,072 sites = 65.6 GB of planes; the 1 M
sites are 8 such steps; two distinct batches are resident, 131 GB of the 288 GB; a launch carries ~0.09 ms
of fill/drain, which is why big batches are the default): ONE STEP = one pass of the whole hot path
(pass 1: tally + EM/LRT/AF/QUAL/strand bias/BaseQRankSum for every site; pass 2: MQ and
ReadPos rank sums for the variant sites) over one batch of --batch-sites sites, followed
(N > 1) by the gather of the batch's result records to rank 0 (RCCL over xGMI).
Inputs are resident in HBM before the timed region starts; several distinct batches are
cycled so no step re-reads cache-resident data (each batch is >> the 256 MiB Infinity Cache).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (pass 1): algorithmic
bytes = 2 B/cell (u8 call plane + u8 phred plane) x sites x samples per launch, divided by the
average launch duration measured with HIP events on the launch stream.  `cpu_baseline` times
the reference's own code (oracle/_ref, kind "reference"; or the C restatement, kind "port",
if that library is absent) on the host cores over a bounded sample of the same batch.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; --tile-job: 6 jobs)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=0, help="samples per site (row length); default 100,000 (--tile-job: 1,000,000)")
    ap.add_argument("--batch-sites", type=int, default=131072, help="sites per step and per GPU")
    ap.add_argument("--coverage", type=float, default=0.08)
    ap.add_argument("--distinct-batches", type=int, default=2)
    ap.add_argument("--no-rank-planes", action="store_true", help="omit mapq/rpr planes (pass 2 skipped)")
    ap.add_argument("--rank-layout", choices=("tagged", "plain"), default="tagged",
                    help="layout of the u16 read-position-rank plane the synthetic producer writes: tagged = BV_SLAB_RPR_TAGGED "
                         "(rank | base << 13 | nocall << 15: every rank here is <= 100, so the producer may choose it; pass 2 then "
                         "reads SURVEY 8d's 3 B/cell: mapq + rpr); plain = the rank alone (pass 2 re-reads the call plane: 4 B/cell)")
    ap.add_argument("--cpu-sites", type=int, default=0, help="sites in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs1", action="store_true",
                    help="skip the `configs1` object of the default run: BASELINE configs[1] (100,000 sites x 10,000 samples, one "
                         "batch) timed after the headline on the same box (never `value`)")
    ap.add_argument("--seed", type=int, default=0xBA5E7A7)
    ap.add_argument("--tally-only", action="store_true", help="diagnostic: time pass 1 without its solver")
    ap.add_argument("--flags", type=int, default=0, help="diagnostic BV_FLAG_* bits (ablation)")
    ap.add_argument("--chain", type=int, default=1,
                    help="batches handed to the engine per step as ONE chained launch (bv_engine_submit_many): the tail of a small "
                         "batch runs under the stream of the next; 1 = one submit per batch")
    ap.add_argument("--streams", type=int, default=1,
                    help="engines/HIP streams used round-robin for consecutive batches (tails of one batch overlap the next)")
    ap.add_argument("--lanes", type=int, default=1, choices=(1, 2),
                    help="2: the engine runs consecutive submits on two internal lanes (BV_FLAG_LANES: a second stream and scratch "
                         "set inside ONE engine), so the solve kernels of a batch run under the streaming kernels of the next; the "
                         "timed region is still K back-to-back steps on resident batches")
    ap.add_argument("--sparse-timing", action="store_true",
                    help="BV_FLAG_SPARSE_TIMING: the engine records its per-pass timing events for one launch in eight (the four "
                         "events cost ~15 us per launch); the kernel averages of the line then rest on those launches")
    ap.add_argument("--groups", type=int, default=0,
                    help="diagnostic: G pop-groups (random membership, ~15 %% of the samples in none): adds the per-group calls of pass 2")
    ap.add_argument("--with-tile-mode", action="store_true",
                    help="also time BASELINE config #5's shape: sample-axis tiles (--tile-width samples each) accumulated in "
                         "HBM, from device-resident tiles and from pinned host memory over PCIe (never `value`)")
    ap.add_argument("--tile-width", type=int, default=200, help="samples per tile (the reference's --batch-count)")
    ap.add_argument("--tile-sites", type=int, default=16384, help="sites per tile job (at most --batch-sites)")
    ap.add_argument("--tile-job", action="store_true",
                    help="BASELINE configs[4]'s shape AS the timed workload: one STEP = one tile job per rank -- --tile-sites sites x "
                         "--samples samples (default here: 1,000,000) delivered as --tile-width-sample tiles from pinned, NUMA-local "
                         "host memory (bv_engine_tiles_add), joined in HBM, both passes, records gathered to rank 0; every rank owns "
                         "its contiguous site range of the job")
    ap.add_argument("--tile-distinct", type=int, default=64, help="distinct host tiles kept resident per rank (cycled over the job)")
    ap.add_argument("--no-packed-tiles", action="store_true", help="--tile-job: skip the packed-tile leg (bv_engine_tiles_add_sparse) that runs beside the dense job")
    ap.add_argument("--verify-sites", type=int, default=256,
                    help="N > 1: rank 0 re-runs the first V sites of EVERY rank's last batch on its own device (the synthetic rows "
                         "are stateless in the global site index) and compares them with the gathered records byte for byte "
                         "(config.ranks_verified); 0 = skip")
    ap.add_argument("--with-host-path", action="store_true",
                    help="also time the PCIe-inclusive path: pinned host planes staged by the engine (never `value`)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --batch-sites per GPU per step (default); strong: --batch-sites is the WHOLE job's batch "
                         "per step and every rank takes its contiguous 1/N of it (the fixed 1 M-site job of BASELINE "
                         "configs[3] = 8 steps of 131072 sites whatever N)")
    args = ap.parse_args()
    if args.samples <= 0:
        args.samples = 1000000 if args.tile_job else 100000
    if args.tile_job:
        if args.chain > 1 or args.lanes > 1 or args.streams > 1 or args.groups or args.scaling != "weak" or args.no_rank_planes:
            ap.error("--tile-job runs plain tile jobs: no --chain / --lanes / --streams / --groups / --scaling strong / --no-rank-planes")
    if args.steps is None:
        args.steps = 6 if args.tile_job else 20  # (a 16,384-site x 1 M-sample tile job moves 85 GB over the host link: ~1.6 s)
    return args


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a torch.distributed environment: start the N ranks ourselves,
    the way the driver does, as a CHILD process tree (`python -m torch.distributed.run ...`), relay rank 0's JSON
    line (the children inherit stdout/stderr) and exit with the launcher's code.  Called before this process has
    touched torch.cuda or HIP, and it never execs: the parent only waits.  (Reference analogue: the in-process
    fan-out of _variants_discovery, src/basetype_caller.cpp:469-525.)"""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env["BASEVAR_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print("[bench] the %d-rank launch failed (exit code %d); see the ranks' messages above" % (args.gpus, rc),
              file=sys.stderr)
    sys.exit(rc if rc != 0 else 0)


def host_cpu():
    """(model name, physical cores this process may run on, logical CPUs it may run on) from /proc/cpuinfo + the affinity mask."""
    allowed = sorted(os.sched_getaffinity(0))
    model, cores, cur = "unknown", set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if cur.get("processor") in allowed:
                    cores.add((cur.get("physical id", 0), cur.get("core id", cur.get("processor"))))
                cur = {}
                continue
            k, v = [t.strip() for t in line.split(":", 1)]
            if k == "model name":
                model = v
            elif k in ("processor", "physical id", "core id"):
                cur[k] = int(v)
        if cur.get("processor") in allowed:
            cores.add((cur.get("physical id", 0), cur.get("core id", cur.get("processor"))))
    except OSError:
        pass
    return model, (len(cores) or len(allowed)), len(allowed)


def cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max, v1 cfs quota), or None if unlimited: the
    GPU boxes of this pool show 256 logical CPUs and `1600000 100000` -- 16 CPUs; threads beyond that only take turns."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(torch, planes, n_samples, maf, want_sites, rank_mask=0xFFFF):
    """Times the reference's per-site path on the host cores over rows copied back from HBM: an all-core leg (one thread per
    physical core, static site-range partition as in _variants_discovery, src/basetype_caller.cpp:489-510) and a
    single-thread leg, each at least ~2 s of wall per repeat, median of 3 repeats."""
    import numpy as np
    import oracle
    try:
        oracle.build(with_ref=True)
    except Exception as ex:  # the prebuilt libraries travel with the repo; a rebuild is optional
        print("[bench] oracle build skipped: %s" % ex, file=sys.stderr)
    if oracle.ref_available():
        chk, kind = oracle.Reference(), "reference"
    else:
        chk, kind = oracle.Restatement(), "port"
    model, physical, logical = host_cpu()
    quota = cpu_quota()
    # one thread per physical core the process can actually RUN on: a CPU quota below the core count (cgroup cpu.max) means
    # more threads only take turns (measured on this pool: 128 cores shown, quota 16 -> the all-core leg scaled 9.4 x)
    threads = max(1, physical if quota is None else min(physical, int(quota + 0.5)))
    bs, q, mq, rp, ref = planes
    S = bs.shape[0]

    def sub(k):
        idx = np.linspace(0, S - 1, num=k).astype(np.int64)  # spread over the 20-site class cycle
        ti = torch.from_numpy(idx).to(bs.device)
        d = {"base_strand": bs[ti].cpu().numpy(), "qual": q[ti].cpu().numpy(), "ref_base": ref[ti].cpu().numpy(),
             "n_samples": n_samples}
        if mq is not None:
            d["mapq"] = mq[ti].cpu().numpy()
            d["rpr"] = rp[ti].cpu().numpy().view(np.uint16) & np.uint16(rank_mask)  # (the tagged layout's call bits are not the reference's)
        return d, idx

    last = {}

    def one_pass(d, nthreads):
        """wall seconds of the path for one pass over the sample"""
        t0 = time.perf_counter()
        if kind == "reference":
            last["rec"], _, secs = chk.run_timed(d, maf, n_threads=nthreads)
            # the threads run concurrently: the path's wall time is the slowest thread's path time
            # (the driver's slab -> BatchInfo conversion is not part of the reference path)
            return float(secs.max())
        last["rec"], _ = chk.run(d, maf, n_threads=nthreads)
        return time.perf_counter() - t0

    def leg(d, nthreads, min_wall=2.0, repeats=3):
        """sites/s: median over `repeats` of (passes x sites) / (sum of the passes' path seconds), each repeat >= min_wall"""
        n = d["base_strand"].shape[0]
        first = one_pass(d, nthreads)  # also the calibration (and a warm-up of the page cache / thread pool)
        passes = max(1, int(np.ceil(min_wall / max(first, 1e-6))))
        rates, walls = [], []
        for _ in range(repeats):
            w = sum(one_pass(d, nthreads) for _ in range(passes))
            rates.append(n * passes / w)
            walls.append(w)
        med = float(np.median(rates))
        return med, (max(rates) - min(rates)) / med, passes, float(np.median(walls))

    n_all = want_sites or max(1024, 32 * threads)  # >= 32 sites per thread: the static partition's slowest thread sets the pace
    n_all = max(threads, ((min(S, n_all) + threads - 1) // threads) * threads)
    d_all, idx_all = sub(n_all)
    rate_all, spread_all, passes_all, wall_all = leg(d_all, threads)
    cpu_records, cpu_idx = last["rec"], idx_all
    n_one = min(S, 512)
    d_one, _ = sub(n_one)
    rate_one, spread_one, passes_one, wall_one = leg(d_one, 1)
    return {
        "value": rate_all, "unit": "sites/s", "cores": threads, "kind": kind,
        "model": model, "physical_cores": physical, "logical_cpus": logical, "cpu_quota": quota, "repeats": 3,
        "spread": spread_all, "wall_s_per_repeat": wall_all, "passes_per_repeat": passes_all,
        "sample": "%d of the batch's %d sites (same synthetic rows and class mix, copied back from HBM), %d samples/site, %d host threads "
                  "(one per physical core the container's CPU quota lets run), static site-range partition, in-memory BatchInfo (no text parsing), timing BaseType ctor "
                  "+ lrt + strand_bias x2 + 3 rank sums; every repeat = %d passes over the sample (%.1f s of wall), median of 3 "
                  "repeats; single thread: %d sites x %d passes per repeat (%.1f s), %.1f sites/s" % (
                      n_all, S, n_samples, threads, passes_all, wall_all, n_one, passes_one, wall_one, rate_one),
        "single_thread_value": rate_one, "single_thread_spread": spread_one, "single_thread_sites": n_one,
    }, cpu_records, cpu_idx


def traffic_record(kernel_name, variant, sites_per_launch, n_samples, rank_layout, chain=1):
    """(HBM bytes per launch, source) of the committed PMC record (profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE of the same kernel VARIANT on the same shape, tools/summarize_profiles.py tkey()), or (None, None)."""
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        tjs = json.load(open(tfile))
    except Exception:
        return None, None
    key = "%s%s|%dx%d" % (kernel_name, variant, sites_per_launch, n_samples) + ("|" + rank_layout if variant == "<p2rows>" else "") + \
          ("|chain%d" % chain if chain > 1 else "")
    keys = (key,) + (("pass1_%dx%d" % (sites_per_launch, n_samples),) if chain == 1 and kernel_name == "bv_pass1_kernel" else ())
    for k in keys:
        if k in tjs:
            return tjs[k]["hbm_bytes_per_launch"], tjs[k].get("source", "profiles/pmc_traffic.json[%s] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, round 1)" % k)
    return None, None


def configs1_leg(torch, basevar_amd, capi, dev, device_index, args, layout):
    """BASELINE configs[1] under the same clock as the headline: ONE batch of 100,000 sites x 10,000 samples (all four planes,
    both passes, HBM-resident, coverage as the headline), warm-up, then `steps` back-to-back submits bracketed by synchronize.
    Returns the `configs1` object of the line: whole-job sites/s, the dominant kernel as the engine names it, its fraction of the
    HBM peak over section 8d's bytes (2 B/cell + 3 B/cell of the variant rows) and the committed PMC traffic over those bytes."""
    S, N = 100000, 10000
    pitch = (N + 255) // 256 * 256
    bs = torch.empty((S, pitch), dtype=torch.uint8, device=dev); q = torch.empty_like(bs); mq = torch.empty_like(bs)
    rp = torch.empty((S, pitch), dtype=torch.int16, device=dev); ref = torch.empty(S, dtype=torch.uint8, device=dev)
    basevar_amd.synth_fill(device_index, S, N, pitch, bs.data_ptr(), q.data_ptr(), ref.data_ptr(), mq.data_ptr(), rp.data_ptr(),
                           seed=args.seed, site_offset=0, coverage=args.coverage, layout=layout)
    out = torch.zeros(S * basevar_amd.SITE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng = basevar_amd.BaseTypeEngine(max_sites=S, min_af_value=basevar_amd.min_af(N), device=device_index)
    steps = 40

    def one():
        eng.submit_ptrs(S, N, pitch, bs.data_ptr(), q.data_ptr(), ref.data_ptr(), out.data_ptr(), mq.data_ptr(), rp.data_ptr(), layout=layout)
    for _ in range(5):
        one()
    eng.wait()
    torch.cuda.synchronize()
    eng.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    eng.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st_ms, p1_ms, p2_ms, nsub = eng.timing_get_ex()
    nvar = eng.last_variant_count()
    form = eng.last_launch_form()
    eng.close()
    fused = bool(form & capi.BV_FORM_SHORT_ROWS) and bool(form & capi.BV_FORM_ONE_KERNEL)
    fused_p2 = bool(form & capi.BV_FORM_PASS2_FUSED)
    kernel = "bv_p1s_fused_kernel" if fused else "bv_p1s_stream_kernel"
    variant = ("<p2rows>" if fused_p2 else "<p1only>") if fused else ""
    algo = 2.0 * S * N + (3.0 * nvar * N if fused_p2 else 0.0)
    avg_s = st_ms / max(nsub, 1) / 1e3
    lay = "tagged" if layout else "plain"
    traffic, src = traffic_record(kernel, variant, S, N, lay)
    return {"workload": "BASELINE configs[1]: 100,000 sites x 10,000 samples, coverage %.2f, one HBM-resident batch, all four planes (rank layout %s), both passes" % (args.coverage, lay),
            "sites_per_s": S * steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps, "kernel": kernel + variant,
            "kernel_avg_ms": avg_s * 1e3, "pass1_avg_ms": p1_ms / max(nsub, 1), "pass2_avg_ms": p2_ms / max(nsub, 1),
            "algorithmic_bytes_per_launch": algo, "frac": algo / avg_s / 1e9 / HBM_PEAK_GBS, "variant_sites": nvar,
            "traffic": traffic, "traffic_ratio": (traffic / algo if traffic else None), "traffic_source": src}


def parity_on_sample(gpu, cpu):
    """Max relative error of the float fields and exactness of the integer fields, GPU records vs the
    CPU baseline's records of the same sites (reported next to the timing)."""
    import numpy as np

    def rel(a, b):
        a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
        m = ~(np.isnan(a) & np.isnan(b))
        with np.errstate(invalid="ignore", divide="ignore"):
            e = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
        e = np.where((a == b) | ~m, 0.0, e)
        return float(np.nanmax(e)) if e.size else 0.0

    def rel_or_abs(a, b, atol=1e-9):
        a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
        with np.errstate(invalid="ignore", divide="ignore"):
            e = np.abs(a - b)
            e = np.where(e <= atol, 0.0, e / np.maximum(np.abs(b), 1e-300))
        e = np.where((a == b) | (np.isnan(a) & np.isnan(b)), 0.0, e)
        return float(np.nanmax(e)) if e.size else 0.0
    ints_ok = all(bool(np.array_equal(gpu[f], cpu[f])) for f in ("depth", "total_depth", "cvg_sb", "var_sb", "n_alt", "alt"))
    return {"sites": int(len(cpu)), "variant_sites": int(((cpu["status"] & 2) != 0).sum()), "integer_fields_bit_exact": ints_ok,
            "max_rel_err_af": rel(gpu["af"], cpu["af"]), "max_rel_err_qual": rel(gpu["qual"], cpu["qual"]),
            "max_rel_err_qd": rel(gpu["qd"], cpu["qd"]),
            "max_rel_err_fs_sor_ranksums": max(rel_or_abs(gpu[f], cpu[f]) for f in (
                "cvg_fs", "var_fs", "cvg_sor", "var_sor", "mq_ranksum", "rpr_ranksum", "bq_ranksum"))}


class TileRig:
    """BASELINE configs[4]'s shape on one rank: the sample axis of `St` sites arrives as `n_tiles` tiles of `W` samples
    (the reference's --batch-count batchfiles, src/basetype_caller.cpp:419-453, re-joined per site at :589-601), cut from a
    synthetic source slab and cycled (`res` distinct tiles resident).  Host tiles are ONE pinned allocation each (the layout of
    bv_tile_packed_layout: a tile crosses the link as one copy), allocated and first touched while the thread is bound to the
    CPUs of the GPU's NUMA node (bv_bind_thread_to_device_node), so every rank streams from node-local DRAM."""

    def __init__(self, torch, eng, dev, device_index, St, W, n_tiles, res, src, host=True, layout=0):
        import basevar_amd
        from basevar_amd import _capi
        self.torch, self.eng, self.lib, self.capi = torch, eng, eng._lib, _capi
        self.St, self.W, self.Wp, self.n_tiles, self.res = St, W, (W + 15) // 16 * 16, n_tiles, res
        self.n_samples = n_tiles * W
        self.layout = layout
        self.device_index = device_index
        bs0, q0, mq0, rp0, ref0 = src
        self.ref = ref0[:St].contiguous()

        def cut(t, lo, dt):
            o = torch.zeros((St, self.Wp), dtype=dt, device=dev)
            o[:, :W] = t[:St, lo:lo + W]
            return o
        self.dtiles = [(cut(bs0, k * W, torch.uint8), cut(q0, k * W, torch.uint8), cut(mq0, k * W, torch.uint8),
                        cut(rp0, k * W, torch.int16)) for k in range(res)]
        self.numa_node = self.lib.bv_device_numa_node(device_index, None, 0)
        self.bound_node = -1
        self.htiles = []
        if host:
            _, offs, tot = basevar_amd.tile_packed_layout(St, W, True, False)
            keep = os.sched_getaffinity(0)
            self.bound_node = self.lib.bv_bind_thread_to_device_node(device_index)
            try:
                for tb, tq, tm_, tr in self.dtiles:
                    buf = torch.zeros(tot, dtype=torch.uint8).pin_memory()
                    views = []
                    for o, t in zip(offs[:4], (tb, tq, tm_, tr)):
                        nb_ = t.numel() * t.element_size()
                        v = buf[o:o + nb_].view(t.dtype).view(t.shape)
                        v.copy_(t)
                        views.append(v)
                    self.htiles.append(tuple(views) + (buf,))
            finally:
                os.sched_setaffinity(0, keep)  # (the CPU baseline and the launch threads want every core again)
            torch.cuda.synchronize()
        self._slabs = {}
        self.ptiles = []       # the same tiles as their covered cells only (bv_engine_tiles_add_sparse), pinned, one allocation each
        self.packed_bytes = 0  # host bytes of ONE pass over the `res` distinct packed tiles

    def build_packed(self):
        """every distinct host tile again as a packed tile: per site the run of its covered cells (sample, call, phred, mapq, rank:
        7 bytes per covered cell), laid out by bv_sparse_tile_packed_layout in ONE pinned allocation, first touched on the GPU's
        NUMA node like the dense tiles; built from the device tiles (torch.nonzero walks row-major: sites in order)"""
        import ctypes as C
        torch = self.torch
        keep = os.sched_getaffinity(0)
        self.lib.bv_bind_thread_to_device_node(self.device_index)
        try:
            for tb, tq, tm_, tr in self.dtiles:
                cb = tb[:, :self.W]
                nz = torch.nonzero(cb != 8)
                rows, cols = nz[:, 0], nz[:, 1]
                E = int(rows.numel())
                offs = (C.c_uint64 * 7)(); total = C.c_uint64()
                rc = self.lib.bv_sparse_tile_packed_layout(self.St, E, self.W, 1, 0, offs, C.byref(total))
                assert rc == 0
                buf = torch.zeros(total.value, dtype=torch.uint8).pin_memory()

                def put(k, t, dt):
                    t = t.to(dt).contiguous().cpu()
                    v = buf[offs[k]:offs[k] + t.numel() * t.element_size()].view(dt)
                    v.copy_(t)
                    return v
                rs = torch.zeros(self.St + 1, dtype=torch.int32)
                rs[1:] = torch.cumsum(torch.bincount(rows.cpu(), minlength=self.St), 0).to(torch.int32)
                vr = put(0, rs, torch.int32)
                vs = put(1, cols, torch.int16)
                vb = put(2, cb[rows, cols], torch.uint8)
                vq = put(3, tq[:, :self.W][rows, cols], torch.uint8)
                vm = put(4, tm_[:, :self.W][rows, cols], torch.uint8)
                rk = tr[:, :self.W][rows, cols].to(torch.int32) & (0x1FFF if self.layout else 0xFFFF)  # plain ranks: the engine tags the joined rows itself
                vk = put(5, rk, torch.int16)
                self.ptiles.append((vr, vs, vb, vq, vm, vk, E, buf))
                self.packed_bytes += int(total.value)
        finally:
            os.sched_setaffinity(0, keep)
        torch.cuda.synchronize()
        self._sparse = [self.capi.SparseTile(self.St, self.W, t[6], 0, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(),
                                             t[5].data_ptr(), None, self.capi.BV_MEM_HOST, self.layout) for t in (self.ptiles[k % self.res] for k in range(self.n_tiles))]

    def job_packed(self, out_ptr, stream=0):
        """one tile job from the packed host tiles"""
        import ctypes as C
        eng, lib = self.eng, self.lib
        rc = lib.bv_engine_tiles_begin(eng._h, self.St, self.n_samples, 0, 1)
        assert rc == 0, eng._err()
        st = C.c_void_p(stream) if stream else None
        for t in self._sparse:
            rc = lib.bv_engine_tiles_add_sparse(eng._h, C.byref(t), st)
            assert rc == 0, eng._err()
        rc = lib.bv_engine_tiles_finish(eng._h, self.ref.data_ptr(), out_ptr, None, self.capi.BV_MEM_DEVICE, st)
        assert rc == 0, eng._err()

    def packed_host_bytes_per_job(self):
        return self.packed_bytes * self.n_tiles // max(1, self.res)

    def slabs(self, kind):
        """the job's n_tiles bv_slab descriptors (ctypes), built once per kind"""
        if kind not in self._slabs:
            tiles = self.htiles if kind == self.capi.BV_MEM_HOST else self.dtiles
            self._slabs[kind] = [self.capi.Slab(self.St, self.W, self.Wp, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                                t[3].data_ptr(), None, None, 0, kind, self.layout) for t in (tiles[k % self.res] for k in range(self.n_tiles))]
        return self._slabs[kind]

    def job(self, out_ptr, kind, many=False, stream=0):
        """one tile job queued on `stream` (0: the engine's own): begin, every tile, finish -> records at device `out_ptr`"""
        import ctypes as C
        eng, lib = self.eng, self.lib
        rc = lib.bv_engine_tiles_begin(eng._h, self.St, self.n_samples, 0, 1)
        assert rc == 0, eng._err()
        sl = self.slabs(kind)
        st = C.c_void_p(stream) if stream else None
        if many:  # device-resident tiles: one launch per 256 tiles (bv_engine_tiles_add_many)
            eng.tiles_add_many(sl, stream=stream)
        else:
            for t in sl:
                rc = lib.bv_engine_tiles_add(eng._h, C.byref(t), st)
                assert rc == 0, eng._err()
        rc = lib.bv_engine_tiles_finish(eng._h, self.ref.data_ptr(), out_ptr, None, self.capi.BV_MEM_DEVICE, st)
        assert rc == 0, eng._err()

    def host_bytes_per_job(self):
        return 5 * self.St * self.n_tiles * self.Wp

    def joined_rows(self, V, dtiles=None):
        """the first V sites as ordinary rows [V][pitch] on the device (what the tiles of a job join to), for verification"""
        torch = self.torch
        tiles = dtiles if dtiles is not None else self.dtiles
        pitch = (self.n_samples + 255) // 256 * 256
        planes = []
        for j, (dt, fill) in enumerate(((torch.uint8, 8), (torch.uint8, 0), (torch.uint8, 0), (torch.int16, 0))):
            o = torch.full((V, pitch), fill, dtype=dt, device=tiles[0][j].device)
            for k in range(self.n_tiles):
                o[:, k * self.W:(k + 1) * self.W] = tiles[k % self.res][j][:V, :self.W]
            planes.append(o)
        return planes, pitch


def main():
    args = parse()
    # dmabuf IPC: RCCL across processes needs it on this driver -- also when a launcher other than self_launch() started the
    # ranks (set before torch / HIP are loaded)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)  # does not return
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it with --nproc-per-node %d, or without a "
                 "torch.distributed environment (it then starts its own ranks)" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the engine has no CPU path")
    # BASEVAR_BENCH_BACKEND=gloo + BASEVAR_BENCH_ONE_DEVICE=1 let the multi-rank plumbing be exercised
    # on a single-GPU box (tests/test_bench_multirank.py); the driver's runs use nccl (= RCCL), one GPU per rank.
    backend = os.environ.get("BASEVAR_BENCH_BACKEND", "nccl")
    if os.environ.get("BASEVAR_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # BASEVAR_BENCH_FORCE_DIST=1: run the process group + record gather even with ONE rank, so that the
    # RCCL code path (nccl backend, async gather ordered on the engine's stream) can be exercised on a 1-GPU box
    dist_on = world > 1 or os.environ.get("BASEVAR_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import basevar_amd
    from basevar_amd import _capi as _capi_mod
    from basevar_amd.shard import RecordGatherer

    N = args.samples
    B = args.batch_sites
    tj = None  # --tile-job geometry
    if args.tile_job:
        W = args.tile_width
        n_tiles = max(1, N // W)
        res = max(1, min(n_tiles, args.tile_distinct))
        tj = {"W": W, "n_tiles": n_tiles, "res": res, "N_job": n_tiles * W, "St": min(args.tile_sites, args.batch_sites)}
        B = tj["St"]      # sites per step and rank = sites per tile job
        N = tj["N_job"]   # samples per site as the engine sees them (whole tiles only)
    if args.scaling == "strong":
        if args.batch_sites % world:
            sys.exit("bench.py --scaling strong: --batch-sites must be a multiple of the number of ranks")
        B = args.batch_sites // world  # this rank's contiguous share of the job's batch
    # (--tile-job: the synthetic SOURCE slab the tiles are cut from holds `res` distinct tiles' worth of samples, cycled over the job)
    N_fill = tj["res"] * tj["W"] if tj else N
    pitch = (N_fill + 255) // 256 * 256
    maf = basevar_amd.min_af(N)
    K = max(1, args.chain)          # batches per launch
    Bl = B * K                      # sites per launch (= per step)
    nb = max(1, min(args.distinct_batches, args.steps + args.warmup))
    if tj:
        nb = 1  # one source slab per rank; the job cycles its tiles
    if K > 1:
        # every batch of a chained launch is a different one (no re-reads that the 256 MB infinity cache could serve), as far
        # as 96 GB of HBM go
        fit = max(1, int(96e9 // (5.0 * B * pitch)))
        nb = max(nb, min(K, fit))
    ranks = not args.no_rank_planes
    LAY = _capi_mod.BV_SLAB_RPR_TAGGED if (ranks and args.rank_layout == "tagged") else 0  # bv_slab.layout of every slab / tile

    # ---- preflight: what this run will allocate against what is there -- every rank's HBM, the NUMA node of every GPU, the
    # host DRAM a tile job pins -- printed (rank 0, stderr) BEFORE anything is allocated; a run that cannot fit ends here on
    # every rank with a message and a non-zero exit code instead of an allocator error (or the OOM killer) minutes in
    rec_b = basevar_amd.SITE_DTYPE.itemsize
    depth_pre = 3 if dist_on else max(1, args.streams) * (args.lanes if args.lanes == 1 else 4)
    need_dev = nb * B * pitch * (5 if ranks else 2) + nb * B + depth_pre * Bl * rec_b
    need_dev += Bl * (48 + 2048 + 12 + 16 + 4) if N <= 49152 else Bl * 4          # short-row scratch / variant list
    need_host_pinned = 0
    if tj:
        jp = (N + 255) // 256 * 256
        _, _, tile_bytes = basevar_amd.tile_packed_layout(B, tj["W"], True, False)
        need_dev += B * jp * 5 + jp + 4 * (tile_bytes + 4096)                      # joined planes + the staging ring
        need_dev += tj["res"] * B * ((tj["W"] + 15) // 16 * 16) * 5                # the device-side source tiles
        need_host_pinned = tj["res"] * tile_bytes
    if args.groups:
        need_dev += depth_pre * Bl * min(args.groups, 255) * basevar_amd.GROUP_DTYPE.itemsize + min(Bl * min(args.groups, 32) * 1544, 8 << 30)
    free_dev, total_dev = torch.cuda.mem_get_info(dev)
    host_avail = None
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                host_avail = int(ln.split()[1]) * 1024
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read())
            host_avail = min(host_avail if host_avail is not None else 1 << 62, int(lim) - cur)
    except (OSError, ValueError):
        pass
    numa = _capi_mod.load().bv_device_numa_node(local_rank, None, 0)
    fits = need_dev <= 0.97 * free_dev and (need_host_pinned == 0 or host_avail is None or need_host_pinned <= 0.9 * host_avail / max(1, world))
    pre = [float(need_dev), float(free_dev), float(total_dev), float(need_host_pinned), float(host_avail if host_avail is not None else -1), float(numa),
           1.0 if fits else 0.0]
    if dist_on:
        t_ = torch.tensor(pre, dtype=torch.float64)
        if backend == "nccl":
            t_ = t_.to(dev)
            lst = [torch.zeros_like(t_) for _ in range(world)]
            dist.all_gather(lst, t_)
        else:
            lst = [torch.zeros(len(pre), dtype=torch.float64) for _ in range(world)]
            dist.all_gather(lst, t_)
        pre_all = [[float(x) for x in v.tolist()] for v in lst]
    else:
        pre_all = [pre]
    all_fit = all(r[6] == 1.0 for r in pre_all)
    if rank == 0:
        print("[bench preflight] " + json.dumps({
            "ranks": world, "fits": all_fit,
            "hbm_needed_GB_per_rank": [round(r[0] / 1e9, 4) for r in pre_all], "hbm_free_GB_per_rank": [round(r[1] / 1e9, 2) for r in pre_all],
            "hbm_total_GB_per_rank": [round(r[2] / 1e9, 2) for r in pre_all], "numa_node_of_gpu": [int(r[5]) for r in pre_all],
            "host_pinned_GB_per_rank": [round(r[3] / 1e9, 4) for r in pre_all],
            "host_available_GB": (round(pre_all[0][4] / 1e9, 2) if pre_all[0][4] >= 0 else None)}), file=sys.stderr, flush=True)
    if not all_fit:
        if rank == 0:
            bad = [i for i, r in enumerate(pre_all) if r[6] != 1.0]
            print("[bench] preflight: the run does not fit on rank(s) %s -- HBM needed %.1f GB against %.1f GB free, pinned host memory %.1f GB per rank "
                  "against %s GB available for %d ranks; lower --batch-sites / --distinct-batches / --tile-sites / --tile-distinct" % (
                      bad, pre_all[bad[0]][0] / 1e9, pre_all[bad[0]][1] / 1e9, pre_all[bad[0]][3] / 1e9,
                      ("%.1f" % (pre_all[bad[0]][4] / 1e9)) if pre_all[bad[0]][4] >= 0 else "?", world), file=sys.stderr, flush=True)
        if dist_on:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(3)

    # ---- synthetic batches, resident in HBM.  Global site index = ((batch * world) + rank) * B + row,
    # so every rank owns a contiguous site range of every batch (sites shard embarrassingly).
    batches = []
    for b in range(nb):
        bs = torch.empty((B, pitch), dtype=torch.uint8, device=dev)
        q = torch.empty((B, pitch), dtype=torch.uint8, device=dev)
        mq = torch.empty((B, pitch), dtype=torch.uint8, device=dev) if ranks else None
        rp = torch.empty((B, pitch), dtype=torch.int16, device=dev) if ranks else None
        ref = torch.empty(B, dtype=torch.uint8, device=dev)
        basevar_amd.synth_fill(local_rank, B, N_fill, pitch, bs.data_ptr(), q.data_ptr(), ref.data_ptr(),
                               mq.data_ptr() if ranks else 0, rp.data_ptr() if ranks else 0, seed=args.seed,
                               site_offset=(b * world + rank) * B, coverage=args.coverage, layout=LAY)
        batches.append((bs, q, mq, rp, ref))
    torch.cuda.synchronize()

    ns = max(1, args.streams)
    engs = [basevar_amd.BaseTypeEngine(max_sites=Bl, min_af_value=maf, device=local_rank, max_samples=(N if tj else 0),
                                       flags=(1 if args.tally_only else 0) | args.flags | (0x10000000 if args.lanes == 2 else 0) | (0x20000000 if args.sparse_timing else 0))
            for _ in range(ns)]
    eng = engs[0]
    rec = basevar_amd.SITE_DTYPE.itemsize
    # record buffers: one per stream at N = 1; a ring of 3 per rank at N > 1 so that the gather of
    # batch i (RCCL, asynchronous) overlaps the kernels of batches i+1 and i+2
    depth = 3 if dist_on else ns * (args.lanes if args.lanes == 1 else 4)  # (submits in flight write distinct record buffers)
    gloo_host = dist_on and backend != "nccl"  # gloo has no GPU gather: stage through the host (test plumbing only)
    outs = [torch.zeros(Bl * rec, dtype=torch.uint8, device=dev) for _ in range(depth)]
    houts = [torch.zeros(Bl * rec, dtype=torch.uint8) for _ in range(depth)] if gloo_host else None
    # Everything of engine k -- its kernels and the gather of its records -- is issued on that engine's OWN
    # HIP stream, wrapped for torch.  (Not torch's default stream: its handle is 0, which the C ABI reads as
    # "engine's own stream", so a gather issued from torch would not be ordered behind the kernels.  Not
    # torch's pooled streams either: two of them may share one hardware queue, and --streams 2 then
    # overlaps nothing.)
    streams = [torch.cuda.ExternalStream(e.stream_handle(), device=dev) for e in engs]
    gatherer = RecordGatherer(Bl * rec, torch.device("cpu") if gloo_host else dev, depth=depth) if dist_on else None
    last_slot = [0]

    G = max(0, min(args.groups, 255))
    gid = gouts = None
    if G:
        gen = torch.Generator(device="cpu").manual_seed(args.seed & 0xFFFF)
        g = torch.randint(0, G + 1, (pitch,), generator=gen, dtype=torch.int64)
        gid = torch.where(g == G, torch.full_like(g, 255), g).to(torch.uint8).to(dev)
        gouts = [torch.zeros(Bl * G * basevar_amd.GROUP_DTYPE.itemsize, dtype=torch.uint8, device=dev) for _ in range(depth)]

    rig = None
    if tj:
        # this rank's tiles: pinned host allocations on the GPU's NUMA node, cut from the rank's own source rows
        rig = TileRig(torch, eng, dev, local_rank, B, tj["W"], tj["n_tiles"], tj["res"], batches[0], host=True, layout=LAY)

    def step(i):
        bs, q, mq, rp, ref = batches[i % nb]
        k = i % ns
        slot = i % depth
        out = outs[slot]
        with torch.cuda.stream(streams[k]):
            if gatherer is not None:
                # the gather that last read this buffer must have completed; Work.wait() orders the
                # CURRENT stream behind the collective, so it has to be called on the stream that writes
                gatherer.before_reuse(slot)
            if rig is not None:
                # one tile job: every tile from pinned host memory (the engine's copy streams run ahead of its kernels),
                # joined in HBM, then both passes over the joined rows
                rig.job(out.data_ptr(), _capi_mod.BV_MEM_HOST, stream=streams[k].cuda_stream)
            elif K > 1:
                # K batches, one launch per pass: segment j writes records [j * B, (j + 1) * B) of the step's buffer
                segs = []
                for j in range(K):
                    sb, sq, smq, srp, sref = batches[(i * K + j) % nb]
                    segs.append((B, sb.data_ptr(), sq.data_ptr(), sref.data_ptr(), out.data_ptr() + j * B * rec,
                                 smq.data_ptr() if ranks else 0, srp.data_ptr() if ranks else 0))
                engs[k].submit_many_ptrs(N, pitch, segs, stream=streams[k].cuda_stream, group_id=gid.data_ptr() if G else 0, n_groups=G, layout=LAY,
                                         gouts=[gouts[slot].data_ptr() + j * B * G * basevar_amd.GROUP_DTYPE.itemsize for j in range(K)] if G else None)
            else:
                engs[k].submit_ptrs(B, N, pitch, bs.data_ptr(), q.data_ptr(), ref.data_ptr(), out.data_ptr(),
                                    mq.data_ptr() if ranks else 0, rp.data_ptr() if ranks else 0,
                                    group_id=gid.data_ptr() if G else 0, n_groups=G, gout=gouts[slot].data_ptr() if G else 0,
                                    stream=(0 if os.environ.get("BASEVAR_BENCH_NULLSTREAM") else streams[k].cuda_stream), layout=LAY)
            if gatherer is not None:
                if args.lanes == 2:
                    engs[k].join(streams[k].cuda_stream)  # the lanes run on streams of their own: order the gather behind them
                if gloo_host:
                    houts[slot].copy_(out)
                    gatherer.issue(slot, houts[slot])
                else:
                    gatherer.issue(slot, out)  # ordered records on rank 0: parts in rank order
        last_slot[0] = slot
        return out

    def fence():
        if gatherer is not None:
            with torch.cuda.stream(streams[0]):
                gatherer.drain()
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    last = None
    for i in range(args.warmup):
        last = step(i)
    fence()
    for e in engs:
        e.timing_reset()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    p1_ms = p2_ms = st_ms = 0.0
    nsub = 0
    for e in engs:
        e.wait()
        s_, a1, a2, n_ = e.timing_get_ex()
        st_ms += s_; p1_ms += a1; p2_ms += a2; nsub += n_
    nvar = eng.last_variant_count()
    form = eng.last_launch_form()
    gathered_ok = None
    ranks_verified = 0
    if rank == 0 and dist_on:
        # the gathered buffer holds world x B records in rank order: every record must be a covered site
        recs = torch.cat([p.cpu() for p in gatherer.parts(last_slot[0])]).numpy().view(basevar_amd.SITE_DTYPE)
        mine = outs[last_slot[0]].cpu().numpy().view(basevar_amd.SITE_DTYPE)
        gathered_ok = bool(len(recs) == world * Bl and (recs["total_depth"] > 0).all() and
                           recs[:Bl].tobytes() == mine.tobytes())
        # ... and rank r's part must be what a ONE-rank run of rank r's site range writes: the synthetic rows are stateless in
        # the global site index, so rank 0 regenerates the first V sites of every rank's last batch on its own device, runs them
        # through its engine as ordinary rows and compares the records byte for byte (site ranges, rank order of the gather,
        # and -- with --tile-job -- the tile realisation against the row realisation)
        V = max(0, min(args.verify_sites, B))
        if V:
            i_last = args.warmup + args.steps - 1
            b_last = 0 if tj else ((i_last * K) % nb)
            vout = torch.zeros(V * rec, dtype=torch.uint8, device=dev)
            for r in range(world):
                vb = torch.empty((V, pitch), dtype=torch.uint8, device=dev); vq = torch.empty_like(vb)
                vm = torch.empty_like(vb) if ranks else None
                vr = torch.empty((V, pitch), dtype=torch.int16, device=dev) if ranks else None
                vref = torch.empty(V, dtype=torch.uint8, device=dev)
                basevar_amd.synth_fill(local_rank, V, N_fill, pitch, vb.data_ptr(), vq.data_ptr(), vref.data_ptr(),
                                       vm.data_ptr() if ranks else 0, vr.data_ptr() if ranks else 0, seed=args.seed,
                                       site_offset=(b_last * world + r) * B, coverage=args.coverage, layout=LAY)
                torch.cuda.synchronize()
                if tj:
                    vrig = TileRig(torch, eng, dev, local_rank, V, tj["W"], tj["n_tiles"], tj["res"], (vb, vq, vm, vr, vref), host=False, layout=LAY)
                    (jb, jq, jm, jr), jp = vrig.joined_rows(V)
                    torch.cuda.synchronize()  # (torch built the rows on ITS stream; the engine's own stream does not wait for it)
                    eng.submit_ptrs(V, N, jp, jb.data_ptr(), jq.data_ptr(), vref.data_ptr(), vout.data_ptr(), jm.data_ptr(), jr.data_ptr(), layout=LAY)
                else:
                    eng.submit_ptrs(V, N, pitch, vb.data_ptr(), vq.data_ptr(), vref.data_ptr(), vout.data_ptr(),
                                    vm.data_ptr() if ranks else 0, vr.data_ptr() if ranks else 0,
                                    group_id=gid.data_ptr() if G else 0, n_groups=G,
                                    gout=gouts[0].data_ptr() if G else 0, layout=LAY)
                eng.wait()
                want = vout.cpu().numpy().view(basevar_amd.SITE_DTYPE)
                got = recs[r * Bl:r * Bl + V]
                same = want.tobytes() == got.tobytes()
                ranks_verified += int(same)
                if not same:  # say what differs: a wrong site range, a wrong rank order and a numerical difference look different
                    for f in want.dtype.names:
                        bad = np.nonzero(~np.all(np.atleast_2d((want[f] == got[f]) | ((want[f] != want[f]) & (got[f] != got[f]))).reshape(V, -1), axis=1))[0]
                        if bad.size:
                            print("[bench] rank %d: %d of its first %d gathered records differ from the one-rank re-run in `%s`, first at site %d: "
                                  "gathered %r, re-run %r" % (r, bad.size, V, f, int(bad[0]), got[f][bad[0]].tolist(), want[f][bad[0]].tolist()), file=sys.stderr)
            gathered_ok = gathered_ok and ranks_verified == world

    # per-rank figures for rank 0's line: this rank's wall time of the timed region, its pass-1 fraction of the HBM peak, and
    # what of its step was NOT kernels (the gather's exposed time + launch gaps)
    # nsub: launches that carried timing events (all of them unless --sparse-timing); n_launch: launches of the timed region
    if args.sparse_timing and K > 16:
        raise SystemExit("--sparse-timing: at most 16 batches per step (one launch per step)")
    n_launch = args.steps if args.sparse_timing else nsub
    my_p1_frac = (2.0 * Bl * N * args.steps / max(n_launch, 1)) / max(p1_ms / max(nsub, 1) / 1e3, 1e-12) / 1e9 / HBM_PEAK_GBS
    my_exposed_ms = max(0.0, elapsed / args.steps * 1e3 - (p1_ms + p2_ms) / max(nsub, 1) * n_launch / max(args.steps, 1))
    # --tile-job: bytes this rank pulled over its host link per second of the timed region; the NUMA node of its GPU and the
    # node its pinned tiles were allocated on (-1: the platform does not say / nothing bound)
    my_host_gbps = rig.host_bytes_per_job() * args.steps / elapsed / 1e9 if rig is not None else 0.0
    my_nodes = (float(rig.numa_node), float(rig.bound_node)) if rig is not None else (-1.0, -1.0)

    def all_ranks(vec):
        """every rank's vector of float64 figures, in rank order (a list of lists)"""
        mine_ = torch.tensor(vec, dtype=torch.float64, device=dev)
        if not dist_on:
            return [[float(x) for x in mine_.tolist()]]
        if backend == "nccl":
            lst = [mine_.clone() for _ in range(world)]
            dist.all_gather(lst, mine_)
        else:
            lst = [torch.zeros(len(vec), dtype=torch.float64) for _ in range(world)]
            dist.all_gather(lst, mine_.cpu())
        return [[float(x) for x in t_.tolist()] for t_ in lst]

    per_rank = all_ranks([elapsed, my_p1_frac, my_exposed_ms, my_host_gbps, my_nodes[0], my_nodes[1]])
    elapsed = max(r[0] for r in per_rank)

    packed_leg = None
    if rig is not None and not args.no_packed_tiles:
        # the same job from PACKED host tiles (bv_engine_tiles_add_sparse: the covered cells only), beside the dense job that is
        # `value`: every rank at once; the records must be the dense job's, byte for byte
        rig.build_packed()
        dense_rec = outs[last_slot[0]].clone()
        pout = torch.zeros_like(dense_rec)
        with torch.cuda.stream(streams[0]):
            rig.job_packed(pout.data_ptr(), stream=streams[0].cuda_stream)
        eng.wait(); torch.cuda.synchronize()
        same_rec = bool(torch.equal(pout, dense_rec))
        if dist_on:
            dist.barrier()
        n_pj = max(2, args.steps // 2)
        t0p = time.perf_counter()
        for _ in range(n_pj):
            with torch.cuda.stream(streams[0]):
                rig.job_packed(pout.data_ptr(), stream=streams[0].cuda_stream)
        eng.wait(); torch.cuda.synchronize()
        dtp = time.perf_counter() - t0p
        legs = all_ranks([dtp, 1.0 if same_rec else 0.0, float(rig.packed_host_bytes_per_job())])
        tmax = max(r[0] for r in legs)
        packed_leg = {"sites_per_s": world * B * n_pj / tmax, "jobs": n_pj, "ms_per_job": tmax / n_pj * 1e3,
                      "host_bytes_per_job_per_rank": int(legs[0][2]), "host_GBps_per_rank": [r[2] * n_pj / r[0] / 1e9 for r in legs],
                      "records_identical_to_dense_tiles": all(r[1] == 1.0 for r in legs),
                      "bytes_per_cell": legs[0][2] / (float(B) * N),
                      "how": "bv_engine_tiles_add_sparse: per site the run of its covered cells, 7 B each (sample u16, call, phred, mapq, rank u16) + 4 B per site and tile"}

    tile_legs = None
    if args.with_tile_mode and not tj:
        # BASELINE configs[4]'s shape beside the row workload, on EVERY rank at once (the ranks share the host's memory
        # channels and PCIe root complexes, so the per-rank link rates are only meaningful measured together): one tile =
        # one batchfile's worth of samples for every site of the rank's site range; never `value`
        St = min(B, args.tile_sites)
        Wt = args.tile_width
        n_tiles_t = max(1, N // Wt)
        trig = TileRig(torch, eng, dev, local_rank, St, Wt, n_tiles_t, min(n_tiles_t, args.tile_distinct), batches[0], host=True, layout=LAY)
        tout = torch.zeros(St * rec, dtype=torch.uint8, device=dev)
        tms = []
        for kind, many in ((_capi_mod.BV_MEM_DEVICE, True), (_capi_mod.BV_MEM_DEVICE, False), (_capi_mod.BV_MEM_HOST, False)):
            trig.job(tout.data_ptr(), kind, many); eng.wait()   # warm-up: scratch, staging ring
            if dist_on:
                dist.barrier()
            t0 = time.perf_counter()
            trig.job(tout.data_ptr(), kind, many); eng.wait()
            tms.append(time.perf_counter() - t0)
        legs = all_ranks(tms + [float(trig.numa_node), float(trig.bound_node)])
        cells = float(St) * n_tiles_t * Wt
        if rank == 0:
            def leg(j, bytes_per_rank):
                tmax = max(r[j] for r in legs)
                return {"value": world * St / tmax, "unit": "sites/s", "GBps": world * bytes_per_rank / tmax / 1e9,
                        "per_rank_GBps": [bytes_per_rank / r[j] / 1e9 for r in legs]}
            tile_legs = {"sites_per_rank": St, "samples": n_tiles_t * Wt, "tile_width": Wt, "tiles": n_tiles_t, "ranks": world,
                         "device_resident": dict(leg(0, 5 * cells), how="bv_engine_tiles_add_many, one launch per 256 tiles"),
                         "device_resident_tile_by_tile": leg(1, 5 * cells),
                         "host_pinned_pcie": dict(leg(2, float(trig.host_bytes_per_job())),
                                                  numa_node_of_gpu=[int(r[3]) for r in legs], tiles_bound_to_node=[int(r[4]) for r in legs])}
        del trig, tout

    if rank == 0:
        sites_per_s = world * Bl * args.steps / elapsed
        p1_avg_s = p1_ms / max(nsub, 1) / 1e3
        p2_avg_s = p2_ms / max(nsub, 1) / 1e3
        st_avg_s = st_ms / max(nsub, 1) / 1e3
        # pass 1: u8 call + u8 phred per cell of every batch of a launch (a chain longer than the engine's queue is
        # split into several launches: bytes of the timed region / its launches)
        algo_bytes = 2.0 * Bl * N * args.steps / max(n_launch, 1)
        # the dominant (HBM-bound) kernel: on short rows pass 1 is a streaming kernel + a solve kernel that reads no
        # planes; on long rows it is one kernel (st_avg_s == p1_avg_s)
        # which kernels the launches took: asked of the engine (bv_engine_last_launch_form), not re-derived here
        two_kernel = bool(form & _capi_mod.BV_FORM_SHORT_ROWS)
        fused = two_kernel and bool(form & _capi_mod.BV_FORM_ONE_KERNEL)
        fused_p2 = bool(form & _capi_mod.BV_FORM_PASS2_FUSED)
        kernel_name = "bv_p1s_fused_kernel" if fused else "bv_p1s_stream_kernel" if two_kernel else "bv_pass1_kernel"
        # variant fraction of the last launch (a chained launch counts its whole queue)
        fvar = nvar / max(1.0, Bl * args.steps / max(n_launch, 1))
        # The fused short-row kernel that also streams the variant sites' pass-2 rows: its algorithmic bytes are section 8d's
        # S*N*(2 + 3 f_var) (its traffic 2 + 4 f_var: the call byte of a variant row is read twice).
        kernel_bytes = algo_bytes * ((1.0 + 1.5 * fvar) if fused_p2 else 1.0)
        achieved = kernel_bytes / st_avg_s / 1e9
        # HBM bytes per launch of that kernel from the PMC counters: NOT measured in this run (counter collection needs
        # rocprofv3 around the process) but read from the committed record of the SAME kernel and configuration, if one
        # exists (tools/collect_profiles.sh + tools/summarize_profiles.py); `traffic_source` says which
        var = ("<p2rows>" if fused_p2 else "<p1only>") if fused else ""
        traffic, traffic_source = traffic_record(kernel_name, var, Bl, N, args.rank_layout, K)
        line = {
            "metric": "genomic sites/sec through basetype caller at N samples",
            "value": sites_per_s, "unit": "sites/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[4] shape: synthetic NIPT pileup, %d samples/site delivered as %d tiles of %d samples "
                             "(the reference's --batch-count batchfiles) from pinned host DRAM on the GPU's NUMA node, coverage %.2f, "
                             "tile jobs of %d sites per GPU, joined in HBM, both passes; weak scaling: every rank owns its "
                             "contiguous site range of the job (1M-site job = %d such jobs per GPU at %d GPUs)" % (
                                 N, tj["n_tiles"], tj["W"], args.coverage, B, (1000000 + world * B - 1) // (world * B), world)) if tj else
                            "%s: synthetic NIPT pileup, %d samples/site, coverage %.2f, "
                            "HBM-resident batches of %d sites per GPU%s (%s; 1M-site job = %d such batches)" % (
                                "BASELINE configs[2]" if N == 100000 else "BASELINE configs[1]" if N == 10000 else
                                "BASELINE configs[4] shape" if N == 1000000 else "diagnostic shape",
                                N, args.coverage, B, " per step" if K == 1 else ", %d batches per step chained into one launch per pass" % K,
                                "strong scaling: the job's %d-site batch split over the ranks" % (world * B)
                                if args.scaling == "strong" else "weak scaling: per-GPU batch fixed",
                                (1000000 + world * B - 1) // (world * B) if args.scaling == "strong" else (1000000 + B - 1) // B),
                "samples": N, "batch_sites": B, "engine_lanes": args.lanes, "chain": K, "sites_per_launch": Bl, "coverage": args.coverage, "planes": "call u8,phred u8" + ((",mapq u8,rpr u16 " + ("(BV_SLAB_RPR_TAGGED: rank | base << 13 | nocall << 15, written so by the producer; pass 2 reads mapq + rpr = 3 B/cell)"
                                                                                     if LAY else "(plain ranks; pass 2 re-reads the call plane: 4 B/cell)")) if ranks else ""),
                "rank_layout": (args.rank_layout if ranks else None),
                "parallelism": "site-sharded x%d, gather of %d-byte records to rank 0" % (world, rec),
                "job_batch_sites": world * Bl, "backend": (backend if dist_on else None),
                "rccl_ranks": (dist.get_world_size() if dist_on and backend == "nccl" else 0),
                "dist_world_size": (dist.get_world_size() if dist_on else 1),
                "variant_sites_last_batch": nvar, "gathered_records_ok": gathered_ok,
                # N > 1: ranks whose first `verify_sites` gathered records equal rank 0's own re-run of that rank's site range
                "ranks_verified": (ranks_verified if dist_on else None), "verify_sites": (max(0, min(args.verify_sites, B)) if dist_on else 0),
                "tile_job": ({"tiles_per_job": tj["n_tiles"], "tile_width": tj["W"], "distinct_host_tiles": tj["res"],
                              "host_bytes_per_job_per_rank": rig.host_bytes_per_job(),
                              "packed": packed_leg, "packed_over_dense": (packed_leg["sites_per_s"] / sites_per_s if packed_leg else None),
                              # what every rank pulled over ITS host link per second of the timed region, and all of them together
                              "host_pinned_pcie_GBps_per_rank": [r[3] for r in per_rank],
                              "host_pinned_pcie_GBps_total": sum(r[3] for r in per_rank),
                              "numa_node_of_gpu": [int(r[4]) for r in per_rank], "tiles_bound_to_node": [int(r[5]) for r in per_rank]}
                             if tj else None),
                # 1: sites of <= 64 covered samples are replayed with the host libm's own log() (ties decided as the reference
                # decides them); 0: the device library's log() (values within 1e-6, exact ties undecided, BV_SITE_LOG_APPROX)
                "host_log_exact": int(eng.host_log_exact),
                "per_rank": {"step_ms": [r[0] / args.steps * 1e3 for r in per_rank], "pass1_frac": [r[1] for r in per_rank],
                             "pass1_frac_min": min(r[1] for r in per_rank), "pass1_frac_max": max(r[1] for r in per_rank),
                             "step_ms_min": min(r[0] for r in per_rank) / args.steps * 1e3,
                             "step_ms_max": max(r[0] for r in per_rank) / args.steps * 1e3,
                             # step time minus this rank's kernel time: launch gaps + what of the record gather is not hidden
                             "exposed_ms_per_step_max": max(r[2] for r in per_rank)},
            },
            "roofline": {
                # achieved / frac / avg_launch_ms describe the DOMINANT kernel named in `kernel` (long rows: all of pass 1;
                # short rows: the streaming kernel of pass 1 -- the solve kernels read no planes); pass1_frac is all of pass 1
                # over the same bytes and whole_path_frac both passes over section 8d's bytes: compare those across row lengths
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": kernel_name,
                "frac_covers": ("pass 1 and the variant sites' pass-2 rows (one persistent kernel)" if fused_p2 else
                                "pass 1 (one persistent kernel: streaming and solver waves)" if fused else
                                "streaming kernel of pass 1" if two_kernel else "pass 1"),
                "algorithmic_bytes_per_launch": kernel_bytes,
                "avg_launch_ms": st_avg_s * 1e3, "pass1_avg_ms": p1_avg_s * 1e3,
                "pass1_frac": algo_bytes / p1_avg_s / 1e9 / HBM_PEAK_GBS,  # all of pass 1 (streaming + solve kernels) over the same bytes
                "pass2_avg_launch_ms": p2_avg_s * 1e3, "launches": n_launch, "launches_timed": nsub,
                # BASELINE.md section 3's whole-path figure: S*N*(2 + 3 f_var) bytes over both kernels' time
                # (pass 2 also re-reads the call byte of variant rows: its own traffic is 4 B/cell)
                "whole_path_GBps": (1.0 + 1.5 * fvar) * algo_bytes / max(p1_avg_s + p2_avg_s, 1e-12) / 1e9,
                "whole_path_frac": (1.0 + 1.5 * fvar) * algo_bytes / max(p1_avg_s + p2_avg_s, 1e-12) / 1e9 / HBM_PEAK_GBS,
                # the same bytes over the WALL time of a step on this rank (launch gaps and, with --lanes 2 / --streams 2, the
                # overlap of consecutive submits included: the per-submit kernel times above then count co-running kernels twice)
                "whole_path_frac_wall": (1.0 + 1.5 * fvar) * algo_bytes * max(n_launch, 1) / args.steps / max(per_rank[0][0] / args.steps, 1e-12) / 1e9 / HBM_PEAK_GBS,
            },
        }
        if world == 1:
            # measured device-copy ceiling on this box (read + write bytes), for context next to the spec peak
            src = torch.empty(1 << 30, dtype=torch.uint8, device=dev); dst = torch.empty_like(src)
            dst.copy_(src); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                dst.copy_(src)
            torch.cuda.synchronize()
            line["roofline"]["measured_copy_GBps"] = 5 * 2 * src.numel() / (time.perf_counter() - t0) / 1e9
            del src, dst
        if world == 1 and args.with_host_path:
            # PCIe-inclusive rate when the boundary hands over HOST buffers (DESIGN.md note; not `value`)
            hb = min(B, 4096)
            bs, q, mq, rp, ref = batches[0]
            host = [t[:hb].cpu().pin_memory() if t is not None else None for t in (bs, q, mq, rp)]
            href = ref[:hb].cpu().pin_memory()
            hout = torch.zeros(hb * rec, dtype=torch.uint8).pin_memory()
            from basevar_amd import _capi
            def hstep():
                eng.submit_ptrs(hb, N, pitch, host[0].data_ptr(), host[1].data_ptr(), href.data_ptr(), hout.data_ptr(),
                                host[2].data_ptr() if ranks else 0, host[3].data_ptr() if ranks else 0,
                                mem_kind=_capi.BV_MEM_HOST, layout=LAY)
                eng.wait()
            hstep()
            t0 = time.perf_counter()
            for _ in range(3):
                hstep()
            dt = (time.perf_counter() - t0) / 3
            line["pcie_inclusive"] = {"value": hb / dt, "unit": "sites/s", "batch_sites": hb,
                                      "host_GBps": hb * pitch * (5 if ranks else 2) / dt / 1e9}
        if tile_legs is not None:
            line["tile_mode"] = tile_legs
        if (world == 1 and not args.no_configs1 and not tj and N == 100000 and K == 1 and ranks and not G and not args.flags and not args.tally_only
                and args.lanes == 1 and ns == 1):
            # the default run only: BASELINE configs[1] on the same box, behind the headline's timed region (never `value`)
            try:
                line["configs1"] = configs1_leg(torch, basevar_amd, _capi_mod, dev, local_rank, args, LAY)
            except Exception as ex:
                line["configs1"] = None
                print("[bench] configs1 leg failed: %r" % (ex,), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            try:
                cb, cpu_rec, cpu_idx = cpu_baseline(torch, batches[0], N, maf, args.cpu_sites, rank_mask=0x1FFF if LAY else 0xFFFF)
                line["cpu_baseline"] = cb
                # GPU records of the same sites (one more submit of batch 0)
                bs0, q0, mq0, rp0, ref0 = batches[0]
                eng.submit_ptrs(B, N, pitch, bs0.data_ptr(), q0.data_ptr(), ref0.data_ptr(), outs[0].data_ptr(),
                                mq0.data_ptr() if ranks else 0, rp0.data_ptr() if ranks else 0,
                                stream=streams[0].cuda_stream, layout=LAY)
                eng.wait()
                gpu_rec = outs[0].cpu().numpy().view(basevar_amd.SITE_DTYPE)[cpu_idx]
                line["parity_sample"] = parity_on_sample(gpu_rec, cpu_rec)
                line["gpu_over_cpu_allcore"] = sites_per_s / cb["value"]
                line["gpu_over_cpu_1thread"] = sites_per_s / cb["single_thread_value"]
            except Exception as ex:
                line["cpu_baseline"] = None
                print("[bench] cpu_baseline failed: %r" % (ex,), file=sys.stderr)
        print(json.dumps(line), flush=True)
    for e in engs:
        e.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
