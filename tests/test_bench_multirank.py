"""bench.py with 2 ranks, started both ways: as the driver does for N > 1 (torch.distributed.run, one
process per rank) and self-launched (`python bench.py --gpus 2` without a torch.distributed environment
starts its own ranks as child processes before touching the GPU).  On the single-GPU box both ranks share cuda:0 and use the gloo backend
(RCCL refuses two ranks on one device), which still exercises rank/env handling, per-rank site
ranges, the ordered gather of records to rank 0, max-over-ranks timing and the JSON contract."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_gloo_one_device():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, BASEVAR_BENCH_BACKEND="gloo", BASEVAR_BENCH_ONE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--samples", "20000", "--batch-sites", "2048", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["unit"] == "sites/s"
    assert d["config"]["gathered_records_ok"] is True
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and "cpu_baseline" not in d


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_self_launch_two_ranks(scaling):
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent starts the two ranks itself."""
    env = dict(os.environ, BASEVAR_BENCH_BACKEND="gloo", BASEVAR_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--samples",
           "20000", "--batch-sites", "2048", "--no-cpu-baseline", "--scaling", scaling]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["config"]["gathered_records_ok"] is True
    assert d["config"]["dist_world_size"] == 2 and d["config"]["backend"] == "gloo"
    assert d["config"]["batch_sites"] == (1024 if scaling == "strong" else 2048)
    assert d["config"]["job_batch_sites"] == (2048 if scaling == "strong" else 4096)


def _run_bench(extra, world, env_extra=None, timeout=900):
    """`python bench.py --gpus <world> ...` self-launched, all ranks on cuda:0 over gloo; returns the parsed line"""
    env = dict(os.environ, BASEVAR_BENCH_BACKEND="gloo", BASEVAR_BENCH_ONE_DEVICE="1")
    env.update(env_extra or {})
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--no-cpu-baseline"] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    d["_stderr"] = "\n".join(l for l in p.stderr.splitlines() if l.startswith("[bench"))  # what a failed verification says; the preflight line
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("samples", [6000, 60000], ids=["short_rows", "long_rows"])
def test_bench_eight_ranks_one_device_every_rank_verified(samples):
    """BASELINE configs[3] in small, at the driver's largest rank count: 8 ranks (one device, gloo), every rank its
    contiguous site range of every batch; rank 0 re-runs the first sites of EVERY rank's last batch as a one-rank job
    and the gathered records must equal them byte for byte, in rank order (reference analogue: the fan-out and ordered
    merge of src/basetype_caller.cpp:469-525)."""
    d = _run_bench(["--steps", "3", "--warmup", "1", "--samples", str(samples), "--batch-sites", "512", "--verify-sites", "128"], 8)
    c = d["config"]
    assert d["n_gpus"] == 8 and c["dist_world_size"] == 8 and c["job_batch_sites"] == 8 * 512
    assert c["gathered_records_ok"] is True and c["ranks_verified"] == 8 and c["verify_sites"] == 128, d["_stderr"]
    assert len(c["per_rank"]["step_ms"]) == 8 and all(t > 0 for t in c["per_rank"]["step_ms"])
    assert len(c["per_rank"]["pass1_frac"]) == 8
    assert d["ms_per_step"] == pytest.approx(max(c["per_rank"]["step_ms"]), rel=1e-9)  # the slowest rank's time


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2, 8])
def test_bench_tile_job_ranks_one_device(world):
    """BASELINE configs[4] in small: `--tile-job` makes the tile job THE timed workload -- every rank streams its own
    site range as tiles of --tile-width samples from pinned host memory (src/basetype_caller.cpp:419-453, 589-601), the
    records are gathered, and rank 0 checks every rank's part against a ROW submit of the same cells (the tile
    realisation against the row realisation, byte for byte)."""
    extra = ["--tile-job", "--steps", "2", "--warmup", "1", "--samples", "5000", "--tile-width", "200", "--tile-sites", "192",
             "--tile-distinct", "7", "--verify-sites", "64"]
    env = {"BASEVAR_BENCH_FORCE_DIST": "1"} if world == 1 else None
    d = _run_bench(extra, world, env)
    c = d["config"]
    assert d["n_gpus"] == world and d["steps"] == 2 and d["unit"] == "sites/s" and d["value"] > 0
    assert c["workload"].startswith("BASELINE configs[4] shape") and c["samples"] == 5000 and c["batch_sites"] == 192
    tjob = c["tile_job"]
    assert tjob["tiles_per_job"] == 25 and tjob["tile_width"] == 200 and tjob["distinct_host_tiles"] == 7
    assert len(tjob["host_pinned_pcie_GBps_per_rank"]) == world and all(g > 0 for g in tjob["host_pinned_pcie_GBps_per_rank"])
    assert tjob["host_pinned_pcie_GBps_total"] == pytest.approx(sum(tjob["host_pinned_pcie_GBps_per_rank"]))
    assert len(tjob["numa_node_of_gpu"]) == world and len(tjob["tiles_bound_to_node"]) == world
    for node, bound in zip(tjob["numa_node_of_gpu"], tjob["tiles_bound_to_node"]):
        assert bound in (node, -1)  # bound to the GPU's node where the platform names it and the rank may run there
    assert c["gathered_records_ok"] is True and c["ranks_verified"] == world, d["_stderr"]
    assert d["roofline"]["launches"] == 2 and 0 < d["roofline"]["frac"] < 1


@pytest.mark.gpu
def test_bench_with_tile_mode_on_two_ranks():
    """`--with-tile-mode` beside the row workload on N ranks: the three tile legs run on every rank at once and the line
    carries the per-rank link rates and their sum (never `value`)."""
    d = _run_bench(["--steps", "2", "--warmup", "1", "--samples", "20000", "--batch-sites", "1024", "--with-tile-mode",
                    "--tile-sites", "256", "--tile-width", "200", "--tile-distinct", "8"], 2)
    t = d["tile_mode"]
    assert t["ranks"] == 2 and t["sites_per_rank"] == 256 and t["tiles"] == 100
    for leg in ("device_resident", "device_resident_tile_by_tile", "host_pinned_pcie"):
        assert len(t[leg]["per_rank_GBps"]) == 2 and t[leg]["value"] > 0 and t[leg]["GBps"] > 0
    assert d["config"]["workload"].startswith("diagnostic shape") and d["config"]["ranks_verified"] == 2


def test_bench_self_launch_fails_in_the_children_without_a_gpu():
    """On a GPU-less box `python bench.py --gpus 2` must get as far as starting its ranks: the failure is the
    ranks' "needs a GPU" check, relayed with a non-zero exit code -- not an argument check in the parent."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by test_bench_self_launch_two_ranks")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert p.returncode != 0
    # the ranks got to the GPU check (torchrun ends the other rank as soon as one has failed: one or both messages)
    assert p.stderr.count("bench.py needs a GPU") >= 1, p.stderr[-2000:]
    assert "2-rank launch failed" in p.stderr and "must be launched with" not in p.stderr


@pytest.mark.gpu
def test_bench_rccl_backend_single_rank():
    """The backend the driver's N > 1 runs use ("nccl" = RCCL), forced on with ONE rank: process-group
    init bound to the device, the asynchronous gather issued on the engine's own HIP stream (wrapped as a
    torch ExternalStream), Work.wait() ordering before buffer reuse, barrier and all_reduce of the timing."""
    env = dict(os.environ, BASEVAR_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BASEVAR_BENCH_BACKEND", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "7", "--warmup", "2", "--samples",
           "20000", "--batch-sites", "2048", "--no-cpu-baseline", "--streams", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["gathered_records_ok"] is True and d["value"] > 0
    assert d["config"]["backend"] == "nccl" and d["config"]["rccl_ranks"] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("samples", [60000, 6000], ids=["long_rows", "short_rows"])
def test_bench_chained_batches(samples):
    """`--chain K`: K batches per step as one chained launch (bv_engine_submit_many); the line keeps the contract and says so."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--samples", str(samples),
           "--batch-sites", "1024", "--chain", "4", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["config"]["chain"] == 4 and d["config"]["batch_sites"] == 1024 and d["config"]["sites_per_launch"] == 4096
    assert d["roofline"]["launches"] == 3 and d["value"] > 0 and 0 < d["roofline"]["frac"] < 1
    if samples > 49152:
        assert d["roofline"]["algorithmic_bytes_per_launch"] == 2.0 * 4096 * samples  # pass 1: 2 B per cell
    else:
        # short rows: ONE kernel streams pass 1 and the variant sites' rank-sum rows: 2 B per cell + 3 B per cell of variant rows
        assert d["roofline"]["kernel"] == "bv_p1s_fused_kernel"
        nvar = d["config"]["variant_sites_last_batch"]
        assert 0 < nvar <= 4096
        assert d["roofline"]["algorithmic_bytes_per_launch"] == pytest.approx((2.0 * 4096 + 3.0 * nvar) * samples, rel=1e-9)


def _n_devices():
    import torch
    return torch.cuda.device_count()  # (counting devices does not initialise the GPU in this process)


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_rccl_two_devices(scaling):
    """The driver's N = 2 run in small: `bench.py --gpus 2` with the RCCL backend, one rank per GPU, the ordered gather of
    records over xGMI (reference analogue: the fan-out and file merge of src/basetype_caller.cpp:469-525).  Skipped on a
    one-GPU box; the first multi-GPU lease produces evidence instead of a first run."""
    if _n_devices() < 2:
        pytest.skip("needs two GPUs (this box has %d)" % _n_devices())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BASEVAR_BENCH_BACKEND", "BASEVAR_BENCH_ONE_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--samples", "20000",
           "--batch-sites", "4096", "--no-cpu-baseline", "--scaling", scaling]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
    assert d["config"]["backend"] == "nccl" and d["config"]["rccl_ranks"] == 2 and d["config"]["dist_world_size"] == 2
    assert d["config"]["gathered_records_ok"] is True
    assert d["config"]["job_batch_sites"] == (4096 if scaling == "strong" else 8192)


@pytest.mark.gpu
def test_bench_preflight_at_eight_ranks_and_a_job_that_does_not_fit():
    """Before anything is allocated every rank reports what it will allocate against its free HBM, its GPU's NUMA node and the
    host memory a tile job pins; rank 0 prints the table.  A job that cannot fit ends on EVERY rank with exit code 3 and a
    message -- not with an allocator error minutes into the run."""
    d = _run_bench(["--steps", "2", "--warmup", "1", "--tile-job", "--tile-sites", "64", "--samples", "20000", "--tile-width", "200", "--tile-distinct", "8",
                    "--verify-sites", "32"], 8)
    pre = [l for l in d["_stderr"].splitlines() if l.startswith("[bench preflight] ")]
    assert len(pre) == 1, d["_stderr"]
    t = json.loads(pre[0][len("[bench preflight] "):])
    assert t["ranks"] == 8 and t["fits"] is True and len(t["hbm_free_GB_per_rank"]) == 8 and len(t["numa_node_of_gpu"]) == 8
    assert all(x > 0 for x in t["host_pinned_GB_per_rank"]) and all(a < b for a, b in zip(t["hbm_needed_GB_per_rank"], t["hbm_free_GB_per_rank"]))
    assert d["config"]["ranks_verified"] == 8
    # ... and one that cannot: 4 M sites x 100 k samples per rank = 2 TB of planes
    env = dict(os.environ, BASEVAR_BENCH_BACKEND="gloo", BASEVAR_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--batch-sites", "4000000", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert p.returncode != 0 and "preflight: the run does not fit" in p.stderr, p.stderr[-2000:]
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
