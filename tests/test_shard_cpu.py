"""CPU tests of the multi-GPU layer: world_size-2 gloo processes shard a site range and
gather fixed-size records to rank 0 in genomic order (the only exchange on the path)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from basevar_amd import _capi  # noqa: E402
from basevar_amd.shard import RecordGatherer, gather_records, gather_records_sized, site_range  # noqa: E402


def test_site_ranges_partition_exactly():
    for S in (1, 7, 8192, 1000003):
        for G in (1, 2, 3, 8):
            r = [site_range(k, G, S) for k in range(G)]
            assert r[0][0] == 0 and r[-1][1] == S
            assert all(r[k][1] == r[k + 1][0] for k in range(G - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_sites, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = site_range(rank, world, n_sites)
    rec = np.zeros(hi - lo, dtype=_capi.SITE_DTYPE)
    rec["total_depth"] = np.arange(lo, hi)          # stands for the genomic position
    rec["qual"] = np.arange(lo, hi) * 0.5
    local = torch.from_numpy(rec.view(np.uint8).copy())
    sizes = [(site_range(k, world, n_sites)[1] - site_range(k, world, n_sites)[0]) * _capi.SITE_DTYPE.itemsize
             for k in range(world)]
    a = gather_records(local, dst=0)
    b = gather_records_sized(local, sizes, dst=0)
    # asynchronous ring gatherer (what bench.py uses): equal-sized buffers, several batches in flight
    nb = 64 * _capi.SITE_DTYPE.itemsize
    g = RecordGatherer(nb, torch.device("cpu"), depth=3)
    bufs = [torch.zeros(nb, dtype=torch.uint8) for _ in range(3)]
    ring_ok = True
    for i in range(7):
        slot = i % 3
        g.before_reuse(slot)
        bufs[slot].fill_((17 * i + rank) % 251)
        g.issue(slot, bufs[slot])
    g.drain()
    if rank == 0:
        for i in (4, 5, 6):  # the last use of every slot
            parts = g.parts(i % 3)
            ring_ok &= all(bool((parts[r] == (17 * i + r) % 251).all()) for r in range(world))
    if rank == 0:
        assert ring_ok
        q.put((a.numpy().tobytes(), b.numpy().tobytes()))
    else:
        assert a is None and b is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_sites", [64, 101])
def test_gather_is_rank_ordered_gloo(n_sites):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_sites, q)) for r in range(world)]
    for p in procs:
        p.start()
    a, b = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for blob in (a, b):
        rec = np.frombuffer(blob, dtype=_capi.SITE_DTYPE)
        assert len(rec) == n_sites
        assert np.array_equal(rec["total_depth"], np.arange(n_sites))   # genomic order restored
        assert np.array_equal(rec["qual"], np.arange(n_sites) * 0.5)
