"""Site-range sharding over the GPUs of one node and the ordered gather of result records.

Sites are independent units (the reference builds a fresh BatchInfo/BaseType per site,
src/basetype_caller.cpp:675, 742), so the path shards embarrassingly: rank r owns the
contiguous range [r*S/G, (r+1)*S/G), mirroring the reference's contiguous 100 kb sub-region
tasks (src/basetype_caller.cpp:474-510).  There is NO data-path collective.  The only
exchange is the end-of-batch gather of fixed-size records to rank 0, whose rank-order
concatenation is genomic order -- it replaces merge_file_by_line (caller.cpp:521-522).
With backend "nccl" (= RCCL on ROCm) it runs over xGMI; ~208 B/site, so it is latency- not
bandwidth-relevant.  The same code runs on CPU tensors with backend "gloo" (tests).
"""
import torch
import torch.distributed as dist


def site_range(rank, world, n_sites):
    """Contiguous site range [lo, hi) of `rank` out of `world` ranks."""
    lo = n_sites * rank // world
    hi = n_sites * (rank + 1) // world
    return lo, hi


def gather_records(local, dst=0, group=None):
    """Gather per-rank record buffers (1-D uint8 tensors, possibly of different lengths) to
    `dst` and return their rank-ordered concatenation there (None elsewhere)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    return gather_records_sized(local, sizes, dst=dst, group=group)


def gather_records_sized(local, sizes, dst=0, group=None):
    """As gather_records() when every rank already knows all sizes (no size exchange: one
    collective per batch).  Equal sizes use a single gather; ragged ones are padded."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mx = max(sizes)
    buf = local
    if local.numel() != mx:
        buf = torch.zeros(mx, dtype=local.dtype, device=local.device)
        buf[: local.numel()] = local
    if rank == dst:
        parts = [torch.empty(mx, dtype=local.dtype, device=local.device) for _ in range(world)]
        dist.gather(buf, gather_list=parts, dst=dst, group=group)
        return torch.cat([p[:s] for p, s in zip(parts, sizes)])
    dist.gather(buf, gather_list=None, dst=dst, group=group)
    return None


class RecordGatherer:
    """Per-batch gather of equal-sized record buffers to rank `dst`, kept off the critical path:
    the collective of batch i is issued asynchronously and only waited for when its buffers are
    about to be reused (ring of `depth` slots), so it overlaps the kernels of batches i+1, i+2.
    On `dst`, `parts(slot)` holds the ranks' buffers in rank order == genomic order."""

    def __init__(self, nbytes, device, depth=3, dst=0, group=None):
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dst, self.group, self.depth = dst, group, depth
        self.pending = [None] * depth
        self._parts = None
        if self.rank == dst:
            self._parts = [[torch.empty(nbytes, dtype=torch.uint8, device=device) for _ in range(self.world)]
                           for _ in range(depth)]

    def before_reuse(self, slot):
        """Call before overwriting the buffer that was passed to issue(slot, ...) last time."""
        w = self.pending[slot]
        if w is not None:
            w.wait()
            self.pending[slot] = None

    def issue(self, slot, local):
        parts = self._parts[slot] if self.rank == self.dst else None
        self.pending[slot] = dist.gather(local, gather_list=parts, dst=self.dst, group=self.group, async_op=True)

    def drain(self):
        for k in range(self.depth):
            self.before_reuse(k)

    def parts(self, slot):
        return self._parts[slot] if self._parts is not None else None
