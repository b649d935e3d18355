"""Batch sharding over the GPUs of a node (one process per GPU, torch.distributed).

Problems are independent (``update_u`` shares nothing between problems), so a batch is
cut into contiguous slices, one per rank, each rank solves its slice with no
communication, and the only exchange of the path is the final gather of costs,
iteration counts and trajectories on one rank (RCCL over xGMI on GPUs; the same code
runs on the ``gloo`` backend for CPU tests).  The reference has no counterpart: it
solves one problem in one Python process.
"""
import numpy as np


def shard_bounds(total, world_size, rank):
    """Contiguous slice [lo, hi) of ``total`` problems owned by ``rank``: the first
    ``total % world_size`` ranks get one extra problem."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank %d of %d" % (rank, world_size))
    base, extra = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(total, world_size):
    return [shard_bounds(total, world_size, r)[1] - shard_bounds(total, world_size, r)[0]
            for r in range(world_size)]


def gather_to_root(local, sizes, dst=0, group=None):
    """Gather per-rank tensors ``local`` [n_r, ...] (n_r = sizes[r]) on rank ``dst`` and
    return the concatenation [sum n_r, ...] there, None elsewhere.  Slices of unequal
    size are padded to the largest for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(sizes) == world and local.shape[0] == sizes[rank]
    pad = max(sizes)
    if local.shape[0] != pad:
        buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf[:local.shape[0]] = local
    else:
        buf = local.contiguous()
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, parts, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


def solve_sharded(local_solve, x0, us_init, dst=0, group=None, keys=("xs", "us", "J", "iters", "status")):
    """Solve the rank's slice of a global batch and gather the results on ``dst``.

    x0 [B, n], us_init [B, N, m]: the GLOBAL batch (every rank passes the same arrays,
    numpy or torch on any device; only the own slice is touched).
    local_solve(x0_slice, us_slice) -> dict of tensors with leading dim = slice size
    (e.g. a closure over ilqr_amd._hip.solve).
    Returns (local_result, gathered): ``gathered`` is a dict of [B, ...] tensors on
    rank ``dst`` and None elsewhere.
    """
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    B = x0.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    res = local_solve(x0[lo:hi], us_init[lo:hi])
    sizes = shard_sizes(B, world)
    gathered = {}
    for key in keys:
        gathered[key] = gather_to_root(res[key], sizes, dst=dst, group=group)
    return res, (gathered if rank == dst else None)


class GatherPipeline(object):
    """Gathers of successive solves that overlap the next solve.

    A batch job that solves one batch after another does not have to wait for the results of
    batch i to reach the root before it starts batch i+1: the collective runs on the backend's
    own stream (RCCL) or thread (gloo) while the device computes.  The caller alternates between
    ``depth`` sets of result tensors; ``submit`` starts the gather of one set, ``reclaim`` must be
    called before a set is overwritten, ``drain`` before the results are read or a timed region
    ends.  Slices must have equal sizes (the weak-scaling layout: every rank owns as many
    problems); the root receives into preallocated [world, n, ...] tensors, one per set and key.
    """

    def __init__(self, depth=2, dst=0, group=None):
        import torch.distributed as dist
        self.depth, self.dst, self.group = depth, dst, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.pending = [[] for _ in range(depth)]  # outstanding work handles per set
        self.recv = [dict() for _ in range(depth)]  # root: key -> [world, n, ...] tensor

    def submit(self, slot, tensors):
        """Start gathering the dict ``tensors`` (key -> [n, ...], contiguous) of set ``slot``."""
        import torch
        import torch.distributed as dist
        assert not self.pending[slot], "reclaim() the set before submitting it again"
        for key, t in tensors.items():
            assert t.is_contiguous()
            parts = None
            if self.rank == self.dst:
                buf = self.recv[slot].get(key)
                if buf is None or buf.shape[1:] != t.shape or buf.dtype != t.dtype:
                    buf = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
                    self.recv[slot][key] = buf
                parts = list(buf.unbind(0))
            self.pending[slot].append(dist.gather(t, parts, dst=self.dst, group=self.group, async_op=True))

    def reclaim(self, slot):
        """Order everything issued afterwards (on the current stream) behind the gathers of the set."""
        for w in self.pending[slot]:
            w.wait()
        self.pending[slot] = []

    def drain(self):
        for slot in range(self.depth):
            self.reclaim(slot)

    def result(self, slot, key):
        """Root, after reclaim/drain: the gathered [world * n, ...] tensor of the set (a view)."""
        buf = self.recv[slot][key]
        return buf.reshape((-1,) + tuple(buf.shape[2:]))
