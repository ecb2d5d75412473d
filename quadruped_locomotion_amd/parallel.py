"""Multi-GPU host logic: the batch of independent instances is sharded contiguously, one
process per GPU.  No collective on the data path; after the solve
rank 0 collects the per-instance solutions and the residual statistics are reduced (RCCL on GPUs,
gloo in the CPU tests).  No reference counterpart: the reference is one process, one state.

Collection (`ResultCollector`): rank 0 owns the final [total, ...] arrays, allocated ONCE; every step each
peer sends its shard and rank 0 receives it straight into that shard's slice -- grouped isend / irecv
(`torch.distributed.batch_isend_irecv`, on RCCL one ncclGroup of ncclSend / ncclRecv: a direct transfer over
each peer's own xGMI link to the root, not a ring).  No per-call allocation, no padding, no concatenation.
With two buffer sets and a side stream the collection of step k overlaps the solve of step k + 1.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of ceil(total / world) instances (last shards may be short or empty)."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError((total, world, rank))
    per = -(-total // world)
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


class ResultCollector:
    """Collects equally keyed per-instance tensors [n_local, ...] of every rank into [total, ...] tensors on
    rank `dst`, instance order == global order.

    spec: {key: (trailing shape tuple, dtype)}.  `buffers` destination sets (2: double buffering).
    `device`: where rank dst keeps the collected arrays (and where the shards live).  On a gloo group
    (CPU tests, one-GPU rehearsals) transfers go through preallocated host staging buffers, since gloo moves
    host tensors only."""

    def __init__(self, spec: dict, total: int, device, dst: int = 0, group=None, buffers: int = 1):
        self.group, self.dst, self.total = group, dst, total
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.spec = {k: (tuple(shape), dtype) for k, (shape, dtype) in sorted(spec.items())}
        self.host = dist.get_backend(group) == "gloo"
        self.device = torch.device("cpu") if self.host else torch.device(device)
        self.lo, self.hi = shard_bounds(total, self.world, self.rank)
        self.full = None
        if self.rank == dst:
            self.full = [{k: torch.empty((total,) + shape, dtype=dtype, device=self.device)
                          for k, (shape, dtype) in self.spec.items()} for _ in range(buffers)]
        # host staging of the local shard (gloo only)
        self.stage = None
        if self.host and self.rank != dst:
            self.stage = {k: torch.empty((self.hi - self.lo,) + shape, dtype=dtype)
                          for k, (shape, dtype) in self.spec.items()}

    def slot(self, i: int = 0):
        """The collected arrays of buffer set i on rank dst (None elsewhere)."""
        return self.full[i % len(self.full)] if self.full is not None else None

    def start(self, local: dict, i: int = 0):
        """Begin the collection of `local` into buffer set i; returns the requests to pass to wait().
        On RCCL the transfers are ordered after the work already queued on the current stream."""
        assert sorted(local) == list(self.spec), (sorted(local), list(self.spec))
        ops = []
        if self.rank == self.dst:
            dstbuf = self.full[i % len(self.full)]
            for k in self.spec:
                t = local[k]
                assert t.shape[0] == self.hi - self.lo, (k, t.shape, self.lo, self.hi)
                if self.hi > self.lo:
                    dstbuf[k][self.lo:self.hi].copy_(t, non_blocking=not self.host)
            for r in range(self.world):
                if r == self.dst:
                    continue
                lo, hi = shard_bounds(self.total, self.world, r)
                if hi > lo:
                    for k in self.spec:
                        ops.append(dist.P2POp(dist.irecv, dstbuf[k][lo:hi], r, self.group))
        elif self.hi > self.lo:
            for k in self.spec:
                t = local[k]
                assert t.shape[0] == self.hi - self.lo and t.is_contiguous(), (k, t.shape)
                if self.host:
                    self.stage[k].copy_(t)
                    t = self.stage[k]
                ops.append(dist.P2POp(dist.isend, t, self.dst, self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    @staticmethod
    def wait(reqs):
        for r in reqs:
            r.wait()

    def collect(self, local: dict, i: int = 0):
        """start + wait; returns slot(i)."""
        self.wait(self.start(local, i))
        return self.slot(i)


def gather_solutions(local: dict, total: int, dst: int = 0, group=None):
    """One-shot form of ResultCollector (allocates the destination): the full dict on dst, None elsewhere."""
    spec = {k: (tuple(v.shape[1:]), v.dtype) for k, v in local.items()}
    dev = next(iter(local.values())).device if local else "cpu"
    return ResultCollector(spec, total, dev, dst=dst, group=group).collect(local)


def reduce_stats(max_stats: torch.Tensor, sum_stats: torch.Tensor, group=None):
    """All-reduce residual statistics: element-wise MAX of `max_stats` (max residuals /
    violations), SUM of `sum_stats` (counts, iteration totals).  In place; returns both."""
    if dist.get_backend(group) == "gloo" and max_stats.is_cuda:
        mx, sm = max_stats.cpu(), sum_stats.cpu()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        dist.all_reduce(sm, op=dist.ReduceOp.SUM, group=group)
        max_stats.copy_(mx)
        sum_stats.copy_(sm)
        return max_stats, sum_stats
    dist.all_reduce(max_stats, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(sum_stats, op=dist.ReduceOp.SUM, group=group)
    return max_stats, sum_stats


def local_stats(out: dict):
    """Per-rank statistics from a solve's output dict (works for CPU or GPU tensors)."""
    mx = torch.stack([out["eq_res_max"].max() if out["eq_res_max"].numel() else out["eq_res_max"].new_zeros(()),
                      out["ineq_viol_max"].max() if out["ineq_viol_max"].numel() else out["ineq_viol_max"].new_zeros(())])
    sm = torch.stack([(out["status"] != 0).sum().to(torch.float64), out["iters"].sum().to(torch.float64),
                      torch.tensor(float(out["status"].shape[0]), dtype=torch.float64, device=out["status"].device)])
    return mx.to(torch.float64), sm
