"""Multi-GPU host logic: the batch of independent instances is sharded contiguously, one
process per GPU.  No collective on the data path; after the solve
rank 0 gathers per-instance solutions and residual statistics are reduced (RCCL on GPUs,
gloo in the CPU tests).  No reference counterpart: the reference is one process, one state.
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


def gather_solutions(local: dict, total: int, dst: int = 0, group=None):
    """Gather equally keyed per-instance tensors [n_local, ...] to `dst`; returns the full
    dict on dst (instance order == global order), None elsewhere.  Shards are padded to the
    common ceil size so that one gather per tensor suffices."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-total // world)
    out = {} if rank == dst else None
    host = dist.get_backend(group) == "gloo"   # gloo gathers host tensors only (CPU tests, rehearsals)
    for key in sorted(local):
        t = local[key].cpu() if host else local[key]
        pad = per - t.shape[0]
        if pad:
            t = torch.cat([t, t.new_zeros((pad,) + tuple(t.shape[1:]))], dim=0)
        t = t.contiguous()
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, bufs, dst=dst, group=group)
        if rank == dst:
            out[key] = torch.cat(bufs, dim=0)[:total]
    return out


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
