"""Ray-sharded data parallelism: one process per GPU, one flat all-reduce per step (RCCL over xGMI; `gloo` in the
CPU tests).  The reference has no multi-GPU code at all -- this is the path SURVEY.md section 8e defines:
rays are independent units, so rank k renders rays [k*R/W, (k+1)*R/W) of the step's permutation and the only exchange
is the SUM of the 2.4 MB of gradients (+ the handful of scalars train.py logs).
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def always_reduce() -> bool:
    """NNR_DP_ALWAYS_REDUCE=1: run the step's gradient all-reduce on a ONE-rank group too (a no-op numerically) -- lets a one-GPU box
    exercise the RCCL path of the step (bench.py with NNR_BENCH_FORCE_DIST=1, tests/test_gpu_bench_ranks.py)."""
    import os
    return os.environ.get('NNR_DP_ALWAYS_REDUCE') == '1' and dist.is_available() and dist.is_initialized()


def shard_bounds(n: int, rank_: int, world: int):
    """Contiguous [lo, hi) slice of n rays owned by `rank_`; slices differ by at most one ray and cover [0, n)."""
    base, rem = divmod(n, world)
    lo = rank_ * base + min(rank_, rem)
    return lo, lo + base + (1 if rank_ < rem else 0)


def gather_rays(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Every rank's shard of a per-ray tensor, concatenated in ray order: (n_total, ...).  This rank's slice is the live tensor
    (autograd flows into it), the others are constants -- the gradient of a loss over ALL rays with respect to the local rays,
    which is what the SUM all-reduce of the parameter gradients needs.  One all-gather of ceil(n_total / W) rows per rank."""
    world, me = world_size(), rank()
    if world == 1:
        return local
    per = -(-n_total // world)
    buf = local.detach().new_zeros((per,) + tuple(local.shape[1:]))
    buf[:local.shape[0]] = local.detach()
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf)
    pieces = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        pieces.append(local if r == me else outs[r][:hi - lo])
    return torch.cat(pieces)


_LOGGED = ('loss', 'loss_rgb', 'loss_depth', 'l2_mean', 'loss_dist_1st', 'loss_dist_2nd', 'loss_pc', 'loss_rgb_s',
           'loss_depth_consistency')


def allreduce_gradients(params: Iterable[torch.nn.Parameter], loss_dict: Optional[Dict[str, torch.Tensor]] = None):
    """SUM-reduce every .grad (and the logged loss scalars) across ranks through ONE flat fp32 bucket.
    Parameters whose grad is None on this rank (e.g. an unused pose row) contribute zeros so that bucket layouts
    agree on every rank."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    dev = params[0].device
    pieces = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params]
    keys = [k for k in _LOGGED if loss_dict is not None and torch.is_tensor(loss_dict.get(k))]
    pieces += [loss_dict[k].detach().reshape(1).float() for k in keys]
    flat = torch.cat(pieces)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    views, off = [], 0
    for p in params:
        n = p.numel()
        views.append(flat[off:off + n].view_as(p))
        off += n
    have = [i for i, p in enumerate(params) if p.grad is not None]
    if have:
        torch._foreach_copy_([params[i].grad for i in have], [views[i] for i in have])   # one multi-tensor launch
    for i, p in enumerate(params):
        if p.grad is None:
            p.grad = views[i].clone()
    for k in keys:
        # the autograd-free logged value ('loss' is no longer needed for backward at this point); a view of the private
        # bucket, not a copy -- each copy would be one more launch per step
        loss_dict[k] = flat[off]
        off += 1
