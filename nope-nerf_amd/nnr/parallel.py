"""Ray-sharded data parallelism: one process per GPU, one flat all-reduce per step (RCCL over xGMI; `gloo` in the
CPU tests).  The reference has no multi-GPU code at all -- this is the path SURVEY.md section 8e defines:
rays are independent units, so rank k renders rays [k*R/W, (k+1)*R/W) of the step's permutation and the only exchange
is the SUM of the 2.4 MB of gradients (+ the handful of scalars train.py logs).
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist


_auto = {"done": False, "made_group": False}


def auto_init() -> bool:
    """Called by `import model`: under a multi-process launcher -- `torchrun --nproc-per-node N train.py cfg.yaml` sets RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR, MASTER_PORT -- bind this process to its GPU and join the process group, so that the reference's UNMODIFIED
    train.py (which never calls init_process_group and asks for the device "cuda": /root/reference/train.py:18-60) trains data-parallel:
      * torch.cuda.set_device(LOCAL_RANK): train.py's torch.device("cuda") then means this rank's GPU;
      * init_process_group: 'nccl' (= RCCL) with the device bound when there is a GPU per rank, 'gloo' otherwise (CPU runs, or several ranks
        sharing one GPU: NNR_DIST_BACKEND=gloo); nothing happens when a group already exists or WORLD_SIZE is 1 / unset;
      * os.makedirs tolerates a directory that another rank created between train.py's os.path.exists() and os.makedirs()
        (train.py:176-177, 234-235: every rank runs those lines at the same moment).
    File outputs are rank 0's: CheckpointIO.save / backup_model_best, model.common.backup and Trainer.render_visdata write nothing elsewhere.
    Returns True when this call created the group."""
    import os
    if _auto["done"]:
        return _auto["made_group"]
    _auto["done"] = True
    try:
        world = int(os.environ.get("WORLD_SIZE", "1"))
    except ValueError:
        world = 1
    if world <= 1 or "RANK" not in os.environ or not dist.is_available() or dist.is_initialized():
        return False
    if os.environ.get("NNR_NO_AUTO_DIST") == "1":
        return False
    rank_, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    backend = os.environ.get("NNR_DIST_BACKEND")
    have_gpu = torch.cuda.is_available()
    if have_gpu:
        n = torch.cuda.device_count()
        if backend is None:      # one GPU per local rank: RCCL; fewer (ranks sharing a GPU, as the one-GPU tests do): gloo
            backend = "nccl" if n >= int(os.environ.get("LOCAL_WORLD_SIZE", world)) else "gloo"
        torch.cuda.set_device(local % n)
    elif backend is None:
        backend = "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
    dist.init_process_group(backend, rank=rank_, world_size=world, **kw)
    _auto["made_group"] = True
    import atexit

    def _bye():
        if dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:
                pass
    atexit.register(_bye)
    real_makedirs = os.makedirs

    def makedirs(name, mode=0o777, exist_ok=False):      # every rank: whoever loses the race between exists() and makedirs() carries on
        try:
            return real_makedirs(name, mode, exist_ok)
        except FileExistsError:
            if not os.path.isdir(name):
                raise
    os.makedirs = makedirs
    return True


def is_writer() -> bool:
    """Whether this process writes the run's files (checkpoints, visualisations, config backups): rank 0, or the only process."""
    return rank() == 0


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


def _shared_buffer(params):
    """The flat buffer most of the gradients are views of (nnr.ops._RenderRays.backward allocates the 24 MLP gradients as ONE tensor with
    spare room behind them, and autograd adopts the views as .grad): (whole storage as a 1-D tensor, end of the used part, the parameters
    whose gradients live in it) -- or None when there is no such buffer (CPU stand-in, torch fallbacks)."""
    groups = {}
    for p in params:
        g = p.grad
        if g is not None and g.dtype == torch.float32 and g.is_contiguous():
            groups.setdefault(g.untyped_storage().data_ptr(), []).append(p)
    if not groups:
        return None
    from . import ops
    for ptr, members in groups.items():      # only a buffer _RenderRays.backward registered: its tail is spare by construction, its gaps are zero
        hit = ops.flat_grads_of(ptr)
        if hit is None:
            continue
        whole, used = hit
        if max(p.grad.storage_offset() + p.grad.numel() for p in members) <= used and whole.numel() > used:
            return whole, used, members
    return None


def allreduce_gradients(params: Iterable[torch.nn.Parameter], loss_dict: Optional[Dict[str, torch.Tensor]] = None):
    """SUM-reduce every .grad (and the logged loss scalars) across ranks with ONE all-reduce of one flat fp32 buffer.
    On the GPU path that buffer is the one the weight-gradient kernels wrote (the 24 MLP gradients are views of a single allocation with a
    spare tail, nnr/ops.py): the all-reduce runs in place, the MLP gradients are never copied, and the few other gradients (pose and
    distortion tables), the logged scalars and the flags below are packed into the tail by one cat and read back by one multi-tensor copy.
    Without such a buffer (CPU stand-in) everything is packed into a private bucket and copied back, as in rounds 1-4.
    A parameter whose grad is None on this rank (an unused table) contributes zeros so that bucket layouts agree on every rank; one flag
    per parameter rides along ("this rank has a gradient"), and a parameter that NO rank has a gradient for keeps
    grad = None afterwards -- exactly the single-process state.  That matters: Adam skips a parameter without a gradient (no momentum
    step, no step-counter increment), and the reference leaves the distortion scales without one whenever the step's frame is the gauge
    camera (model/distortions.py:23-24); handing Adam a zero-filled gradient instead moved the scales by their momentum in those steps
    and a two-rank run of train.py drifted from the single-process run (tests/test_dropin_torchrun.py)."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    shared = _shared_buffer(params)
    in_place = set(id(p) for p in shared[2]) if shared else set()
    rest = [p for p in params if id(p) not in in_place]
    pieces = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in rest]
    keys = [k for k in _LOGGED if loss_dict is not None and torch.is_tensor(loss_dict.get(k))]
    pieces += [loss_dict[k].detach().reshape(1).float() for k in keys]
    dev = (shared[0] if shared else pieces[0]).device
    pieces.append(torch.tensor([0.0 if p.grad is None else 1.0 for p in rest], dtype=torch.float32).to(dev, non_blocking=True))
    n_small = sum(t.numel() for t in pieces)
    if shared and shared[1] + n_small <= shared[0].numel():
        whole, used, _ = shared
        flat = whole[used:used + n_small]
        torch.cat(pieces, out=flat)
        dist.all_reduce(whole[:used + n_small], op=dist.ReduceOp.SUM)      # (alignment padding between the views is summed too: never read)
    else:
        if shared:      # (the tail is too small for this model's tables: private bucket for everything)
            rest, pieces = params, [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params] + pieces[len(rest):-1] + \
                [torch.tensor([0.0 if p.grad is None else 1.0 for p in params], dtype=torch.float32).to(dev, non_blocking=True)]
        flat = torch.cat(pieces)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    views, off = [], 0
    for p in rest:
        n = p.numel()
        views.append(flat[off:off + n].view_as(p))
        off += n
    have = [i for i, p in enumerate(rest) if p.grad is not None]
    if have:
        torch._foreach_copy_([rest[i].grad for i in have], [views[i] for i in have])   # one multi-tensor launch
    missing = [i for i, p in enumerate(rest) if p.grad is None]
    n_keys = len(keys)
    if missing:      # rare (a rank without a gradient for a table): did any other rank have one?  One small host read, only then.
        flags = flat[off + n_keys:].cpu()
        for i in missing:
            if float(flags[i]) > 0.0:
                rest[i].grad = views[i].clone()
    if keys:
        # the autograd-free logged values ('loss' is no longer needed for backward at this point): ONE small copy out of the reduced buffer -- as
        # views of it they pinned the whole gradient buffer and changed under the caller when the next step (or zero_grad(set_to_none=False))
        # rewrote it
        vals = flat[off:off + n_keys].clone()
        for i, k in enumerate(keys):
            loss_dict[k] = vals[i]
