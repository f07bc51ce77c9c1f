"""torch.autograd glue around the C ABI of libnnr.so.

`render_rays` is the one differentiable operator the host-side `model.Renderer` calls: sampling along rays,
the positional-encoded MLP and alpha-compositing, forward and backward, entirely in the HIP kernels
(reference model/rendering.py:95-132 + model/official_nerf.py:60-96 and autograd through them).
PyTorch is used for device memory, the current stream and the autograd graph edge -- nothing is computed here.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import lib as L


def _require_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("nnr: the render path runs only on an AMD GPU (HIP); got a CPU tensor and there is no CPU fallback")
    if t.device.index is not None and t.device.index != torch.cuda.current_device():
        # the kernels are launched on the current device's current stream (one process per GPU: torch.cuda.set_device(local_rank))
        raise RuntimeError("nnr: tensors live on cuda:%d but the current device is cuda:%d; call torch.cuda.set_device first"
                           % (t.device.index, torch.cuda.current_device()))


# ----------------------------------------------------------------------------------------------------------------------
# packed weights: re-packed only when a parameter tensor changed (optimizer.step bumps tensor._version)
# ----------------------------------------------------------------------------------------------------------------------
_param_epoch = 0   # bumped after every torch optimizer step, see _after_any_optimizer_step


def _after_any_optimizer_step(optimizer, args, kwargs):
    """Global optimizer post-step hook.  tensor._version is NOT a reliable change detector: torch's fused optimizers
    (Adam(fused=True) -- which model.Trainer selects) update parameters in place without bumping it, and a stale packed
    copy means the render kernels silently keep using old weights.  Any optimizer step therefore invalidates every pack."""
    global _param_epoch
    _param_epoch += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_register_step_hook(_after_any_optimizer_step)


def invalidate_packed_weights():
    """For code that changes parameters behind torch's back (custom kernels writing through .data_ptr())."""
    global _param_epoch
    _param_epoch += 1


class _PackCache:
    """Valid only for the very same tensor objects (weak references -- a data_ptr or id() can be recycled by a new model
    after the old one is freed) at the very same in-place versions, with no optimizer step since."""

    def __init__(self, hidden, tensors, packed):
        self.hidden = hidden   # (width, bf16 mode): the two modes have different packed layouts
        self.refs = [weakref.ref(t) for t in tensors]
        self.versions = [t._version for t in tensors]
        self.epoch = _param_epoch
        self.packed = packed

    def matches(self, hidden, tensors):
        return (self.hidden == hidden and self.epoch == _param_epoch and len(self.refs) == len(tensors)
                and all(r() is t for r, t in zip(self.refs, tensors))
                and self.versions == [t._version for t in tensors])


_pack_caches = {}   # id(first weight) -> _PackCache; a handful of live models at most


def _packed_for(cfg, weights, biases):
    """Packed (MFMA-fragment order) copy of the 24 parameter tensors; `weights`/`biases` must be the caller's long-lived
    tensor objects (nn.Parameters), not temporaries."""
    tensors = [*weights, *biases]
    mode = (cfg.hidden, cfg.flags & (L.NNR_F_BF16 | L.NNR_F_SPLIT3 | L.NNR_F_SPLIT2))     # each product mode has its own packed layout
    key = (id(tensors[0]), mode[1])
    hit = _pack_caches.get(key)
    if hit is not None and hit.matches(mode, tensors):
        return hit.packed
    lib = L.load()
    n = lib.nnr_packed_floats(C.byref(cfg))
    packed = torch.empty(n, dtype=torch.float32, device=tensors[0].device)
    ps = L.params_struct([w.detach() for w in weights], [b.detach() for b in biases])
    L.check(lib.nnr_pack_weights(C.byref(cfg), C.byref(ps), L.ptr(packed),
                                 L.stream()), "nnr_pack_weights")
    for k in [k for k, v in _pack_caches.items() if any(r() is None for r in v.refs)]:
        del _pack_caches[k]             # drop entries of models that no longer exist
    _pack_caches[key] = _PackCache(mode, tensors, packed)
    return packed


# ----------------------------------------------------------------------------------------------------------------------
# workspaces and plans, cached per (device, shape); a training workspace is held by the autograd node until backward
# ----------------------------------------------------------------------------------------------------------------------
_ws_pool = {}
_plan_cache = {}


def _cfg_key(cfg: L.Cfg, device):
    return (str(device), cfg.n_rays, cfg.n_samples, cfg.hidden, cfg.flags)


def _take_workspace(cfg: L.Cfg, device) -> torch.Tensor:
    key = _cfg_key(cfg, device)
    pool = _ws_pool.setdefault(key, [])
    if pool:
        return pool.pop()
    n = L.load().nnr_workspace_floats(C.byref(cfg))
    if n == 0:
        raise RuntimeError("nnr: unsupported configuration (hidden_dim must be 128 or 256; num_points <= 1024 for training)")
    return torch.empty(n, dtype=torch.float32, device=device)


def _give_workspace(cfg: L.Cfg, device, ws: torch.Tensor):
    pool = _ws_pool.setdefault(_cfg_key(cfg, device), [])
    if len(pool) < 2:
        pool.append(ws)


def _plan_for(cfg: L.Cfg, device) -> torch.Tensor:
    key = _cfg_key(cfg, device)
    plan = _plan_cache.get(key)
    if plan is None:
        lib = L.load()
        nbytes = lib.nnr_plan_bytes(C.byref(cfg))
        host = np.zeros(nbytes, dtype=np.uint8)
        L.check(lib.nnr_plan_build(C.byref(cfg), host.ctypes.data_as(C.c_void_p)), "nnr_plan_build")
        plan = torch.from_numpy(host).to(device)
        _plan_cache[key] = plan
    return plan


def plan_jobs(cfg: L.Cfg, with_waves: bool = False):
    """Host copy of the weight-gradient plan as a list of WgradJob (for tests / DESIGN inspection); with_waves also
    returns the wave_first table (wave w runs jobs [wave_first[w], wave_first[w+1]))."""
    lib = L.load()
    nj, nw = C.c_int32(0), C.c_int32(0)
    L.check(lib.nnr_plan_counts(C.byref(cfg), C.byref(nj), C.byref(nw)), "nnr_plan_counts")
    nbytes = lib.nnr_plan_bytes(C.byref(cfg))
    assert nbytes >= nj.value * C.sizeof(L.WgradJob) + 4 * (nw.value + 1 + 1)      # + n_heads and the head list (split 0 of every tile)
    raw = (C.c_uint8 * nbytes)()
    L.check(lib.nnr_plan_build(C.byref(cfg), C.cast(raw, C.c_void_p)), "nnr_plan_build")
    jobs = list((L.WgradJob * nj.value).from_buffer_copy(raw, 0))
    if not with_waves:
        return jobs
    first = list((C.c_int32 * (nw.value + 1)).from_buffer_copy(raw, nj.value * C.sizeof(L.WgradJob)))
    return jobs, first


def plan_bf16(cfg: L.Cfg):
    """Host copy of the bf16-mode weight-gradient plan: (jobs, block_first, outputs) -- for tests / DESIGN inspection."""
    lib = L.load()
    nj, nw = C.c_int32(0), C.c_int32(0)
    L.check(lib.nnr_plan_counts(C.byref(cfg), C.byref(nj), C.byref(nw)), "nnr_plan_counts")
    nbytes = lib.nnr_plan_bytes(C.byref(cfg))
    raw = (C.c_uint8 * nbytes)()
    L.check(lib.nnr_plan_build(C.byref(cfg), C.cast(raw, C.c_void_p)), "nnr_plan_build")
    n_blocks = nw.value // 4
    o1 = nj.value * C.sizeof(L.WgradJobB)
    o2 = o1 + 4 * (n_blocks + 1)
    n_out = (nbytes - o2) // C.sizeof(L.WgradOutB)
    assert o2 + n_out * C.sizeof(L.WgradOutB) == nbytes
    jobs = list((L.WgradJobB * nj.value).from_buffer_copy(raw, 0))
    first = list((C.c_int32 * (n_blocks + 1)).from_buffer_copy(raw, o1))
    outs = list((L.WgradOutB * n_out).from_buffer_copy(raw, o2))
    return jobs, first, outs


def workspace_plane(cfg: L.Cfg, ws: torch.Tensor, plane: int, n_rows: Optional[int] = None) -> torch.Tensor:
    """One workspace plane as (rows, width) -- used by the parity tests to localise a mismatch.  A view for the fp32 planes; the
    planes a bf16 training workspace stores as tile-major bf16 (hidden activations 11..18, 20, the encodings' copies 21, 22 and the
    gradients 31..38, 40: nnr_layout.h) come back in natural [sample][feature] order, converted to fp32."""
    pitch = C.c_int32(0)
    off = L.load().nnr_ws_plane(C.byref(cfg), plane, C.byref(pitch))
    if off < 0:
        raise KeyError(plane)
    S = cfg.n_rays * cfg.n_samples
    rows = n_rows if n_rows is not None else S
    if L.load().nnr_ws_plane_layout(C.byref(cfg), plane) == 2:
        # tile-major fp32 (the gradient planes of a three-term training workspace): [chunk of 32 samples][octet j][half h][sample c][i],
        # feature = 8 j + 4 h + i  (nnr_layout.h: tile32_index); back in natural [sample][feature] order (a copy)
        W = pitch.value
        chunks = (rows + 31) // 32
        t = ws[off: off + chunks * 32 * W].view(chunks, W // 8, 2, 32, 4)                       # [chunk][j][h][c][i]
        return t.permute(0, 3, 1, 2, 4).reshape(chunks * 32, W)[:rows]                          # [chunk][c] x [j][h][i]
    stored_bf16 = (cfg.flags & L.NNR_F_BF16) and (cfg.flags & L.NNR_F_TRAIN) and (11 <= plane <= 18 or 20 <= plane <= 22 or 31 <= plane <= 38 or plane == 40)
    if not stored_bf16:
        return ws[off: off + rows * pitch.value].view(rows, pitch.value)
    # tile-major: [chunk of 32 samples][group of 16 features][half h][sample c][second quad j][i]; feature = 16 g + 4 h + 8 j + i
    G = pitch.value // 8
    chunks = (rows + 31) // 32
    t = ws[off: off + chunks * 32 * pitch.value].view(torch.bfloat16).view(chunks, G, 2, 32, 2, 4)    # [chunk][g][h][c][j][i]
    return t.permute(0, 3, 1, 4, 2, 5).reshape(chunks * 32, G * 16)[:rows].float()                    # [chunk][c] x [g][j][h][i]


# ----------------------------------------------------------------------------------------------------------------------
# the operator
# ----------------------------------------------------------------------------------------------------------------------
class _RenderRays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pts_o, pts_d, view_d, z_lo, z_hi, jitter, opts, *params):
        weights, biases = params[:L.N_LAYERS], params[L.N_LAYERS:]
        _require_gpu(pts_o)
        dev = pts_o.device
        R, N = pts_o.shape[0], z_lo.shape[0]
        need_grad = any(ctx.needs_input_grad)
        cfg = L.make_cfg(R, N, opts["hidden"], dist_alpha=opts["dist_alpha"], white_bg=opts["white_bg"],
                         relu_sigma=opts["relu_sigma"], train=need_grad, bf16=opts.get("bf16", False))
        lib = L.load()
        f32 = dict(dtype=torch.float32, device=dev)
        pts_o, pts_d, view_d = (t.detach().contiguous().float() for t in (pts_o, pts_d, view_d))
        z_lo, z_hi = z_lo.detach().contiguous().float(), z_hi.detach().contiguous().float()
        jit = jitter.detach().contiguous().float().view(R, N) if jitter is not None else None
        packed = _packed_for(cfg, *opts["params"])
        ws = _take_workspace(cfg, dev)
        rgb = torch.empty(R, 3, **f32)
        dist = torch.empty(R, **f32)
        # forward-only callers that do not want the per-sample outputs (samples=False) get the fused path of nnr_render_fwd: the
        # compositing happens in the MLP kernel's epilogue and nothing per-sample is written to HBM
        want_samples = need_grad or opts.get("samples", True)
        alpha = torch.empty(R, N, **f32) if want_samples else None
        zv = torch.empty(R, N, **f32) if want_samples else None
        st = L.stream()
        L.check(lib.nnr_render_fwd(C.byref(cfg), L.ptr(pts_o), L.ptr(pts_d), L.ptr(view_d), L.ptr(z_lo), L.ptr(z_hi),
                                   L.ptr(jit), L.ptr(packed), L.ptr(rgb), L.ptr(dist), L.ptr(alpha), L.ptr(zv), L.ptr(ws), st),
                "nnr_render_fwd")
        if need_grad:
            ctx.cfg, ctx.ws, ctx.packed, ctx.dev = cfg, ws, packed, dev
            ctx.shapes = [tuple(p.shape) for p in params]
        else:
            _give_workspace(cfg, dev, ws)
        if want_samples:
            ctx.mark_non_differentiable(alpha, zv)
        ctx.set_materialize_grads(False)      # undefined upstream gradients arrive as None, not as zero-filled tensors
        return rgb, dist, alpha, zv

    @staticmethod
    def backward(ctx, d_rgb, d_dist, _da, _dz):
        cfg, ws, dev = ctx.cfg, ctx.ws, ctx.dev
        if ws is None:
            raise RuntimeError("nnr: backward called twice on the same render (workspace already released)")
        lib = L.load()
        R = cfg.n_rays
        f32 = dict(dtype=torch.float32, device=dev)
        d_rgb = (d_rgb if d_rgb is not None else torch.zeros(R, 3, **f32)).contiguous().float()
        d_dist = (d_dist if d_dist is not None else torch.zeros(R, **f32)).contiguous().float()
        st = L.stream()
        need_w = any(ctx.needs_input_grad[7:])
        need_rays = any(ctx.needs_input_grad[:3])
        L.check(lib.nnr_composite_bwd(C.byref(cfg), L.ptr(d_rgb), L.ptr(d_dist), L.ptr(ws), st), "nnr_composite_bwd")
        L.check(lib.nnr_mlp_dgrad(C.byref(cfg), L.ptr(ctx.packed), L.ptr(ws), st), "nnr_mlp_dgrad")
        grads: List[Optional[torch.Tensor]] = [None] * (2 * L.N_LAYERS)
        if need_w:
            sizes = [int(np.prod(s)) for s in ctx.shapes]
            offs = np.cumsum([0] + [(n + 3) // 4 * 4 for n in sizes])          # keep every view 16-byte aligned
            # ONE allocation for all 24 gradients (the weight-gradient stage overwrites every element: no zero-fill launch) + GRAD_TAIL spare
            # floats behind them: autograd adopts the views as the parameters' .grad, so under data parallelism the step's all-reduce runs
            # IN PLACE on this buffer, the handful of pose / distortion gradients and logged scalars riding in the tail (nnr/parallel.py)
            # Under data parallelism the buffer is all-reduced as it lies, alignment gaps and tail included: zero-filled then (a NaN left over in a
            # gap would be summed -- never read, but a NaN check on the collective would trip), and REGISTERED with its used length: the in-place
            # path of nnr/parallel.py is taken for a buffer this function made, not for any storage that happens to hold many gradients.
            from . import parallel as _par
            if _par.world_size() > 1 or _par.always_reduce():
                flat = torch.zeros(int(offs[-1]) + GRAD_TAIL, **f32)
                register_flat_grads(flat, int(offs[-1]))
            else:
                flat = torch.empty(int(offs[-1]) + GRAD_TAIL, **f32)
            views = [flat[offs[i]: offs[i] + sizes[i]].view(ctx.shapes[i]) for i in range(2 * L.N_LAYERS)]
            gs = L.params_struct(views[:L.N_LAYERS], views[L.N_LAYERS:])
            L.check(lib.nnr_mlp_wgrad(C.byref(cfg), L.ptr(ctx.packed), C.byref(gs), L.ptr(_plan_for(cfg, dev)), L.ptr(ws), st), "nnr_mlp_wgrad")
            grads = [v if ctx.needs_input_grad[7 + i] else None for i, v in enumerate(views)]
        d_o = d_d = d_v = None
        if need_rays:
            d_o, d_d, d_v = (torch.empty(R, 3, **f32) for _ in range(3))
            L.check(lib.nnr_ray_reduce(C.byref(cfg), L.ptr(d_o), L.ptr(d_d), L.ptr(d_v), L.ptr(ws), st), "nnr_ray_reduce")
        ctx.ws = None
        _give_workspace(cfg, dev, ws)
        return (d_o, d_d, d_v, None, None, None, None, *grads)


GRAD_TAIL = 2048      # spare floats behind the flat weight-gradient buffer (see _RenderRays.backward)
_flat_grads = {}      # storage data_ptr -> (weak reference to the flat buffer, floats used by the gradient views): buffers _RenderRays.backward made


def register_flat_grads(flat: torch.Tensor, used: int):
    import weakref
    for k in [k for k, (r, _) in _flat_grads.items() if r() is None]:
        del _flat_grads[k]
    _flat_grads[flat.untyped_storage().data_ptr()] = (weakref.ref(flat), used)


def flat_grads_of(storage_ptr: int):
    """(flat buffer, used floats) if `storage_ptr` is the storage of a LIVE buffer made by _RenderRays.backward under data parallelism, else None."""
    hit = _flat_grads.get(storage_ptr)
    if hit is None or hit[0]() is None:
        return None
    return hit[0](), hit[1]


def render_rays(pts_o: torch.Tensor, pts_d: torch.Tensor, view_d: torch.Tensor, z_lo: torch.Tensor, z_hi: torch.Tensor,
                jitter: Optional[torch.Tensor], weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], *,
                hidden: int, dist_alpha: bool, white_bg: bool, relu_sigma: bool, bf16: bool = False, samples: bool = True):
    """(R,3) sampling origin / direction / view direction, (N) z interval tables, optional (R,N) jitter, the 12
    nn.Linear weights and biases in state_dict order  ->  rgb (R,3), dist (R), alpha (R,N), z (R,N).
    Differentiable w.r.t. pts_o, pts_d, view_d, weights, biases.  bf16: bf16-MFMA products (fp32 accumulation) in the MLP
    forward and input-gradient kernels (NNR_F_BF16); weight gradients stay fp32.  samples=False (only honoured under torch.no_grad):
    alpha and z come back as None and the renderer composites inside the MLP kernel (16 bytes written per ray)."""
    opts = dict(hidden=hidden, dist_alpha=dist_alpha, white_bg=white_bg, relu_sigma=relu_sigma, bf16=bool(bf16), samples=bool(samples),
                params=(list(weights), list(biases)))    # the caller's own tensor objects: identity keys the pack cache
    if not torch.is_grad_enabled():
        # forward-only (eval / visualisation inside torch.no_grad): no stash, small workspace
        return _RenderRays.apply(pts_o.detach(), pts_d.detach(), view_d.detach(), z_lo, z_hi, jitter, opts,
                                 *[w.detach() for w in weights], *[b.detach() for b in biases])
    return _RenderRays.apply(pts_o, pts_d, view_d, z_lo, z_hi, jitter, opts, *weights, *biases)


def mlp_points(pts: torch.Tensor, view: torch.Tensor, weights, biases, *, hidden: int):
    """Forward-only evaluation of the MLP on free-standing points: (S,3),(S,3) -> rgb (S,3), sigma_raw (S,).
    Each point is rendered as a one-sample ray (origin = point, direction = 0) through the same fused kernel."""
    _require_gpu(pts)
    S, dev = pts.shape[0], pts.device
    cfg = L.make_cfg(S, 1, hidden)
    lib = L.load()
    pts = pts.detach().contiguous().float()
    view = view.detach().contiguous().float()
    zeros3 = torch.zeros_like(pts)
    z0 = torch.zeros(1, dtype=torch.float32, device=dev)
    packed = _packed_for(cfg, list(weights), list(biases))
    ws = _take_workspace(cfg, dev)
    st = L.stream()
    L.check(lib.nnr_mlp_fwd(C.byref(cfg), L.ptr(pts), L.ptr(zeros3), L.ptr(view), L.ptr(z0), L.ptr(z0), None,
                            L.ptr(packed), L.ptr(ws), st), "nnr_mlp_fwd")
    out = workspace_plane(cfg, ws, 0).clone()
    _give_workspace(cfg, dev, ws)
    return out[:, :3], out[:, 3]
