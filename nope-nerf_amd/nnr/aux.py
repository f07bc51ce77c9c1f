"""The per-image losses between a frame and its neighbour as two fused HIP calls (SURVEY 8 f1 + f2): point-cloud loss and
surface re-projection loss of reference model/training.py:315-358 + model/losses.py:114-157, forward and backward.
CUDA tensors only; model/training.py keeps the torch expression for what this does not cover (CPU, batches of images)."""
import ctypes as C

import torch

from . import lib as L


def _st():
    return L.stream()


class _AuxTerms(torch.autograd.Function):
    @staticmethod
    def forward(ctx, d1_img, d2_img, rel, scale2, img1r, img2r, K, Kinv, hr, wr, nl, flags, shard, aff=None, mats=None, weights=None):
        lib = L.load()
        dev = d1_img.device
        f32 = dict(dtype=torch.float32, device=dev)
        d1, d2 = d1_img.detach().contiguous().float(), d2_img.detach().contiguous().float()
        hd, wd = d1.shape[-2:]
        if K.requires_grad or Kinv.requires_grad:      # a learnable focal length (reference model/training.py:247-252)
            flags |= L.AUX_GRAD_K
        m = None
        if mats is not None:                           # rel, the distortion pairs and scale2 are slices of ONE tensor (nnr.camera.step_rays' mats)
            m = mats.detach()
            if m.dtype != torch.float32 or not m.is_contiguous() or m.numel() != 56 or (flags & L.AUX_GRAD_K):
                raise RuntimeError("nnr_aux: mats must be the 56 contiguous floats of nnr.camera.step_rays (and K constant)")
            rel_c, af = m[34:50], m[50:54]
            s2 = m[54:55] if (flags & L.AUX_SCALE_PCS) else None
            flags |= L.AUX_AFFINE | L.AUX_MATS_GRAD
        else:
            rel_c = rel.detach().reshape(16).contiguous().float()
            s2 = scale2.detach().reshape(1).contiguous().float() if scale2 is not None else None
            af = aff.detach().reshape(4).contiguous().float() if aff is not None else None
            if aff is not None:                        # the maps are RAW; (scale1, shift1, scale2, shift2) is applied in the kernels
                flags |= L.AUX_AFFINE
        cfg = L.AuxCfg(int(hd), int(wd), int(hr), int(wr), float(nl), int(flags), int(shard[0]), int(shard[1]))
        if weights is not None:                        # out[3] = w_pc loss_pc + w_rgbs loss_rgb_s
            cfg.flags |= L.AUX_WEIGHTED
            cfg.w_pc, cfg.w_rgbs = float(weights[0]), float(weights[1])
        n_ws = lib.nnr_aux_workspace_floats(C.byref(cfg))
        if n_ws == 0:
            raise RuntimeError("nnr_aux: bad configuration %r" % ((hd, wd, hr, wr),))
        ws = torch.empty(n_ws + 2, **f32)
        ws = ws[(ws.data_ptr() % 8) // 4:]                         # 8-byte alignment for the 64-bit items
        K_c, Kinv_c = (t.detach().reshape(16).contiguous().float() for t in (K, Kinv))
        i1 = img1r.detach().contiguous().float() if img1r is not None else None
        i2 = img2r.detach().contiguous().float() if img2r is not None else None
        out = torch.empty(4, **f32)
        p = lambda t: L.ptr(t) if t is not None else None
        L.check(lib.nnr_aux_terms_fwd(C.byref(cfg), p(d1), p(d2), p(i1), p(i2), p(K_c), p(Kinv_c), p(rel_c), p(s2), p(af), p(out), p(ws),
                                      _st()), "nnr_aux_terms_fwd")
        ctx.cfg, ctx.ws = cfg, ws
        ctx.tensors = (d1, d2, i1, i2, K_c, Kinv_c, rel_c, s2, af, m)
        ctx.shapes = (d1_img.shape, d2_img.shape, None if rel is None else rel.shape, None if scale2 is None else scale2.shape, K.shape, Kinv.shape)
        ctx.aff_shape = None if aff is None else aff.shape
        ctx.mats_shape = None if mats is None else mats.shape
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def backward(ctx, g_pc, g_rgbs, _g_count, g_w):
        lib = L.load()
        d1, d2, i1, i2, K_c, Kinv_c, rel_c, s2, af, m = ctx.tensors
        need_d1, need_d2, need_rel, need_s2 = ctx.needs_input_grad[:4]
        f32 = dict(dtype=torch.float32, device=d1.device)
        cfg = L.AuxCfg.from_buffer_copy(ctx.cfg)
        weighted = bool(cfg.flags & L.AUX_WEIGHTED)
        if weighted and g_w is not None and g_pc is None and g_rgbs is None:
            # the usual case: only the weighted sum is in the loss -- its ONE upstream gradient goes to the kernels as it is (they apply the weights)
            g_out = g_w.reshape(1).float().contiguous()
        else:
            cfg.flags &= ~L.AUX_WEIGHTED
            parts = []
            for g, w in ((g_pc, cfg.w_pc), (g_rgbs, cfg.w_rgbs)):
                t = g.reshape(()).float() if g is not None else None
                if weighted and g_w is not None:
                    t = g_w.reshape(()).float() * w if t is None else t + g_w.reshape(()).float() * w
                parts.append(t if t is not None else torch.zeros((), **f32))
            g_out = torch.stack(parts)
        g_d1 = torch.zeros_like(d1) if (need_d1 and af is None) else None
        g_d2 = torch.zeros_like(d2) if (need_d2 and af is None) else None
        g_rs = torch.empty(56 if m is not None else 44, **f32)
        p = lambda t: L.ptr(t) if t is not None else None
        L.check(lib.nnr_aux_terms_bwd(C.byref(cfg), p(d1), p(d2), p(i1), p(i2), p(K_c), p(Kinv_c), p(rel_c), p(s2), p(af), p(g_out),
                                      p(g_d1), p(g_d2), p(g_rs), p(ctx.ws), _st()), "nnr_aux_terms_bwd")
        sh1, sh2, shr, shs, shk, shki = ctx.shapes
        g1 = g_d1.view(sh1) if g_d1 is not None else None
        g2 = g_d2.view(sh2) if g_d2 is not None else None
        if m is not None:      # one gradient tensor for the one input tensor
            return (g1, g2, None, None, None, None, None, None, None, None, None, None, None, None,
                    g_rs.view(ctx.mats_shape) if ctx.needs_input_grad[14] else None, None)
        rows = lambda lo, shape: torch.cat([g_rs[lo:lo + 12], torch.zeros(4, **f32)]).view(shape)      # the last row is not read
        g_rel = rows(0, shr) if need_rel else None
        g_s2 = g_rs[12].view(shs) if (need_s2 and shs is not None) else None
        grad_k = bool(cfg.flags & L.AUX_GRAD_K)
        g_k = rows(16, shk) if (grad_k and ctx.needs_input_grad[6]) else None
        g_kinv = rows(28, shki) if (grad_k and ctx.needs_input_grad[7]) else None
        g_aff = g_rs[40:44].view(ctx.aff_shape) if (af is not None and ctx.needs_input_grad[13]) else None
        return (g1, g2, g_rel, g_s2, None, None, g_k, g_kinv, None, None, None, None, None, g_aff, None, None)


def aux_terms(d1_img, d2_img, rel, scale2, img1r, img2r, K, Kinv, res, nearest_limit, *, rgb_s=True, pc=True, scale_pcs=True,
              detach_rgbs_scale=False, ssim=False, shard=(0, 0), aff=None, shift_first=False, mats=None, weights=None):
    """(loss_pc, loss_rgb_s, n_valid) for one frame pair.  d1_img/d2_img: (..., hd, wd) depth maps (scaled + shifted), rel:
    (..., 4, 4) relative transform, scale2: scalar tensor, img1r/img2r: (..., 3, hr, wr), K/Kinv: (..., 4, 4).
    ssim: training.with_ssim (reference losses.py:153-155).
    shard = (lo, hi): data parallelism -- only the sums over the source points [lo, hi) of the res[0]*res[1] grid, with the global
    normalisers, so that the SUM over ranks is the single-GPU loss / gradient ((0, 0) = all points).
    aff = (4,) tensor (scale1, shift1, scale2, shift2): d1_img / d2_img are then the RAW mono-depth maps and the per-image distortion
    (shift_first: (depth + shift) * scale) is applied to the sampled values inside the kernels; its gradient comes back as one (4,) tensor
    instead of two depth-map-sized gradient images and their reductions.
    mats = the 56-float block of nnr.camera.step_rays(ref >= 0) INSTEAD of rel / scale2 / aff (pass None for those): the kernels read the
    three slices in place and the backward returns one gradient for the one tensor (no slice backward launches).
    weights = (pc_weight, rgb_s_weight): a fourth result, pc_weight * loss_pc + rgb_s_weight * loss_rgb_s, formed by the finishing kernel
    (three launches forward, two backward less than the torch expression; the same roundings)."""
    if not d1_img.is_cuda:
        raise RuntimeError("nnr.aux needs CUDA tensors (no CPU fallback)")
    flags = (L.AUX_RGBS if rgb_s else 0) | (L.AUX_PC if pc else 0) | (L.AUX_SCALE_PCS if scale_pcs else 0) | \
            (L.AUX_DETACH_RGBS if detach_rgbs_scale else 0) | (L.AUX_SSIM if (ssim and rgb_s) else 0) | \
            (L.AUX_SHIFT_FIRST if (shift_first and (aff is not None or mats is not None)) else 0)
    out = _AuxTerms.apply(d1_img, d2_img, rel, scale2 if (scale_pcs and mats is None) else None, img1r, img2r, K, Kinv, int(res[0]), int(res[1]),
                          float(nearest_limit), flags, tuple(shard), aff, mats, weights)
    return out if weights is not None else out[:3]
