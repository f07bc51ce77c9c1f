"""autograd wrappers of the camera front end and the loss heads of libnnr.so (include/nnr.h, nnr_camera.hip):
one HIP launch each for SE(3) exp, 4x4 inverses, ray generation, the nearest-resize depth gather and the rgb/depth loss
heads -- the reference spends ~300 tiny ATen kernels (incl. four rocSOLVER LU inverses) per step on the same work."""
from __future__ import annotations


import torch

from . import lib as L


def _st():
    return L.stream()


def _f32(t):
    return t.detach().contiguous().float()


class _Se3Exp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, r_all, t_all, idx):
        r, t = _f32(r_all), _f32(t_all)
        c2w = torch.empty(4, 4, dtype=torch.float32, device=r.device)
        L.check(L.load().nnr_se3_exp_fwd(L.ptr(r), L.ptr(t), int(idx), L.ptr(c2w), _st()), "nnr_se3_exp_fwd")
        ctx.save_for_backward(r)
        ctx.idx = int(idx)
        return c2w

    @staticmethod
    def backward(ctx, g):
        (r,) = ctx.saved_tensors
        g = _f32(g)
        d_r, d_t = torch.empty_like(r), torch.empty_like(r)
        L.check(L.load().nnr_se3_exp_bwd(L.ptr(r), ctx.idx, r.shape[0], L.ptr(g), L.ptr(d_r), L.ptr(d_t), _st()),
                "nnr_se3_exp_bwd")
        return d_r, d_t, None


def se3_exp(r_all: torch.Tensor, t_all: torch.Tensor, idx: int) -> torch.Tensor:
    """(n_cams,3) axis-angle and translation tables -> 4x4 camera-to-world of camera idx."""
    return _Se3Exp.apply(r_all, t_all, idx)


class _Inv4(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        x = _f32(a).view(-1, 16)
        y = torch.empty_like(x)
        L.check(L.load().nnr_inv4_fwd(L.ptr(x), L.ptr(y), x.shape[0], _st()), "nnr_inv4_fwd")
        ctx.save_for_backward(y)
        return y.view(a.shape)

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        gg = _f32(g).view(-1, 16)
        d = torch.empty_like(y)
        L.check(L.load().nnr_inv4_bwd(L.ptr(y), L.ptr(gg), L.ptr(d), y.shape[0], _st()), "nnr_inv4_bwd")
        return d.view(g.shape)


def inverse4(a: torch.Tensor) -> torch.Tensor:
    """Inverse of (...,4,4) matrices (cofactor formula), differentiable."""
    return _Inv4.apply(a)


class _RaySetup(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pixels, depth, K, W, S, normalise, use_dir):
        pix = _f32(pixels).view(-1, 2)
        R, dev = pix.shape[0], pix.device
        dep = _f32(depth).view(-1) if depth is not None else None
        k, w, s = _f32(K).view(16), _f32(W).view(16), _f32(S).view(16)
        f = dict(dtype=torch.float32, device=dev)
        pts_o, dirs, view = torch.empty(R, 3, **f), torch.empty(R, 3, **f), torch.empty(R, 3, **f)
        norm, d_gt = torch.empty(R, **f), torch.empty(R, **f)
        mask = torch.empty(R, dtype=torch.bool, device=dev)
        L.check(L.load().nnr_ray_setup_fwd(L.ptr(pix), L.ptr(dep), L.ptr(k), L.ptr(w), L.ptr(s), R, int(normalise),
                                           int(use_dir), L.ptr(pts_o), L.ptr(dirs), L.ptr(view), L.ptr(norm), L.ptr(d_gt),
                                           L.ptr(mask), _st()), "nnr_ray_setup_fwd")
        ctx.save_for_backward(pix, dep, k, w, s)
        ctx.flags = (int(normalise), int(use_dir))
        ctx.shapes = (None if depth is None else depth.shape, K.shape, W.shape, S.shape)
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)     # outputs a step does not use arrive as None in backward, not as zero-filled tensors (a launch each)
        return pts_o, dirs, view, norm, d_gt, mask

    @staticmethod
    def backward(ctx, g_o, g_dir, g_view, g_norm, g_dgt, _gm):
        pix, dep, k, w, s = ctx.saved_tensors
        R, dev = pix.shape[0], pix.device
        opt = lambda g: _f32(g) if g is not None else None
        g_o, g_dir, g_view, g_norm, g_dgt = (opt(g) for g in (g_o, g_dir, g_view, g_norm, g_dgt))
        f = dict(dtype=torch.float32, device=dev)
        d_depth = torch.empty(R, **f) if dep is not None else None
        out = torch.empty(3 * 16 + 12, **f)
        dK, dW, dS, scratch = out[0:16], out[16:32], out[32:48], out[48:60]
        L.check(L.load().nnr_ray_setup_bwd(L.ptr(pix), L.ptr(dep), L.ptr(k), L.ptr(w), L.ptr(s), R, ctx.flags[0], ctx.flags[1],
                                           L.ptr(g_o), L.ptr(g_dir), L.ptr(g_view), L.ptr(g_norm), L.ptr(g_dgt), L.ptr(d_depth),
                                           L.ptr(dK), L.ptr(dW), L.ptr(dS), L.ptr(scratch), _st()), "nnr_ray_setup_bwd")
        dshape, kshape, wshape, sshape = ctx.shapes
        return (None, d_depth.view(dshape) if d_depth is not None else None, dK.view(kshape), dW.view(wshape),
                dS.view(sshape), None, None)


def ray_setup(pixels, depth, camera_mat, world_mat, scale_mat, normalise: bool, use_dir: bool):
    """pixels (1,R,2), depth (1,R,1)|None, three (1,4,4) matrices -> pts_o, dir, view (R,3), ray_norm, d_gt (R), mask (R) bool."""
    return _RaySetup.apply(pixels, depth, camera_mat, world_mat, scale_mat, normalise, use_dir)


class _DepthGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth_img, ray_idx, h, w):
        img = _f32(depth_img)
        hd, wd = img.shape[-2:]
        idx = ray_idx.detach().contiguous().long()
        R = idx.shape[0]
        out = torch.empty(1, R, 1, dtype=torch.float32, device=img.device)
        L.check(L.load().nnr_depth_gather_fwd(L.ptr(img), L.ptr(idx), L.ptr(out), R, int(h), int(w), hd, wd, _st()),
                "nnr_depth_gather_fwd")
        ctx.save_for_backward(idx)
        ctx.dims = (int(h), int(w), hd, wd, tuple(depth_img.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        h, w, hd, wd, shape = ctx.dims
        g = _f32(g).view(-1)
        g_img = torch.empty(shape, dtype=torch.float32, device=g.device)
        L.check(L.load().nnr_depth_gather_bwd(L.ptr(g), L.ptr(idx), L.ptr(g_img), idx.shape[0], h, w, hd, wd, _st()),
                "nnr_depth_gather_bwd")
        return g_img, None, None, None


def depth_gather(depth_img: torch.Tensor, ray_idx: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """(1,1,hd,wd) mono-depth map -> (1,R,1) values at the rays' pixels of the nearest-resized (h,w) map."""
    assert depth_img.shape[0] == 1 and depth_img.shape[1] == 1
    return _DepthGather.apply(depth_img, ray_idx, h, w)


_UNIT = {}


def register_unit_gradient(t: torch.Tensor) -> torch.Tensor:
    """Declare `t` -- a 0-dim float32 tensor holding exactly 1.0 that nobody writes to -- as a root gradient: a backward that receives THIS
    tensor (same storage, same version) as its upstream gradient skips the multiplication by it (model/training.py hands it to
    torch.autograd.backward every step; an add in between passes it through untouched)."""
    assert t.dim() == 0 and t.dtype == torch.float32 and float(t) == 1.0
    _UNIT[(t.device, t.data_ptr())] = (t, t._version)
    return t


def _is_unit(g: torch.Tensor) -> bool:
    hit = _UNIT.get((g.device, g.data_ptr()))
    return hit is not None and g.dim() == 0 and g.dtype == torch.float32 and hit[0]._version == hit[1]


class _RenderLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, rgb_gt, dist, d_gt, mask, r_total, m_total, w_rgb, w_depth, rgb_l2, ndc, detach_gt):
        a, b, c, d = _f32(rgb).view(-1, 3), _f32(rgb_gt).view(-1, 3), _f32(dist).view(-1), _f32(d_gt).view(-1)
        m = mask.detach().contiguous()
        R, dev = a.shape[0], a.device
        f = dict(dtype=torch.float32, device=dev)
        out = torch.empty(5, **f)
        flat = torch.empty(5 * R, **f)                       # the three gradient tensors in one allocation: scaled by one launch
        g_rgb, g_dist, g_dgt = flat[:3 * R].view(R, 3), flat[3 * R:4 * R], flat[4 * R:]
        m_dev = None
        if torch.is_tensor(m_total):                      # device scalar (data-parallel global count): no host sync
            m_dev = m_total.detach().float().reshape(1).contiguous()
            m_total = -1.0
        L.check(L.load().nnr_render_loss(L.ptr(a), L.ptr(b), L.ptr(c), L.ptr(d), L.ptr(m), R, float(r_total), float(m_total),
                                         float(w_rgb), float(w_depth), int(rgb_l2), int(ndc), int(detach_gt), L.ptr(m_dev), L.ptr(out),
                                         L.ptr(g_rgb), L.ptr(g_dist), L.ptr(g_dgt), _st()), "nnr_render_loss")
        ctx.save_for_backward(flat)
        ctx.shapes = (rgb.shape, dist.shape, d_gt.shape, R)
        aux = out[1:]
        ctx.mark_non_differentiable(aux)
        ctx.set_materialize_grads(False)                     # (no zeros(4) launch for the logged parts' absent gradient)
        return out[0], aux

    @staticmethod
    def backward(ctx, g, _ga):
        (flat,) = ctx.saved_tensors
        s0, s1, s2, R = ctx.shapes
        if g is None:
            return (None,) * 12
        # the root gradient of a training step is the trainer's cached constant 1 (register_unit_gradient): the kernel's gradients are the
        # answer as they stand; any other upstream gradient scales them, one launch for all three
        scaled = flat if _is_unit(g) else flat * g
        return (scaled[:3 * R].view(s0), None, scaled[3 * R:4 * R].view(s1), scaled[4 * R:].view(s2)) + (None,) * 8


def render_loss(rgb, rgb_gt, dist, d_gt, mask, *, r_total, m_total=-1.0, w_rgb, w_depth, rgb_l2=False, ndc=False,
                detach_gt=False):
    """Weighted rgb + depth loss heads on dense per-ray tensors + validity mask.
    Returns (loss, aux) with aux = [loss_rgb, loss_depth, l2_mean, n_valid]."""
    return _RenderLoss.apply(rgb, rgb_gt, dist, d_gt, mask, r_total, m_total, w_rgb, w_depth, rgb_l2, ndc, detach_gt)


def pixels_from_index(ray_idx: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """Flat pixel indices (R,) -> scaled pixel coordinates (1,R,2) in [-1,1] (no gradient)."""
    idx = ray_idx.detach().contiguous().long()
    out = torch.empty(1, idx.shape[0], 2, dtype=torch.float32, device=idx.device)
    L.check(L.load().nnr_pixels_from_index(L.ptr(idx), L.ptr(out), idx.shape[0], int(h), int(w), _st()), "nnr_pixels_from_index")
    return out


class _NdcRays(torch.autograd.Function):
    @staticmethod
    def forward(ctx, origin, ray, camera_mat, near):
        o, d, K = _f32(origin), _f32(ray), _f32(camera_mat)
        R = d.shape[0]
        o = o.expand(R, 3).contiguous()
        out_o, out_d = torch.empty(R, 3, dtype=torch.float32, device=d.device), torch.empty(R, 3, dtype=torch.float32, device=d.device)
        L.check(L.load().nnr_ndc_rays_fwd(L.ptr(o), L.ptr(d), L.ptr(K), float(near), L.ptr(out_o), L.ptr(out_d), R, _st()), "nnr_ndc_rays_fwd")
        ctx.save_for_backward(o, d, K)
        ctx.near, ctx.o_shape = float(near), tuple(origin.shape)
        return out_o, out_d

    @staticmethod
    def backward(ctx, g_o_ndc, g_d_ndc):
        o, d, K = ctx.saved_tensors
        R = d.shape[0]
        g_o, g_d = torch.empty_like(o), torch.empty_like(d)
        L.check(L.load().nnr_ndc_rays_bwd(L.ptr(o), L.ptr(d), L.ptr(K), ctx.near, L.ptr(_f32(g_o_ndc)), L.ptr(_f32(g_d_ndc)), L.ptr(g_o),
                                          L.ptr(g_d), R, _st()), "nnr_ndc_rays_bwd")
        if ctx.o_shape != tuple(g_o.shape):      # the origin was one point broadcast over the rays
            g_o = g_o.sum(0).reshape(ctx.o_shape) if len(ctx.o_shape) == 1 else g_o.sum(0, keepdim=True).expand(ctx.o_shape)
        return g_o, g_d, None, None


def ndc_rays(origin: torch.Tensor, ray: torch.Tensor, camera_mat: torch.Tensor, near: float = 1.0):
    """get_ndc_rays_fxfy (reference model/common.py:632-675) for (R,3) world rays in one launch each way; camera_mat (1,4,4) or
    (4,4) is a constant of the graph (the caller keeps the torch expression when the focal is being learned)."""
    return _NdcRays.apply(origin, ray, camera_mat.reshape(-1, 4, 4)[0], near)


class _DepthGatherAffine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth_img, ray_idx, scale, shift, h, w, shift_first):
        img, sc, sh = _f32(depth_img), _f32(scale).reshape(1), _f32(shift).reshape(1)
        hd, wd = img.shape[-2:]
        idx = ray_idx.detach().contiguous().long()
        R = idx.shape[0]
        out = torch.empty(1, R, 1, dtype=torch.float32, device=img.device)
        L.check(L.load().nnr_depth_gather_affine_fwd(L.ptr(img), L.ptr(idx), L.ptr(sc), L.ptr(sh), int(bool(shift_first)), L.ptr(out), R,
                                                    int(h), int(w), hd, wd, _st()), "nnr_depth_gather_affine_fwd")
        ctx.save_for_backward(img, idx, sc, sh)
        ctx.meta = (int(h), int(w), hd, wd, int(bool(shift_first)), tuple(scale.shape), tuple(shift.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        img, idx, sc, sh = ctx.saved_tensors
        h, w, hd, wd, shift_first, s_shape, h_shape = ctx.meta
        g_ss = torch.empty(2, dtype=torch.float32, device=img.device)
        L.check(L.load().nnr_depth_gather_affine_bwd(L.ptr(_f32(g).view(-1)), L.ptr(img), L.ptr(idx), L.ptr(sc), L.ptr(sh), shift_first,
                                                    L.ptr(g_ss), idx.shape[0], h, w, hd, wd, _st()), "nnr_depth_gather_affine_bwd")
        return None, None, g_ss[0].reshape(s_shape), g_ss[1].reshape(h_shape), None, None, None


def depth_gather_affine(depth_img, ray_idx, scale, shift, h: int, w: int, shift_first: bool = False) -> torch.Tensor:
    """depth_gather of the distorted map (depth * scale + shift, or (depth + shift) * scale) without distorting the map:
    (1,1,hd,wd) RAW mono depth + the frame's (1,) scale / shift -> (1,R,1); gradients reach scale and shift only."""
    assert depth_img.shape[0] == 1 and depth_img.shape[1] == 1
    return _DepthGatherAffine.apply(depth_img, ray_idx, scale, shift, h, w, shift_first)


_BWD_SCRATCH = {}


def _bwd_scratch(device):
    """The persistent scratch of nnr_step_rays_bwd (partial sums of its workgroups + their ticket counter): one per device and stream,
    zero-filled once -- every call leaves the counter zero (no memset launch per step)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    buf = _BWD_SCRATCH.get(key)
    if buf is None:
        buf = _BWD_SCRATCH[key] = torch.zeros(L.STEP_BWD_SCRATCH_FLOATS, dtype=torch.float32, device=device)
    return buf


class _StepRays(torch.autograd.Function):
    """The fused front end of a training step (nnr_step_rays_fwd / _bwd): pose tables -> world_mat, depth-distortion tables -> the
    frame's scale / shift, pixel coordinates, distorted mono depth and colour targets of the picked pixels, ray generation."""

    @staticmethod
    def forward(ctx, r_all, t_all, scales, shifts, depth_img, img, ray_idx, K, S, meta):
        import ctypes as C
        cam, h, w, fix_last, shift_first, normalise, use_dir, ref, detach_ref = meta
        r, t, sc, sh = _f32(r_all), _f32(t_all), _f32(scales).reshape(-1), _f32(shifts).reshape(-1)
        dimg = _f32(depth_img)
        hd, wd = dimg.shape[-2:]
        idx = ray_idx.detach().contiguous().long()
        R, dev = idx.shape[0], r.device
        k, s = _f32(K).reshape(16), _f32(S).reshape(16)
        im = _f32(img).reshape(3, h * w) if img is not None else None
        cfg = L.StepCfg(R, int(h), int(w), int(hd), int(wd), int(cam), int(r.shape[0]),
                        (L.STEP_NORMALISE if normalise else 0) | (L.STEP_USE_DIR if use_dir else 0) |
                        (L.STEP_SHIFT_FIRST if shift_first else 0) | (L.STEP_FIX_LAST_SCALE if fix_last else 0) |
                        (L.STEP_DETACH_REF if detach_ref else 0), int(ref))
        f = dict(dtype=torch.float32, device=dev)
        # one allocation for the per-ray float outputs: pts_o, dir, view, rgb_gt (3 each), pixels (2), ray_norm, d_gt (1 each)
        flat = torch.empty(R * 16, **f)
        pts_o, dirs, view, rgb_gt = (flat[3 * R * j: 3 * R * (j + 1)].view(R, 3) for j in range(4))
        pixels = flat[12 * R: 14 * R].view(1, R, 2)
        norm, d_gt = flat[14 * R: 15 * R], flat[15 * R: 16 * R]
        mask = torch.empty(R, dtype=torch.bool, device=dev)
        mats = torch.empty(56 if ref >= 0 else 34, **f)
        L.check(L.load().nnr_step_rays_fwd(C.byref(cfg), L.ptr(r), L.ptr(t), L.ptr(sc), L.ptr(sh), L.ptr(k), L.ptr(s), L.ptr(idx),
                                           L.ptr(dimg), L.ptr(im), L.ptr(pts_o), L.ptr(dirs), L.ptr(view), L.ptr(norm), L.ptr(d_gt),
                                           L.ptr(mask), L.ptr(rgb_gt) if im is not None else None, L.ptr(pixels), L.ptr(mats), _st()),
                "nnr_step_rays_fwd")
        ctx.save_for_backward(r, t, sc, sh, k, s, idx, dimg)
        ctx.cfg = cfg
        ctx.shapes = (tuple(r_all.shape), tuple(t_all.shape), tuple(scales.shape), tuple(shifts.shape))
        # the gauge (model/distortions.py: the last camera's scale is torch.ones_like): the scale table is not part of that step's
        # graph, its .grad stays None and Adam skips it -- a zero-filled gradient would still move it by momentum
        ctx.gauge = (bool(fix_last) and int(cam) == int(r.shape[0]) - 1 and int(sc.shape[0]) == int(r.shape[0])
                     and (ref < 0 or detach_ref))      # (a live reference camera keeps the scale table in the graph through its own row)
        if ref >= 0:      # the pair entries of mats (rel, the two distortions, scale2) carry the per-image losses' gradients back
            ctx.mark_non_differentiable(mask, rgb_gt, pixels)
        else:
            ctx.mark_non_differentiable(mask, rgb_gt, pixels, mats)
        ctx.set_materialize_grads(False)
        return pts_o, dirs, view, norm, d_gt, mask, rgb_gt, pixels, mats

    @staticmethod
    def backward(ctx, g_o, g_dir, g_view, g_norm, g_dgt, _gm, _gc, _gp, _gmat):
        import ctypes as C
        r, t, sc, sh, k, s, idx, dimg = ctx.saved_tensors
        opt = lambda g: _f32(g) if g is not None else None
        g_o, g_dir, g_view, g_norm, g_dgt, g_mats = (opt(g) for g in (g_o, g_dir, g_view, g_norm, g_dgt, _gmat))
        n = r.shape[0]
        out = torch.empty(8 * n, dtype=torch.float32, device=r.device)
        d_r, d_t, d_sc, d_sh = out[:3 * n], out[3 * n:6 * n], out[6 * n:7 * n], out[7 * n:]
        L.check(L.load().nnr_step_rays_bwd(C.byref(ctx.cfg), L.ptr(r), L.ptr(t), L.ptr(sc), L.ptr(sh), L.ptr(k), L.ptr(s), L.ptr(idx),
                                           L.ptr(dimg), L.ptr(g_o), L.ptr(g_dir), L.ptr(g_view), L.ptr(g_norm), L.ptr(g_dgt),
                                           L.ptr(g_mats) if ctx.cfg.ref >= 0 else None,
                                           L.ptr(d_r), L.ptr(d_t), L.ptr(d_sc), L.ptr(d_sh), L.ptr(_bwd_scratch(r.device)), _st()), "nnr_step_rays_bwd")
        rs, ts, ss, hs = ctx.shapes
        return d_r.view(rs), d_t.view(ts), (None if ctx.gauge else d_sc.view(ss)), d_sh.view(hs), None, None, None, None, None, None


def step_rays(r_all, t_all, scales, shifts, depth_img, img, ray_idx, camera_mat, scale_mat, *, cam: int, h: int, w: int,
              fix_last_scale: bool, shift_first: bool, normalise: bool, use_dir: bool, ref: int = -1, detach_ref: bool = True):
    """Fused front end of a training step, one launch each way.  (n_cams,3) pose tables, (n_cams,1) distortion tables, the RAW
    (1,1,hd,wd) mono-depth map, the (1,3,h,w) frame (or None), (R,) pixel indices, (1,4,4) camera / scale matrices (constants)
    -> pts_o, dir, view (R,3), ray_norm, d_gt (R), mask (R) bool, rgb_gt (R,3), pixels (1,R,2), mats (34: c2w, world_mat, scale,
    shift).  Differentiable with respect to the four tables.
    ref >= 0: the frame pair of the per-image losses rides along -- mats has 56 entries, [34:50] the relative transform, [50:54] the two
    clouds' distortions (scale1, shift1, scale2, shift2), [54] scale2 -- and gradients arriving at those entries are chained into the
    tables by the same backward launch (detach_ref = training.detach_ref_img: none into the reference camera's rows)."""
    return _StepRays.apply(r_all, t_all, scales, shifts, depth_img, img, ray_idx, camera_mat, scale_mat,
                           (int(cam), int(h), int(w), bool(fix_last_scale), bool(shift_first), bool(normalise), bool(use_dir), int(ref),
                            bool(detach_ref)))
