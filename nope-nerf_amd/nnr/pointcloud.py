"""Point-cloud nearest neighbour + point-to-point error on the GPU: the HIP replacement of
Loss.comp_closest_pts_idx_with_split / comp_point_point_error (reference model/losses.py:125-148).  CUDA tensors only:
there is no CPU fallback here -- model/losses.py keeps the torch expression for CPU tensors."""

import torch

from . import lib as L


def _st():
    return L.stream()


def _rows(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("nnr.pointcloud needs CUDA tensors (no CPU fallback)")
    if t.dim() != 2 or t.shape[1] != 3:
        raise ValueError("expected an (n, 3) point cloud, got %s" % (tuple(t.shape),))
    return t.contiguous().float()


def nearest(src: torch.Tensor, dst: torch.Tensor):
    """For every row of src (S,3): index (int64) of the nearest row of dst (D,3) and the distance to it."""
    src, dst = _rows(src.detach()), _rows(dst.detach())
    S, D = src.shape[0], dst.shape[0]
    idx = torch.empty(S, dtype=torch.int64, device=src.device)
    dist = torch.empty(S, dtype=torch.float32, device=src.device)
    scratch = torch.empty(S, dtype=torch.int64, device=src.device)
    L.check(L.load().nnr_pc_nearest(L.ptr(src), L.ptr(dst), S, D, L.ptr(idx), L.ptr(dist), L.ptr(scratch), _st()), "nnr_pc_nearest")
    return idx, dist


class _PointPointError(torch.autograd.Function):
    """mean_s || src_s - dst_{nn(s)} ||; the match is not differentiated (it is an argmin), the distance is."""

    @staticmethod
    def forward(ctx, src, dst):
        s, d = _rows(src), _rows(dst)
        idx, dist = nearest(s, d)
        ctx.save_for_backward(s, d, idx, dist)
        return dist.mean()

    @staticmethod
    def backward(ctx, g):
        s, d, idx, dist = ctx.saved_tensors
        need_s, need_d = ctx.needs_input_grad
        g_s = torch.empty_like(s) if need_s else None
        g_d = torch.zeros_like(d) if need_d else None
        if need_s or need_d:
            g = g.contiguous().float()
            L.check(L.load().nnr_pc_error_bwd(L.ptr(s), L.ptr(d), L.ptr(idx), L.ptr(dist), L.ptr(g), s.shape[0], d.shape[0],
                                              L.ptr(g_s) if need_s else None, L.ptr(g_d) if need_d else None, _st()),
                    "nnr_pc_error_bwd")
        return g_s, g_d


def point_point_error(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """comp_point_point_error for (S,3) / (D,3) row-major clouds."""
    return _PointPointError.apply(src, dst)
