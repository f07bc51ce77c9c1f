"""Geometry helpers on the render path plus the small utilities the callers import from model.common.

Hot-path items and the reference lines they stand for:
  arange_pixels            model/common.py:13-40
  get_mask                 model/common.py:60-72
  pixel_to_world_matrix    the S^-1 W^-1 K^-1 chain used by origin_to_world / transform_to_world (:112-160, :186-215)
  transform_to_world, origin_to_world, image_points_to_world   same lines, public API kept
  get_ndc_rays_fxfy        model/common.py:632-675
  vec2skew / Exp / make_c2w / convert3x4_4x4                   model/common.py:277-330
Everything here is a handful of O(R) or O(1) torch ops per step; the per-sample work lives in the HIP kernels.
The novel-view path generators vis/render.py imports from here (spiral, slerp / B-spline interpolation of the learned poses,
common.py:333-404,511-615) live in model/trajectories.py and are re-exported below.
"""
import logging
import os
import shutil

import numpy as np
import torch

from model.trajectories import (create_spheric_poses, generate_spiral_nerf, get_poses_at_times, interp_poses,  # noqa: F401
                                interp_poses_bspline, interp_t, normalize, poses_avg, render_path_spiral, scipy_bspline,
                                viewmatrix)

logger_py = logging.getLogger(__name__)


# ---------------------------------------------------------------------------------------------------- pixels
def arange_pixels(resolution=(128, 128), batch_size=1, image_range=(-1., 1.), device=torch.device("cpu")):
    """Integer pixel locations (x, y) in row-major order and their coordinates scaled to image_range.
    Returns (B, h*w, 2) int64 and (B, h*w, 2) float32."""
    h, w = resolution
    ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    loc = torch.stack([xs, ys], dim=-1).reshape(1, h * w, 2).long().repeat(batch_size, 1, 1)
    span = image_range[1] - image_range[0]
    half = span / 2
    scaled = loc.float()
    scaled[..., 0] = span * scaled[..., 0] / (w - 1) - half
    scaled[..., 1] = span * scaled[..., 1] / (h - 1) - half
    return loc, scaled


def to_pytorch(tensor, return_type=False):
    is_numpy = isinstance(tensor, np.ndarray)
    out = torch.from_numpy(tensor) if is_numpy else tensor
    out = out.clone()
    return (out, is_numpy) if return_type else out


def get_mask(tensor):
    """True where the value is neither +-inf nor NaN."""
    t, is_numpy = to_pytorch(tensor, True)
    mask = torch.isfinite(t)
    return mask.numpy() if is_numpy else mask


# ---------------------------------------------------------------------------------------------------- unprojection
def pixel_to_world_matrix(camera_mat, world_mat, scale_mat, invert=True):
    """(B,4,4) matrix taking homogeneous pixel-space points [x*d, y*d, d, 1] to world space."""
    if invert:
        camera_mat, world_mat, scale_mat = (torch.inverse(m) for m in (camera_mat, world_mat, scale_mat))
    return scale_mat @ world_mat @ camera_mat


def _identity(device):
    return torch.eye(4, dtype=torch.float32, device=device).unsqueeze(0)


def transform_to_world(pixels, depth, camera_mat, world_mat=None, scale_mat=None, invert=True,
                       device=None):
    """Pixels (B,N,2) with depth (B,N,1) -> world points (B,N,3).  `device` is accepted for signature
    compatibility; identity defaults are created on the pixels' device (the reference defaults to cuda)."""
    assert pixels.shape[-1] == 2
    pixels, is_numpy = to_pytorch(pixels, True)
    depth, camera_mat = to_pytorch(depth), to_pytorch(camera_mat)
    world_mat = _identity(pixels.device) if world_mat is None else to_pytorch(world_mat)
    scale_mat = _identity(pixels.device) if scale_mat is None else to_pytorch(scale_mat)
    m = pixel_to_world_matrix(camera_mat, world_mat, scale_mat, invert)
    ones = torch.ones_like(depth)
    ph = torch.cat([pixels * depth, depth, ones], dim=-1)            # (B,N,4) = [x d, y d, d, 1]
    out = (m @ ph.transpose(1, 2))[:, :3].transpose(1, 2)
    return out.numpy() if is_numpy else out


def origin_to_world(n_points, camera_mat, world_mat, scale_mat, invert=True):
    """Camera centre in world coordinates, repeated to (B, n_points, 3)."""
    m = pixel_to_world_matrix(camera_mat, world_mat, scale_mat, invert)
    return m[:, :3, 3].unsqueeze(1).expand(-1, n_points, -1)


def image_points_to_world(image_points, camera_mat, world_mat, scale_mat, invert=True):
    """Points on the image plane (depth 1) to world coordinates."""
    assert image_points.shape[-1] == 2
    ones = torch.ones(*image_points.shape[:2], 1, device=image_points.device)
    return transform_to_world(image_points, ones, camera_mat, world_mat, scale_mat, invert=invert)


def transform_to_camera_space(p_world, camera_mat, world_mat, scale_mat):
    ph = torch.cat([p_world, torch.ones_like(p_world[..., :1])], dim=-1).transpose(1, 2)
    return (camera_mat @ world_mat @ scale_mat @ ph)[:, :3].transpose(1, 2)


def get_ndc_rays_fxfy(fxfy, near, rays_o, rays_d):
    """World rays -> NDC rays for forward-facing scenes; fxfy = (K00, K11) with K11 < 0."""
    t = -(near + rays_o[..., 2]) / rays_d[..., 2]
    rays_o = rays_o + t[..., None] * rays_d
    ox = rays_o[..., 0] / rays_o[..., 2]
    oy = rays_o[..., 1] / rays_o[..., 2]
    gx, gy = -1. / (1 / fxfy[0]), -1. / (1 / fxfy[1])
    o_ndc = torch.stack([gx * ox, gy * oy, 1. + 2. * near / rays_o[..., 2]], -1)
    d_ndc = torch.stack([gx * (rays_d[..., 0] / rays_d[..., 2] - ox),
                         gy * (rays_d[..., 1] / rays_d[..., 2] - oy),
                         1 - o_ndc[..., 2]], -1)
    return o_ndc, d_ndc


# ---------------------------------------------------------------------------------------------------- SO(3)
def vec2skew(v):
    z = torch.zeros(1, dtype=torch.float32, device=v.device)
    return torch.stack([torch.cat([z, -v[2:3], v[1:2]]),
                        torch.cat([v[2:3], z, -v[0:1]]),
                        torch.cat([-v[1:2], v[0:1], z])], dim=0)


def Exp(r):
    """Rodrigues; theta = |r| + 1e-15 keeps the map (and its gradient) finite at r = 0."""
    k = vec2skew(r)
    th = r.norm() + 1e-15
    eye = torch.eye(3, dtype=torch.float32, device=r.device)
    return eye + (torch.sin(th) / th) * k + ((1 - torch.cos(th)) / th ** 2) * (k @ k)


def convert3x4_4x4(input):
    """Append the [0 0 0 1] row to (3,4) / (N,3,4), torch or numpy."""
    if torch.is_tensor(input):
        if input.dim() == 3:
            out = torch.cat([input, torch.zeros_like(input[:, 0:1])], dim=1)
            out[:, 3, 3] = 1.0
            return out
        last = torch.tensor([[0, 0, 0, 1]], dtype=input.dtype, device=input.device)
        return torch.cat([input, last], dim=0)
    if input.ndim == 3:
        out = np.concatenate([input, np.zeros_like(input[:, 0:1])], axis=1)
        out[:, 3, 3] = 1.0
        return out
    return np.concatenate([input, np.array([[0, 0, 0, 1]], dtype=input.dtype)], axis=0)


def make_c2w(r, t):
    return convert3x4_4x4(torch.cat([Exp(r), t.unsqueeze(1)], dim=1))


# ---------------------------------------------------------------------------------------------------- sampling helpers used by the trainer's auxiliary losses
def get_tensor_values(tensor, p, mode='nearest', scale=True, detach=True, detach_p=True, align_corners=False):
    """grid_sample `tensor` (B,C,H,W) at p (B,N,2) -> (B,N,C)  (reference model/common.py:75-109)."""
    _, _, h, w = tensor.shape
    if detach_p:
        p = p.detach()
    if scale:
        p[:, :, 0] = 2. * p[:, :, 0] / w - 1
        p[:, :, 1] = 2. * p[:, :, 1] / h - 1
    vals = torch.nn.functional.grid_sample(tensor, p.unsqueeze(1), mode=mode, align_corners=align_corners).squeeze(2)
    if detach:
        vals = vals.detach()
    return vals.permute(0, 2, 1)


def project_to_cam(points, camera_mat, device):
    """Camera-space points (B,N,3) -> image plane xy (B,N,2) and in-frustum mask (reference model/common.py:436-457)."""
    ph = torch.cat([points, torch.ones_like(points[..., :1])], dim=-1).transpose(1, 2)
    q = (camera_mat @ ph)[:, :3].transpose(1, 2)
    xy = q[..., :2] / q[..., 2:]
    valid = (xy.abs().max(dim=-1)[0] <= 1).unsqueeze(-1).bool()
    return xy, valid


def mse2psnr(mse):
    return (-10.0 * np.log10(np.maximum(mse, 1e-10))).astype(np.float32)


def backup(out_dir, config):
    """Snapshot the run's config and sources under <out_dir>/backup (reference model/common.py:492-506: config,
    train.py, configs/default.yaml and the top-level files of ./model and ./dataloading).  Missing files are skipped.  Data parallel: rank 0 only."""
    from nnr import parallel
    if not parallel.is_writer():
        return
    dst = os.path.join(out_dir, 'backup')
    os.makedirs(dst, exist_ok=True)
    shutil.copyfile(config, os.path.join(dst, 'config.yaml'))
    for f in ('train.py', os.path.join('configs', 'default.yaml')):
        if os.path.isfile(f):
            shutil.copy(f, dst)
    for sub in ('model', 'dataloading'):
        if os.path.isdir(sub):
            os.makedirs(os.path.join(dst, sub), exist_ok=True)
            for f in os.listdir(sub):
                if os.path.isfile(os.path.join(sub, f)):
                    shutil.copy(os.path.join(sub, f), os.path.join(dst, sub))


def compute_errors(gt, pred):
    """Standard monocular-depth error metrics (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3) on numpy arrays."""
    ratio = np.maximum(gt / pred, pred / gt)
    a1, a2, a3 = ((ratio < 1.25 ** k).mean() for k in (1, 2, 3))
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    return np.mean(np.abs(gt - pred) / gt), np.mean((gt - pred) ** 2 / gt), rmse, rmse_log, a1, a2, a3
