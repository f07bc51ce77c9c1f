"""CPU oracle for the NoPe-NeRF volumetric-rendering hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``nope-nerf_amd/``) may
import this file; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker.

What it is: a plain-PyTorch (CPU, fp32) restatement of the reference algorithm
for the path named in BASELINE.json -- ray generation from a learnable pose,
stratified / NDC sampling, the positional-encoded MLP, alpha-compositing, and
the two loss heads that feed the backward.  Gradients come from stock autograd.
Every function cites the reference lines it restates (paths relative to the
upstream repository root).

Pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself: ``oracle/gen_golden.py``
imports the real reference modules on CPU, runs them on seeded inputs, asserts that
this restatement reproduces them, and freezes the tensors under ``tests/golden/``.
``tests/test_oracle_golden.py`` re-checks the oracle against those files on every run.

All functions are explicit about randomness: the stratified jitter ``u`` and the ray
indices are *inputs*, never drawn here, so the HIP path can be fed the identical values.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

EPS_T = 1e-6  # model/rendering.py:9  (epsilon inside the transmittance product)

# ----------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------

LAYER_NAMES = (
    "layers0.0", "layers0.2", "layers0.4", "layers0.6",
    "layers1.0", "layers1.2", "layers1.4", "layers1.6",
    "fc_density", "fc_feature", "rgb_layers.0", "fc_rgb",
)


def layer_shapes(hidden: int, pos_levels: int = 10, dir_levels: int = 4):
    """(out, in) of the 12 nn.Linear layers -- model/official_nerf.py:11-37."""
    p = (2 * pos_levels + 1) * 3
    q = (2 * dir_levels + 1) * 3
    d = hidden
    return {
        "layers0.0": (d, p), "layers0.2": (d, d), "layers0.4": (d, d), "layers0.6": (d, d),
        "layers1.0": (d, d + p), "layers1.2": (d, d), "layers1.4": (d, d), "layers1.6": (d, d),
        "fc_density": (1, d), "fc_feature": (d, d), "rgb_layers.0": (d // 2, d + q), "fc_rgb": (3, d // 2),
    }


def init_params(hidden: int, seed: int, white_bkgd: bool = False) -> Dict[str, torch.Tensor]:
    """nn.Linear default init (kaiming-uniform a=sqrt(5) == U(-1/sqrt(in), 1/sqrt(in)) for both
    weight and bias) plus the bias overrides of model/official_nerf.py:39-44.  Not bit-identical
    to constructing the reference module (RNG consumption order differs); used for synthetic
    benches.  Golden fixtures carry the reference module's own weights instead."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name, (o, i) in layer_shapes(hidden).items():
        bound = 1.0 / math.sqrt(i)
        out[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * bound
        out[name + ".bias"] = (torch.rand(o, generator=g) * 2 - 1) * bound
    out["fc_density.bias"] = torch.tensor([0.1])
    out["fc_rgb.bias"] = torch.full((3,), 0.8 if white_bkgd else 0.02)
    return out


# ----------------------------------------------------------------------------------------
# SO(3) exponential and pose  (model/common.py:277-330, model/poses.py:23-31)
# ----------------------------------------------------------------------------------------

def skew(v: torch.Tensor) -> torch.Tensor:
    """model/common.py:277-287."""
    z = torch.zeros(1, dtype=v.dtype)
    return torch.stack([
        torch.cat([z, -v[2:3], v[1:2]]),
        torch.cat([v[2:3], z, -v[0:1]]),
        torch.cat([-v[1:2], v[0:1], z]),
    ])


def so3_exp(r: torch.Tensor) -> torch.Tensor:
    """Rodrigues with theta = |r| + 1e-15 -- model/common.py:290-299."""
    k = skew(r)
    th = r.norm() + 1e-15
    return torch.eye(3, dtype=r.dtype) + (torch.sin(th) / th) * k + ((1 - torch.cos(th)) / th ** 2) * (k @ k)


def pose_c2w(r: torch.Tensor, t: torch.Tensor, init_c2w: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[[Exp(r), t], [0 0 0 1]] (optionally @ init) -- model/common.py:301-330, model/poses.py:23-31."""
    top = torch.cat([so3_exp(r), t.unsqueeze(1)], dim=1)
    c2w = torch.cat([top, torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=top.dtype)], dim=0)
    if init_c2w is not None:
        c2w = c2w @ init_c2w
    return c2w


def distortion(scales: torch.Tensor, shifts: torch.Tensor, cam: int, num_cams: int, fix_scale_n: bool = True):
    """model/distortions.py:19-26: clamp scale at 0.01 (to a constant), last camera's scale == 1."""
    s = scales[cam]
    if float(s) < 0.01:
        s = torch.tensor(0.01)
    if fix_scale_n and cam == num_cams - 1:
        s = torch.tensor(1)
    return s, shifts[cam]


# ----------------------------------------------------------------------------------------
# pixels, ray generation  (model/common.py:13-40,112-160,186-237)
# ----------------------------------------------------------------------------------------

def pixel_grid(h: int, w: int) -> torch.Tensor:
    """Scaled pixel centres in [-1,1], row-major flatten (idx = y*w + x) -- model/common.py:13-40."""
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    p = torch.stack([xs, ys], dim=-1).reshape(1, -1, 2).float()
    p[..., 0] = 2.0 * p[..., 0] / (w - 1) - 1.0
    p[..., 1] = 2.0 * p[..., 1] / (h - 1) - 1.0
    return p.to(torch.get_default_dtype())      # the fp32 values, whatever dtype the caller evaluates the rest in (fp64 ground truths)


def unproject(pixels: torch.Tensor, depth: torch.Tensor, camera_mat, world_mat, scale_mat) -> torch.Tensor:
    """p_world = S^-1 W^-1 K^-1 [x*d, y*d, d, 1] -- model/common.py:112-160 (three 4x4 inverses)."""
    ki, wi, si = torch.inverse(camera_mat), torch.inverse(world_mat), torch.inverse(scale_mat)
    ph = pixels.permute(0, 2, 1)
    ph = torch.cat([ph, torch.ones_like(ph)], dim=1)
    pd = ph.clone()
    pd[:, :3] = ph[:, :3] * depth.permute(0, 2, 1)
    return (si @ wi @ ki @ pd)[:, :3].permute(0, 2, 1)


def camera_origin(n: int, camera_mat, world_mat, scale_mat) -> torch.Tensor:
    """model/common.py:186-215."""
    p = torch.zeros(camera_mat.shape[0], 4, n)
    p[:, -1] = 1.0
    ki, wi, si = torch.inverse(camera_mat), torch.inverse(world_mat), torch.inverse(scale_mat)
    return (si @ wi @ ki @ p)[:, :3].permute(0, 2, 1)


def ndc_rays(fxfy, near, o, d):
    """model/common.py:632-675 (note -1/(1/f) == -f, and fxfy[1] = K11 < 0)."""
    t = -(near + o[..., 2]) / d[..., 2]
    o = o + t[..., None] * d
    ox, oy = o[..., 0] / o[..., 2], o[..., 1] / o[..., 2]
    o0 = -1.0 / (1 / fxfy[0]) * ox
    o1 = -1.0 / (1 / fxfy[1]) * oy
    o2 = 1.0 + 2.0 * near / o[..., 2]
    d0 = -1.0 / (1 / fxfy[0]) * (d[..., 0] / d[..., 2] - ox)
    d1 = -1.0 / (1 / fxfy[1]) * (d[..., 1] / d[..., 2] - oy)
    d2 = 1 - o2
    return torch.stack([o0, o1, o2], -1), torch.stack([d0, d1, d2], -1)


# ----------------------------------------------------------------------------------------
# MLP  (model/official_nerf.py:60-119)
# ----------------------------------------------------------------------------------------

def posenc(x: torch.Tensor, levels: int) -> torch.Tensor:
    """[x, sin(2^0 x), cos(2^0 x), ...] in 3-wide blocks -- model/official_nerf.py:99-119."""
    parts = [x]
    for i in range(levels):
        a = 2.0 ** i * x
        parts += [torch.sin(a), torch.cos(a)]
    return torch.cat(parts, dim=-1)


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    """fp32 -> bf16 (round to nearest even, what v_cvt_pk_bf16_f32 does) -> fp32."""
    return x.to(torch.bfloat16).to(torch.float32)


class _Bf16Matmul(torch.autograd.Function):
    """y = bf16(x) bf16(W)^T with fp32 accumulation -- the arithmetic of BASELINE configs[2] ("MFMA bf16 with fp32 accumulate") as
    the HIP kernels of the bf16 mode carry it out, forward AND backward: the input-gradient product rounds the incoming gradient
    and the weights, the weight-gradient product rounds the incoming gradient and the input (both kernels consume the very bf16
    values that were stored), the bias gradient sums the rounded gradient.  bf16 x bf16 products are exact in fp32, so this
    emulation differs from the MFMA only in the order of the fp32 additions."""

    @staticmethod
    def forward(ctx, x, w):
        xr, wr = bf16_round(x), bf16_round(w)
        ctx.save_for_backward(xr, wr)
        return xr @ wr.t()

    @staticmethod
    def backward(ctx, g):
        xr, wr = ctx.saved_tensors
        gr = bf16_round(g)
        return gr @ wr, gr.t() @ xr


class _Bf16Head(torch.autograd.Function):
    """The two narrow heads (1-row density, 3-row rgb) of the bf16 mode: the value is the bf16 x bf16 product with fp32 accumulation
    like every other layer (the forward kernel holds the activations only as bf16 and evaluates the heads as per-lane bf16 dot
    products); the INPUT gradient is g W in fp32 (3 / 1 FMAs per value on the VALU in the input-gradient kernel, from the unrounded
    gradient and weights); the weight gradient comes from the rounded operands like every other layer's."""

    @staticmethod
    def forward(ctx, x, w):
        xr, wr = bf16_round(x), bf16_round(w)
        ctx.save_for_backward(xr, w)
        return xr @ wr.t()

    @staticmethod
    def backward(ctx, g):
        xr, w = ctx.saved_tensors
        return g @ w, bf16_round(g).t() @ xr


class _RoundGrad(torch.autograd.Function):
    """Identity whose gradient is rounded to bf16: the bias gradients of the bf16 mode are sums of the ROUNDED pre-activation
    gradients (the weight-gradient kernel reads them from the bf16 planes)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return bf16_round(g)


def mlp_bf16(params: Dict[str, torch.Tensor], pts: torch.Tensor, viewdir: torch.Tensor, *, dist_alpha: bool,
             occ_activation: str = "softplus", pos_levels: int = 10, dir_levels: int = 4):
    """The MLP of `mlp` with every MFMA-shaped contraction in bf16 x bf16 -> fp32 (rendering.mfma_dtype: bf16), restating what
    the bf16 kernels compute (nnr_mlp_fwd_bf16.hip / nnr_mlp_dgrad_bf16.hip / nnr_wgrad_bf16.hip): all hidden layers incl. the two encoding
    inputs; the feature layer folded into the colour-hidden layer (W' = Wg[:, :D] Wf and b' = Wg[:, :D] bf + bg formed in fp32,
    then W' rounded); the 1-row density head and the 3-row rgb head as bf16 products too, their input gradients in fp32 (_Bf16Head).  The reference for the tight parity test of the bf16 mode; the fp32
    `mlp` stays the reference for how far bf16 arithmetic is from the reference's fp32."""
    W = lambda n: params[n + ".weight"]
    B = lambda n: params[n + ".bias"]

    def lin16(n, v):       # hidden layer: rounded product + fp32 bias; bias gradient = sum of the rounded gradient
        return _RoundGrad.apply(_Bf16Matmul.apply(v, W(n))) + B(n)

    def head(n, v):        # bf16 product; input gradient in fp32, weight gradient from rounded operands (_Bf16Head)
        return _Bf16Head.apply(v, W(n)) + _RoundGrad.apply(B(n).expand(v.shape[0], -1))

    e = posenc(pts, pos_levels)
    h = e
    for n in ("layers0.0", "layers0.2", "layers0.4", "layers0.6"):
        h = F.relu(lin16(n, h))
    h = torch.cat([h, e], dim=-1)
    for n in ("layers1.0", "layers1.2", "layers1.4", "layers1.6"):
        h = F.relu(lin16(n, h))
    raw = head("fc_density", h)
    occ = F.softplus(raw) if occ_activation == "softplus" else raw.relu()
    if not dist_alpha:
        occ = 1 - torch.exp(-1.0 * occ)
    D = W("fc_feature").shape[0]
    wg = W("rgb_layers.0")
    w_merged = wg[:, :D] @ W("fc_feature")                                # fp32, as the pack kernel forms it
    b_merged = wg[:, :D] @ B("fc_feature") + B("rgb_layers.0")
    pre = _RoundGrad.apply(_Bf16Matmul.apply(h, w_merged) + _Bf16Matmul.apply(posenc(viewdir, dir_levels), wg[:, D:])) + b_merged
    g = F.relu(pre)
    rgb = torch.sigmoid(head("fc_rgb", g))
    return rgb, occ


def mlp(params: Dict[str, torch.Tensor], pts: torch.Tensor, viewdir: torch.Tensor, *, dist_alpha: bool,
        occ_activation: str = "softplus", pos_levels: int = 10, dir_levels: int = 4):
    """OfficialStaticNerf.forward(return_addocc=True) -- model/official_nerf.py:60-96.
    Returns (rgb (S,3), occ (S,1)) where occ = alpha if not dist_alpha else sigma."""
    lin = lambda n, v: F.linear(v, params[n + ".weight"], params[n + ".bias"])
    e = posenc(pts, pos_levels)
    h = e
    for n in ("layers0.0", "layers0.2", "layers0.4", "layers0.6"):
        h = F.relu(lin(n, h))
    h = torch.cat([h, e], dim=-1)
    for n in ("layers1.0", "layers1.2", "layers1.4", "layers1.6"):
        h = F.relu(lin(n, h))
    raw = lin("fc_density", h)
    occ = F.softplus(raw) if occ_activation == "softplus" else raw.relu()
    if not dist_alpha:
        occ = 1 - torch.exp(-1.0 * occ)
    f = lin("fc_feature", h)
    g = F.relu(lin("rgb_layers.0", torch.cat([f, posenc(viewdir, dir_levels)], dim=-1)))
    rgb = torch.sigmoid(lin("fc_rgb", g))
    return rgb, occ


def density_gradient(params: Dict[str, torch.Tensor], p: torch.Tensor, pos_levels: int = 10) -> torch.Tensor:
    """OfficialStaticNerf.gradient -- model/official_nerf.py:46-67: -d(raw density)/dp as (S,1,3), with create_graph=True
    (the normal-consistency term differentiates it again)."""
    lin = lambda n, v: F.linear(v, params[n + ".weight"], params[n + ".bias"])
    with torch.enable_grad():
        if not p.requires_grad:
            p = p.requires_grad_(True)
        e = posenc(p, pos_levels)
        h = e
        for n in ("layers0.0", "layers0.2", "layers0.4", "layers0.6"):
            h = F.relu(lin(n, h))
        h = torch.cat([h, e], dim=-1)
        for n in ("layers1.0", "layers1.2", "layers1.4", "layers1.6"):
            h = F.relu(lin(n, h))
        y = lin("fc_density", h)
        g = torch.autograd.grad(outputs=y, inputs=p, grad_outputs=torch.ones_like(y), create_graph=True, retain_graph=True,
                                only_inputs=True, allow_unused=True)[0]
    return -g.unsqueeze(1)


def normal_consistency(params, cam2: torch.Tensor, ray2: torch.Tensor, d_gt0: torch.Tensor, obj_mask: torch.Tensor,
                       noise: torch.Tensor) -> torch.Tensor:
    """model/rendering.py:76-93,133-141: normalised density gradients at the surface points (camera + ray * depth distance, valid
    rays only) and at neighbours perturbed by (noise - 0.5) * 0.01; `noise` replaces torch.rand_like(surface_points) at :137."""
    surface = (cam2 + ray2 * d_gt0.unsqueeze(-1))[obj_mask]
    n = surface.shape[0]
    neigh = surface + (noise - 0.5) * 0.01
    g = density_gradient(params, torch.cat([surface, neigh], dim=0))
    normals = g[:, 0, :] / (g[:, 0, :].norm(2, dim=1).unsqueeze(-1) + 10 ** (-5))
    return torch.norm(normals[:n] - normals[n:], dim=-1)


# ----------------------------------------------------------------------------------------
# sampling + renderer  (model/rendering.py:36-197)
# ----------------------------------------------------------------------------------------

def sample_z(n_rays: int, n_samples: int, near: float, far: float, jitter: Optional[torch.Tensor]):
    """model/rendering.py:95-96,184-190: linspace, affine to [near,far], optional stratified jitter.
    jitter: (1, R, N) uniform [0,1) or None."""
    z = torch.linspace(0.0, 1.0, steps=n_samples).view(1, 1, -1).repeat(1, n_rays, 1)
    z = near * (1.0 - z) + far * z
    if jitter is not None:
        mid = 0.5 * (z[:, :, 1:] + z[:, :, :-1])
        hi = torch.cat([mid, z[:, :, -1:]], dim=-1)
        lo = torch.cat([z[:, :, :1], mid], dim=-1)
        z = lo + (hi - lo) * jitter
    return z


def render(params, pixels, depth, camera_mat, world_mat, scale_mat, cfg: dict, *,
           jitter: Optional[torch.Tensor] = None, eval_: bool = False, chunk: int = 64000,
           normal_noise: Optional[torch.Tensor] = None) -> dict:
    """Renderer.nope_nerf -- model/rendering.py:36-167.

    pixels (1,R,2), depth (1,R,1), matrices (1,4,4).  cfg keys as configs/default.yaml `rendering`
    plus `occ_activation`.  `jitter` replaces torch.rand at :189 (None == add_noise False); `normal_noise` (M,3) replaces
    torch.rand_like at :137 when cfg['normal_loss'] is on (the normal-consistency branch :133-143)."""
    n_samples = cfg["num_points"]
    dist_alpha = cfg["dist_alpha"]
    option = cfg["sample_option"]
    near, far = cfg["depth_range"]
    R = pixels.shape[1]

    cam = camera_origin(R, camera_mat, world_mat, scale_mat)                       # :54-56
    pw = unproject(pixels, depth, camera_mat, world_mat, scale_mat)                 # :57
    d_gt = torch.norm(pw - cam, p=2, dim=-1)                                        # :60
    pix_w = unproject(pixels, torch.ones(1, R, 1), camera_mat, world_mat, scale_mat)  # :63-65
    ray = pix_w - cam
    ray_norm = ray.norm(2, 2)
    if cfg["normalise_ray"]:
        ray = ray / ray.norm(2, 2).unsqueeze(-1)                                    # :69
    else:
        d_gt = d_gt / ray_norm                                                      # :71

    d_i = d_gt.clone()
    finite = (d_i.abs() != float("inf")) & ~torch.isnan(d_i)                        # common.py:60-72
    obj_mask = (finite & ~(d_i == 0))[0]                                            # :73-87

    cam2, ray2 = cam.reshape(-1, 3), ray.reshape(-1, 3)
    if option == "ndc":                                                             # :168-180
        focal = torch.cat([camera_mat[:, 0, 0], camera_mat[:, 1, 1]])
        o_s, d_s = ndc_rays(focal, 1.0, cam2, ray2)
        z = sample_z(R, n_samples, 0.0, 1.0, None)
    else:                                                                           # :182-197
        o_s, d_s = cam2, ray2
        z = sample_z(R, n_samples, float(torch.tensor(near)), float(torch.tensor(far)), jitter)
    pts = (o_s.unsqueeze(-2) + d_s.unsqueeze(-2) * z[0].unsqueeze(-1)).reshape(-1, 3)
    view = -1 * ray2.unsqueeze(-2).repeat(1, n_samples, 1).reshape(-1, 3)
    if not cfg.get("use_ray_dir", True):
        view = torch.ones_like(view)                                                # :104-105
    z = z.view(-1, n_samples, 1)

    rgbs, occs = [], []
    net = mlp_bf16 if str(cfg.get("mfma_dtype", "fp32")).lower() == "bf16" else mlp   # bf16: emulation of the bf16 kernels' arithmetic
    for i in range(0, pts.shape[0], chunk):                                         # :108-117
        c, a = net(params, pts[i:i + chunk], view[i:i + chunk], dist_alpha=dist_alpha,
                   occ_activation=cfg.get("occ_activation", "softplus"))
        rgbs.append(c)
        occs.append(a)
    rgb = torch.cat(rgbs).reshape(R, n_samples, 3)
    alpha = torch.cat(occs).view(R, n_samples)

    if dist_alpha:                                                                  # :121-128
        t = z.view(R, n_samples)
        delta = torch.cat([t[:, 1:] - t[:, :-1], torch.full((R, 1), 1e10)], dim=-1)
        alpha = 1 - torch.exp(-1.0 * alpha * delta)
        alpha[:, -1] = 1.0

    trans = torch.cumprod(torch.cat([torch.ones(R, 1), 1.0 - alpha + EPS_T], -1), -1)[:, :-1]   # :130
    w = alpha * trans
    rgb_out = torch.sum(w.unsqueeze(-1) * rgb, dim=-2)                              # :131
    dist = torch.sum(w.unsqueeze(-1) * z, dim=-2).squeeze(-1)                       # :132
    if cfg.get("white_background", False):
        rgb_out = rgb_out + (1.0 - torch.sum(w, -1).unsqueeze(-1))                  # :145-147

    d_gt0 = d_gt[0]
    normal = None
    if cfg.get("normal_loss", False) and not eval_:                                 # :133-143
        normal = normal_consistency(params, cam2, ray2, d_gt0, obj_mask, normal_noise)
    if eval_ and cfg["normalise_ray"]:                                              # :150-154
        dist = dist / ray_norm[0]
        d_gt0 = d_gt0 / ray_norm[0]
    depth_gt = d_gt0[obj_mask]
    if option == "ndc":
        depth_gt = 1 - 1 / depth_gt                                                 # :157-158
    return {
        "rgb": rgb_out.reshape(1, -1, 3),
        "z_vals": z.squeeze(-1),
        "normal": normal,
        "depth_pred": dist[obj_mask],
        "depth_gt": depth_gt,
        "alpha": alpha,
        "mask": obj_mask,
        "dist_dense": dist,
    }


def nearest_gather(depth_img: torch.Tensor, img_size, ray_idx: torch.Tensor) -> torch.Tensor:
    """nope_nerf.forward -- model/network.py:22-24: nearest resize to (h,w), flatten, gather."""
    d = F.interpolate(depth_img, img_size, mode="nearest").view(1, 1, -1).permute(0, 2, 1)
    return d[:, ray_idx]


# ----------------------------------------------------------------------------------------
# loss heads that feed the backward  (model/losses.py:27-32,59-64,196-202)
# ----------------------------------------------------------------------------------------

def loss_heads(out: dict, rgb_gt: torch.Tensor, rgb_weight=1.0, depth_weight=0.04, rgb_type="l1"):
    """rgb: L1|L2 *sum* / R (losses.py:27-32); depth: L1 sum / M (:59-64); weighted total (:196-202)."""
    diff = out["rgb"] - rgb_gt
    lrgb = (diff.abs().sum() if rgb_type == "l1" else (diff ** 2).sum()) / float(out["rgb"].shape[1])
    ldep = (out["depth_pred"] - out["depth_gt"]).abs().sum() / float(out["depth_pred"].shape[0])
    return rgb_weight * lrgb + depth_weight * ldep, lrgb, ldep


# ----------------------------------------------------------------------------------------
# point-cloud loss between a frame and its reference frame  (model/losses.py:114-148)
# ----------------------------------------------------------------------------------------

def closest_idx(pts_src: torch.Tensor, pts_des: torch.Tensor, split: int = 500000) -> torch.Tensor:
    """Loss.comp_closest_pts_idx_with_split -- model/losses.py:125-142: (3,S),(3,D) -> (S,) argmin over the dense distance matrix."""
    out = []
    for sec in torch.split(pts_src, split, dim=1):
        diff = sec[:, :, None] - pts_des[:, None, :]
        out.append(torch.argmin(torch.linalg.norm(diff, dim=0), dim=1))
    return torch.cat(out)


def point_point_error(xt: torch.Tensor, yt: torch.Tensor) -> torch.Tensor:
    """Loss.comp_point_point_error -- model/losses.py:143-148."""
    idx = closest_idx(xt, yt)
    return torch.mean(torch.linalg.norm(xt - yt[:, idx], dim=0))


def pc_loss(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """Loss.get_pc_loss, match_method 'dense' -- model/losses.py:114-121: x (1,S,3), y (1,D,3), symmetric sum."""
    xt, yt = x[0].permute(1, 0), y[0].permute(1, 0)
    return point_point_error(xt, yt) + point_point_error(yt, xt)


def mean_on_mask(diff: torch.Tensor, valid: torch.Tensor) -> torch.Tensor:
    """Loss.mean_on_mask -- model/losses.py:77-85."""
    mask = valid.expand_as(diff)
    if mask.sum() > 0:
        return diff[mask].sum() / mask.sum()
    return torch.tensor(0.0)


def project_to_cam(points: torch.Tensor, camera_mat: torch.Tensor):
    """model/common.py:436-457: K [p;1], perspective divide, |x|,|y| <= 1 mask."""
    ph = torch.cat([points, torch.ones_like(points[..., :1])], dim=-1).permute(0, 2, 1)
    q = (camera_mat @ ph)[:, :3].permute(0, 2, 1)
    xy = q[..., :2] / q[..., 2:]
    return xy, (xy.abs().max(dim=-1)[0] <= 1).unsqueeze(-1).bool()


def ssim_dissimilarity(x, y):
    """model/losses.py:222-252: the SSIM module of the reference, applied the way the reference calls it -- on 4-d tensors whatever
    their meaning.  losses.py:154 passes (1, hr, wr, 3) colour tensors, so the 3x3 reflect-padded average pool runs over the last two
    axes: (grid x, colour channel)."""
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    x, y = F.pad(x, (1, 1, 1, 1), mode="reflect"), F.pad(y, (1, 1, 1, 1), mode="reflect")
    pool = lambda t: F.avg_pool2d(t, 3, 1)
    mu_x, mu_y = pool(x), pool(y)
    sigma_x, sigma_y = pool(x ** 2) - mu_x ** 2, pool(y ** 2) - mu_y ** 2
    sigma_xy = pool(x * y) - mu_x * mu_y
    n = (2 * mu_x * mu_y + c1) * (2 * sigma_xy + c2)
    d = (mu_x ** 2 + mu_y ** 2 + c1) * (sigma_x + sigma_y + c2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def aux_scope(pose_r, pose_t, scales, shifts, cam: int, ref: int, camera_mat, depth_img, depth_ref_img, img, ref_img, *,
              pc_weight=1.0, rgb_s_weight=1.0, pc_ratio=4, nearest_limit=0.01, shift_first=False, detach_ref_img=True,
              scale_pcs=True, detach_rgbs_scale=False, with_ssim=False):
    """The per-image terms between a frame and its reference frame -- model/training.py:280-365 (inputs) and
    model/losses.py:114-157 (point-cloud and surface-reprojection losses; with_ssim: losses.py:153-155 + 222-252).  depth images (1,1,hd,wd) are the
    raw mono depths; returns (weighted sum, loss_pc, loss_rgb_s)."""
    num_cams = pose_r.shape[0]
    world_mat = torch.inverse(pose_c2w(pose_r[cam], pose_t[cam])).unsqueeze(0)         # :238
    sc, sh = distortion(scales, shifts, cam, num_cams)
    depth_input = (depth_img + sh) * sc if shift_first else depth_img * sc + sh         # :240-245
    c2w_ref = pose_c2w(pose_r[ref], pose_t[ref])                                        # :281
    sc_r, sh_r = distortion(scales, shifts, ref, num_cams)
    depth_ref = sc_r * (depth_ref_img + sh_r) if shift_first else sc_r * depth_ref_img + sh_r
    if detach_ref_img:                                                                  # :288-292
        c2w_ref, depth_ref = c2w_ref.detach(), depth_ref.detach()
        sc_r = sc_r.detach()
    ref_rt = torch.inverse(c2w_ref).unsqueeze(0)
    if cam < num_cams - 1:                                                              # :296-313
        d1, d2, img1, img2 = depth_input, depth_ref, img, ref_img
        rel = ref_rt @ torch.inverse(world_mat)
        scale2 = sc_r
    else:
        d1, d2, img1, img2 = depth_ref, depth_input, ref_img, img
        rel = world_mat @ torch.inverse(ref_rt)
        scale2 = sc
    r_rel, t_rel = rel[:, :3, :3], rel[:, :3, 3]
    res = (int(depth_img.shape[-2] / pc_ratio), int(depth_img.shape[-1] / pc_ratio))    # :315-316
    p_pc = pixel_grid(*res)
    d1 = F.interpolate(d1, res, mode="nearest")
    d2 = F.interpolate(d2, res, mode="nearest")
    d1[d1 < nearest_limit] = nearest_limit                                              # :320-321 (in place: no gradient there)
    d2[d2 < nearest_limit] = nearest_limit
    eye = torch.eye(4).unsqueeze(0)
    pc1 = unproject(p_pc, d1.view(1, -1, 1), camera_mat, eye, eye)                      # :322-323
    pc2 = unproject(p_pc, d2.view(1, -1, 1), camera_mat, eye, eye)
    loss_rgb_s = torch.tensor(0.0)
    if rgb_s_weight != 0.0:                                                             # :325-341
        i1 = F.interpolate(img1, res, mode="bilinear")
        i2 = F.interpolate(img2, res, mode="bilinear")
        sample = lambda im, p: F.grid_sample(im, p.unsqueeze(1), mode="bilinear", align_corners=True).squeeze(2).permute(0, 2, 1)
        rgb_pc1 = sample(i1, p_pc)
        src = pc1.detach().clone() if detach_rgbs_scale else pc1
        rot = src @ r_rel.transpose(1, 2) + t_rel
        behind = (-rot[:, :, 2:] < nearest_limit).expand_as(rot)
        rot[behind] = nearest_limit
        p_re, valid = project_to_cam(rot, camera_mat)
        rgb_proj = sample(i2, p_re)
        shape = (1, res[0], res[1])
        diff = (rgb_pc1.view(*shape, 3) - rgb_proj.view(*shape, 3)).abs().clamp(0, 1)   # losses.py:150-157
        if with_ssim:                                                                   # losses.py:153-155
            diff = 0.15 * diff + 0.85 * ssim_dissimilarity(rgb_pc1.view(*shape, 3), rgb_proj.view(*shape, 3))
        loss_rgb_s = mean_on_mask(diff, valid.view(*shape, 1))
    pc1 = pc1 @ r_rel.transpose(1, 2) + t_rel                                           # :353-358
    if scale_pcs:
        pc1, pc2 = pc1 / scale2, pc2 / scale2
    loss_pc = pc_loss(pc1, pc2) if pc_weight != 0.0 else torch.tensor(0.0)
    return pc_weight * loss_pc + rgb_s_weight * loss_rgb_s, loss_pc, loss_rgb_s


def train_step_scope(params, pose_r, pose_t, scales, shifts, cam: int, camera_mat, depth_img, img, img_size,
                     ray_idx, jitter, cfg: dict, *, rgb_weight=1.0, depth_weight=0.04, rgb_type="l1",
                     shift_first=False, normal_noise=None):
    """The slice of Trainer.compute_loss that is on the hot path -- model/training.py:235-274 + loss heads.
    All leaves (params, pose_r, pose_t, scales, shifts) may require grad.  Returns (loss, out)."""
    h, w = img_size
    num_cams = pose_r.shape[0]
    c2w = pose_c2w(pose_r[cam], pose_t[cam])
    world_mat = torch.inverse(c2w).unsqueeze(0)                                     # :238
    sc, sh = distortion(scales, shifts, cam, num_cams)
    d_img = (depth_img + sh) * sc if shift_first else depth_img * sc + sh           # :240-245
    rgb_gt = img.view(1, 3, h * w).permute(0, 2, 1)[:, ray_idx]                     # :258-259
    p = pixel_grid(h, w)[:, ray_idx]                                                # :260-261
    depth = nearest_gather(d_img, (h, w), ray_idx)
    out = render(params, p, depth, camera_mat, world_mat, torch.eye(4).unsqueeze(0), cfg, jitter=jitter, normal_noise=normal_noise)
    loss, lrgb, ldep = loss_heads(out, rgb_gt, rgb_weight, depth_weight, rgb_type)
    out["loss_rgb"], out["loss_depth"] = lrgb, ldep
    return loss, out
