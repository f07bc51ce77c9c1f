"""TEST-ONLY stand-in for nnr.render_rays built on the CPU oracle, so the host-side logic of model.Renderer /
model.Trainer (ray generation, masks, sharding, loss scaling, all-reduce) can be exercised without a GPU.
Never imported by the product: tests monkeypatch `model.rendering.nnr.render_rays` with it."""
import torch

import nerf_oracle as orc
from nnr import LAYER_NAMES


def render_rays(pts_o, pts_d, view_d, z_lo, z_hi, jitter, weights, biases, *, hidden, dist_alpha, white_bg, relu_sigma, bf16=False, samples=True):
    R, N = pts_o.shape[0], z_lo.shape[0]
    params = {}
    for n, w, b in zip(LAYER_NAMES, weights, biases):
        params[n + ".weight"], params[n + ".bias"] = w, b
    z = z_lo.view(1, N).expand(R, N)
    if jitter is not None:
        z = z_lo + (z_hi - z_lo) * jitter.view(R, N)
    pts = (pts_o.unsqueeze(1) + pts_d.unsqueeze(1) * z.unsqueeze(-1)).reshape(-1, 3)
    view = view_d.unsqueeze(1).expand(R, N, 3).reshape(-1, 3)
    rgb, occ = orc.mlp(params, pts, view, dist_alpha=dist_alpha, occ_activation="relu" if relu_sigma else "softplus")
    rgb, alpha = rgb.view(R, N, 3), occ.view(R, N)
    if dist_alpha:
        delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), 1e10)], dim=-1)
        alpha = 1 - torch.exp(-1.0 * alpha * delta)
        alpha = torch.cat([alpha[:, :-1], torch.ones(R, 1)], dim=-1)
    trans = torch.cumprod(torch.cat([torch.ones(R, 1), 1.0 - alpha + orc.EPS_T], -1), -1)[:, :-1]
    w = alpha * trans
    out = (w.unsqueeze(-1) * rgb).sum(-2)
    dist = (w * z).sum(-1)
    if white_bg:
        out = out + (1.0 - w.sum(-1, keepdim=True))
    return out, dist, alpha.detach(), z.detach()
