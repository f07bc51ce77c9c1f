"""Oracle with every intermediate exposed: used by the GPU diagnostics / stage tests to localise a mismatch to one
kernel stage and one layer.  Built on oracle/nerf_oracle.py (same formulas), CPU fp32."""
import torch
import torch.nn.functional as F

import nerf_oracle as orc
from nnr import LAYER_NAMES


def keep(t):
    if t.requires_grad:
        t.retain_grad()


def traced_mlp(params, pts, view):
    """Returns dict of activations; pre-activations are leaves of interest (retain_grad)."""
    t = {}
    lin = lambda n, v: F.linear(v, params[n + ".weight"], params[n + ".bias"])
    e = orc.posenc(pts, 10)
    t["e"] = e
    h = e
    names = LAYER_NAMES[:8]
    for i, n in enumerate(names):
        if i == 4:
            h = torch.cat([h, e], dim=-1)
        pre = lin(n, h)
        keep(pre)
        t[f"pre{i + 1}"] = pre
        h = F.relu(pre)
        t[f"h{i + 1}"] = h
    raw = lin("fc_density", h)
    keep(raw)
    t["raw"] = raw
    f = lin("fc_feature", h)
    keep(f)
    t["f"] = f
    dirv = orc.posenc(view, 4)
    t["dir"] = dirv
    gpre = lin("rgb_layers.0", torch.cat([f, dirv], dim=-1))
    keep(gpre)
    t["gpre"] = gpre
    g = F.relu(gpre)
    t["g"] = g
    rgbpre = lin("fc_rgb", g)
    keep(rgbpre)
    t["rgbpre"] = rgbpre
    t["rgb"] = torch.sigmoid(rgbpre)
    return t


def traced_render(params, pts_o, pts_d, view_d, z_lo, z_hi, jitter, *, dist_alpha, white_bg, relu_sigma=False):
    """Same contract as nnr.render_rays, CPU, plus the trace."""
    R, N = pts_o.shape[0], z_lo.shape[0]
    z = z_lo.view(1, N).expand(R, N)
    if jitter is not None:
        z = z_lo + (z_hi - z_lo) * jitter.view(R, N)
    pts = (pts_o.unsqueeze(1) + pts_d.unsqueeze(1) * z.unsqueeze(-1)).reshape(-1, 3)
    keep(pts)
    view = view_d.unsqueeze(1).expand(R, N, 3).reshape(-1, 3)
    if view.requires_grad:
        view.retain_grad()
    t = traced_mlp(params, pts, view)
    t["pts"], t["view"], t["z"] = pts, view, z
    raw = t["raw"].view(R, N)
    sigma = raw.relu() if relu_sigma else F.softplus(raw)
    if dist_alpha:
        delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), 1e10)], dim=-1)
        alpha = 1 - torch.exp(-1.0 * sigma * delta)
        alpha = torch.cat([alpha[:, :-1], torch.ones(R, 1)], dim=-1)
    else:
        alpha = 1 - torch.exp(-1.0 * sigma)
    trans = torch.cumprod(torch.cat([torch.ones(R, 1), 1.0 - alpha + orc.EPS_T], -1), -1)[:, :-1]
    w = alpha * trans
    rgb = (w.unsqueeze(-1) * t["rgb"].view(R, N, 3)).sum(-2)
    dist = (w * z).sum(-1)
    if white_bg:
        rgb = rgb + (1.0 - w.sum(-1, keepdim=True))
    t["alpha"] = alpha
    return rgb, dist, t
