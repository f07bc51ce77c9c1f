"""Experiment driver (round 3): the three-term fp32 mode against the fp32-MFMA mode on one synthetic batch -- inference forward, training
forward, backward; NNR_LIB selects an experiment library (csrc/build.py --split-variant)."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "nope-nerf_amd")]
from nnr import lib as L, ops          # noqa: E402
import nerf_oracle as orc               # noqa: E402

R, N, D = (int(x) for x in sys.argv[1:4])
dev = torch.device("cuda")
g = torch.Generator().manual_seed(5)
params = orc.init_params(D, 9)
o0 = 0.1 * torch.randn(R, 3, generator=g)
d0 = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
z = torch.linspace(0.1, 4.0, N + 1)
z_lo, z_hi = z[:-1].contiguous().to(dev), z[1:].contiguous().to(dev)
jit = torch.rand(R, N, generator=g).to(dev)


def run(kind):
    L.set_fp32_products(kind)
    w = [params[n + ".weight"].to(dev).requires_grad_(True) for n in L.LAYER_NAMES]
    b = [params[n + ".bias"].to(dev).requires_grad_(True) for n in L.LAYER_NAMES]
    o, d = o0.to(dev).requires_grad_(True), d0.to(dev).requires_grad_(True)
    v = d0.to(dev).requires_grad_(True)
    kw = dict(hidden=D, dist_alpha=False, white_bg=False, relu_sigma=False)
    with torch.no_grad():
        rgb0, _, _, _ = ops.render_rays(o, d, v, z_lo, z_hi, jit, w, b, **kw)
    torch.cuda.synchronize()
    rgb, dist, alpha, _ = ops.render_rays(o, d, v, z_lo, z_hi, jit, w, b, **kw)
    torch.cuda.synchronize()
    (rgb.sum() + dist.sum()).backward()
    torch.cuda.synchronize()
    return dict(rgb_inf=rgb0.detach(), rgb=rgb.detach(), alpha=alpha.detach(), gw0=w[0].grad, gw4=w[4].grad, gw7=w[7].grad, gw10=w[10].grad,
                gw11=w[11].grad, go=o.grad, gv=v.grad)


print("cfg", R, N, D, os.environ.get("NNR_LIB", "(product library)"), flush=True)
a = run("mfma")
b = run("split3")
rel = lambda x, y: float((x - y).abs().max()) / max(1e-30, float(y.abs().max()))
print("fp32 MFMA: training vs inference forward rgb  %.2e" % rel(a["rgb"], a["rgb_inf"]))
print("split3   : training vs inference forward rgb  %.2e" % rel(b["rgb"], b["rgb_inf"]))
print("split3 vs fp32 MFMA (max |diff| / max |ref|): " + "  ".join("%s %.2e" % (k, rel(b[k], a[k])) for k in a), flush=True)
