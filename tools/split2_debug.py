"""Experiment driver (round 6): the two-term fp16 products (NNR_FP32_PRODUCTS=split2, csrc/nnr_split2.h) against the fp32-MFMA kernels and the
six-term bf16 products on one synthetic batch -- inference forward, training forward, backward (every gradient tensor).
    python tools/split2_debug.py R N D"""
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
grgb, gdist = torch.randn(R, 3, generator=g).to(dev) / R, torch.randn(R, generator=g).to(dev) / R


def run(kind):
    L.set_fp32_products(kind)
    w = [params[n + ".weight"].to(dev).requires_grad_(True) for n in L.LAYER_NAMES]
    b = [params[n + ".bias"].to(dev).requires_grad_(True) for n in L.LAYER_NAMES]
    o, d = o0.to(dev).requires_grad_(True), d0.to(dev).requires_grad_(True)
    v = d0.to(dev).requires_grad_(True)
    kw = dict(hidden=D, dist_alpha=False, white_bg=False, relu_sigma=False)
    with torch.no_grad():
        rgb0, dist0, _, _ = ops.render_rays(o, d, v, z_lo, z_hi, jit, w, b, **kw)
    torch.cuda.synchronize()
    rgb, dist, alpha, _ = ops.render_rays(o, d, v, z_lo, z_hi, jit, w, b, **kw)
    torch.cuda.synchronize()
    ((rgb * grgb).sum() + (dist * gdist).sum()).backward()
    torch.cuda.synchronize()
    out = dict(rgb_inf=rgb0.detach(), dist_inf=dist0.detach(), rgb=rgb.detach(), dist=dist.detach(), alpha=alpha.detach(), go=o.grad, gd=d.grad, gv=v.grad)
    for i, n in enumerate(L.LAYER_NAMES):
        out["w." + n] = w[i].grad
        out["b." + n] = b[i].grad
    return out


print("cfg", R, N, D, os.environ.get("NNR_LIB", "(product library)"), flush=True)
a = run("mfma")
rel = lambda x, y: float((x - y).abs().max()) / max(1e-30, float(y.abs().max()))
l2 = lambda x, y: float((x - y).double().norm() / max(1e-300, float(y.double().norm())))
for kind in ("split3", "split2"):
    b = run(kind)
    print("%s: training vs inference forward rgb %.2e dist %.2e" % (kind, rel(b["rgb"], b["rgb_inf"]), rel(b["dist"], b["dist_inf"])))
    bad = [k for k in a if not torch.isfinite(b[k]).all()]
    print("%s vs fp32 MFMA, max |diff| / max |ref| (relative L2): " % kind + "  ".join("%s %.1e (%.1e)" % (k, rel(b[k], a[k]), l2(b[k], a[k])) for k in a)
          + ("   NON-FINITE: %s" % bad if bad else ""), flush=True)
