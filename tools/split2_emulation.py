"""CPU study (no GPU): would fp32 products taken as THREE fp16 MFMA terms of two-term operands pass the fp64 yardstick?

VERDICT r05 item 1, gates (i) / (ii) before any kernel is written.  x = x_h + x_m with x_h = fp16(x), x_m = fp16 of what is left; the
products hh + hm + mh, fp32 accumulation in MFMA order (16 products summed exactly, one rounding per 16).  Variants:

  exact     products exact, fp32 accumulation per 16 (what the six-term bf16 kernels deliver, the dropped terms are < 2^-24)
  split2    two-term fp16 operands: weights scaled per tensor so that max |w| sits in [2^13, 2^14); activations / gradients scaled per
            SAMPLE so that the sample's max sits in [2^3, 2^4); residual term plain: x_m = fp16(x - x_h) (goes subnormal for small x)
  split2r   the same with the residual carried at 2^11: x_m' = fp16((x - x_h) 2^11) against w_hs = fp16(w 2^-11) -- the residual keeps 11
            bits down to the smallest normal x_h
  split2u   forward activations UNSCALED (s = 1), residual at 2^11 (the cheap forward: no per-sample scale)

The whole Trainer-scope step of tests/test_gpu_bench_shape_parity.py at 256 x 64 is evaluated in fp64, by the fp32 CPU oracle (the
yardstick: == the reference bit for bit) and with every MFMA-shaped product of forward and input-gradient chain emulated (the weight
gradients stay exact products: the weight-gradient kernel is a separate decision).  Prints, per variant, what tests/test_gpu_split3.py
asserts: geometric mean over all tensors and seeds of (variant error / CPU-fp32 error) in relative L2, the worst tensor's geometric
mean, the worst per-tensor median.

    python tools/split2_emulation.py [--seeds 12] [--D 256]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("tests", "oracle", "nope-nerf_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))

import golden_util as gu          # noqa: E402
import nerf_oracle as orc         # noqa: E402

H, W = 270, 480


def f16(x):
    return x.to(torch.float16).to(torch.float64)


def pow2_scale(m, top):
    """power of two s with m s in [2^(top-1), 2^top) (m > 0)"""
    e = torch.ceil(torch.log2(m.clamp_min(1e-300)))
    e = torch.where(2.0 ** e == m, e + 1, e)
    return 2.0 ** (top - e)


def terms(x64, scale, residual_shift):
    """the two fp16 terms of x scale (as float64 values); the second carried at 2^residual_shift"""
    xs = x64 * scale
    h = f16(xs)
    m = f16((xs - h) * 2.0 ** residual_shift)
    return h, m


def mfma_sum(parts, K):
    """sum_k of the given (S, K, M)-shaped product streams accumulated like the MFMA: groups of 16 k exact, fp32 rounding per group"""
    raise NotImplementedError


def emu_matmul(x, w, kind, per_sample=True):
    """x (S, K) @ w (M, K)^T -> (S, M) float32, products per `kind`; fp32 accumulation in groups of 16 k"""
    x64, w64 = x.double(), w.double()
    S, K = x64.shape
    pad = (-K) % 16
    if pad:
        x64 = F.pad(x64, (0, pad))
        w64 = F.pad(w64, (0, pad))
    acc = torch.zeros(S, w64.shape[0], dtype=torch.float32)
    if kind == "exact":
        for k0 in range(0, x64.shape[1], 16):
            acc = (acc.double() + x64[:, k0:k0 + 16] @ w64[:, k0:k0 + 16].t()).float()
        return acc
    shift = 0 if kind == "split2" else 11
    sw = pow2_scale(w64.abs().max(), 14)
    if kind == "split2u" or not per_sample:
        sx = torch.ones(S, 1, dtype=torch.float64)
    else:
        sx = pow2_scale(x64.abs().amax(dim=1, keepdim=True).clamp_min(1e-300), 4)
    xh, xm = terms(x64, sx, shift)
    wh, wm = terms(w64, sw, 0)
    whs = f16(w64 * sw * 2.0 ** -shift) if shift else wh
    for k0 in range(0, x64.shape[1], 16):
        sl = slice(k0, k0 + 16)
        # the three MFMAs of a row, small products first; each accumulates into fp32
        for a, b in ((xh[:, sl], wm[:, sl]), (xm[:, sl], whs[:, sl]), (xh[:, sl], wh[:, sl])):
            acc = (acc.double() + a @ b.t()).float()
    return (acc.double() / (sw * sx)).float()


class EmuLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, kind_f, kind_b):
        ctx.save_for_backward(x, w)
        ctx.kind_b = kind_b
        return emu_matmul(x, w, kind_f)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        dx = emu_matmul(g, w.t().contiguous(), ctx.kind_b)
        dw = (g.double().t() @ x.double()).float()          # weight gradient: exact products (its kernel is a separate decision)
        return dx, dw, None, None


def make_mlp(kind_f, kind_b):
    def lin(params, n, v, narrow=False):
        w, b = params[n + ".weight"], params[n + ".bias"]
        if narrow:      # the 1-row / 3-row heads are per-lane fp32 dot products in the kernels
            return F.linear(v, w, b)
        return EmuLinear.apply(v, w, kind_f, kind_b) + b

    def mlp(params, pts, viewdir, *, dist_alpha, occ_activation="softplus", pos_levels=10, dir_levels=4):
        e = orc.posenc(pts, pos_levels)
        h = e
        for n in ("layers0.0", "layers0.2", "layers0.4", "layers0.6"):
            h = F.relu(lin(params, n, h))
        h = torch.cat([h, e], dim=-1)
        for n in ("layers1.0", "layers1.2", "layers1.4", "layers1.6"):
            h = F.relu(lin(params, n, h))
        raw = lin(params, "fc_density", h, narrow=True)
        occ = F.softplus(raw) if occ_activation == "softplus" else raw.relu()
        if not dist_alpha:
            occ = 1 - torch.exp(-1.0 * occ)
        f = lin(params, "fc_feature", h)
        g = F.relu(lin(params, "rgb_layers.0", torch.cat([f, orc.posenc(viewdir, dir_levels)], dim=-1)))
        rgb = torch.sigmoid(lin(params, "fc_rgb", g, narrow=True))
        return rgb, occ
    return mlp


def case_of(R, N, D, seed):
    g = torch.Generator().manual_seed(seed)
    f = 0.7 * W
    K = torch.diag(torch.tensor([2 * f / W, -2 * f / H, -1.0, 1.0])).unsqueeze(0)
    case = {
        "cfg.hidden": D, "cfg.N": N, "cfg.dist_alpha": 0, "cfg.ndc": 0, "cfg.near": 0.01, "cfg.far": 10.0, "cfg.normalise_ray": 1,
        "cfg.white": 0, "cfg.h": H, "cfg.w": W, "cfg.cam": 1, "cfg.eval": 0,
        "in.K": K.numpy(), "in.pose_r": (0.01 * torch.randn(gu.N_CAMS, 3, generator=g)).numpy(),
        "in.pose_t": (0.01 * torch.randn(gu.N_CAMS, 3, generator=g)).numpy(),
        "in.scales": (1 + 0.05 * torch.randn(gu.N_CAMS, 1, generator=g)).numpy(),
        "in.shifts": (0.05 * torch.randn(gu.N_CAMS, 1, generator=g)).numpy(),
        "in.depth_img": (1 + 2 * torch.rand(1, 1, H, W, generator=g)).numpy(),
        "in.img": torch.rand(1, 3, H, W, generator=g).numpy(),
        "in.ray_idx": torch.randperm(H * W, generator=g)[:R].numpy(),
        "in.jitter": torch.rand(1, R, N, generator=g).numpy(),
    }
    case["weights"] = orc.init_params(D, seed + 1)
    return case


def run(case, dtype, net=None):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    saved = orc.mlp
    try:
        if net is not None:
            orc.mlp = net
        t = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in gu.tensors(case).items()}
        cfg = gu.render_cfg(case)
        params = {k: v.to(dtype).clone().requires_grad_(True) for k, v in case["weights"].items()}
        leaves = {k: t[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
        loss, out = orc.train_step_scope(params, leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], int(case["cfg.cam"]),
                                         t["K"], t["depth_img"], t["img"], (H, W), t["ray_idx"], t["jitter"], cfg)
        loss.backward()
    finally:
        orc.mlp = saved
        torch.set_default_dtype(prev)
    res = {"out." + k: out[k].detach().double().numpy() for k in ("rgb", "depth_pred", "alpha")}
    res.update({"w." + k: v.grad.double().numpy() for k, v in params.items()})
    res.update({k: v.grad.double().numpy() for k, v in leaves.items()})
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--D", type=int, default=256)
    ap.add_argument("--R", type=int, default=256)
    ap.add_argument("--N", type=int, default=64)
    ap.add_argument("--variants", default="exact,split2,split2r,split2u")
    a = ap.parse_args()
    variants = {"exact": ("exact", "exact"), "split2": ("split2", "split2"), "split2r": ("split2r", "split2r"),
                "split2u": ("split2u", "split2r")}
    variants = {k: variants[k] for k in a.variants.split(",")}
    ratios = {v: {} for v in variants}
    for i in range(a.seeds):
        seed = (77 + a.D) if i == 0 else 177 + a.D + 100 * (i - 1)
        case = case_of(a.R, a.N, a.D, seed)
        ref = run(case, torch.float64)
        cpu = run(case, torch.float32)
        l2c = {k: gu.rel_l2(cpu[k], ref[k]) for k in ref}
        line = "seed %d: cpu mean %.2e" % (seed, np.mean(list(l2c.values())))
        for v, (kf, kb) in variants.items():
            got = run(case, torch.float32, make_mlp(kf, kb))
            l2v = {k: gu.rel_l2(got[k], ref[k]) for k in ref}
            for k in ref:
                ratios[v].setdefault(k, []).append(l2v[k] / max(l2c[k], 1e-6))
            line += " | %s mean %.2e" % (v, np.mean(list(l2v.values())))
        print(line, flush=True)
    gm = lambda v: float(np.exp(np.mean(np.log(np.maximum(np.asarray(v, dtype=np.float64), 1e-12)))))
    print("\nD=%d %dx%d, %d seeds; ratio = variant rel-L2 vs fp64 / CPU-fp32 rel-L2 vs fp64 (bars of tests/test_gpu_split3.py: overall <= 1.5, "
          "worst tensor <= 5, worst median <= 2.5)" % (a.D, a.R, a.N, a.seeds))
    for v in variants:
        per = {k: gm(x) for k, x in ratios[v].items()}
        med = {k: float(np.median(x)) for k, x in ratios[v].items()}
        overall = gm([x for xs in ratios[v].values() for x in xs])
        wk, wm = max(per.items(), key=lambda kv: kv[1]), max(med.items(), key=lambda kv: kv[1])
        print("%-8s overall %.3f | worst tensor (geo-mean) %.2f %s | worst median %.2f %s" % (v, overall, wk[1], wk[0], wm[1], wm[0]))
        outs = {k: gm(x) for k, x in ratios[v].items() if k.startswith("out.")}
        print("         outputs: " + ", ".join("%s %.2f" % kv for kv in outs.items()))


if __name__ == "__main__":
    main()
