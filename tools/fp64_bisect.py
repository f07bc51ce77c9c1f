"""Where does the HIP fp32 path leave fp64 further behind than CPU fp32 does?   python tools/fp64_bisect.py [D R N seed]

VERDICT r03 (weak 2): at D = 256 the whole-step comparison against an fp64 evaluation of the oracle has the HIP kernels (both product
modes) at worst 1.66e-2 / mean 3.0e-3 of the tensors' scale where the CPU fp32 oracle sits at 5.3e-3 / 6.3e-4 -- a max-abs metric that
single ReLU-gate flips dominate.  This tool walks the render operator STAGE BY STAGE (gpu_diag.py's harness: every workspace plane
through the C ABI) and prints, per stage, the error against the fp64 trace of three fp32 evaluations: HIP three-term, HIP fp32-MFMA, CPU
fp32 (the oracle) -- as relative L2 (flip-insensitive) and as max-abs / max|ref|, plus the number of ReLU gates that differ from fp64's.
The first stage where a HIP column leaves the CPU column is where the gap is made.  Encodings are additionally compared in ulps.

Writes nothing; run it on the GPU box and keep the output under profiles/."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("nope-nerf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import numpy as np
import torch

import nerf_oracle as orc
import trace_util
from nnr import lib as L
from nnr import ops


def trace(params, pts_o, pts_d, view, z_lo, z_hi, jitter, d_rgb, d_dist, dtype):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        P = {k: v.clone().to(dtype).requires_grad_(True) for k, v in params.items()}
        po, pd, pv = (t.clone().to(dtype).requires_grad_(True) for t in (pts_o, pts_d, view))
        rgb, dist, t = trace_util.traced_render(P, po, pd, pv, z_lo.to(dtype), z_hi.to(dtype), jitter.to(dtype), dist_alpha=False, white_bg=False)
        ((rgb * d_rgb.to(dtype)).sum() + (dist * d_dist.to(dtype)).sum()).backward()
    finally:
        torch.set_default_dtype(prev)
    R, N = jitter.shape
    S = R * N
    st = {"z": t["z"].reshape(-1), "posenc": t["e"], "direnc": t["dir"]}
    for i in range(8):
        st["h%d" % (i + 1)] = t["h%d" % (i + 1)]
    st.update({"sigma_raw": t["raw"].reshape(-1), "colour hidden g": t["g"], "rgb (per sample)": t["rgb"], "alpha": t["alpha"].reshape(-1),
               "RGB (composited)": rgb, "dist (composited)": dist,
               "d rgb_pre": t["rgbpre"].grad, "d sigma_raw": t["raw"].grad.reshape(-1), "d g_pre": t["gpre"].grad})
    for i in range(7, -1, -1):
        st["d pre%d" % (i + 1)] = t["pre%d" % (i + 1)].grad
    st.update({"d point": t["pts"].grad, "d pts_o": po.grad, "d pts_d": pd.grad, "d view": pv.grad})
    for n in L.LAYER_NAMES:
        st["dW " + n], st["db " + n] = P[n + ".weight"].grad, P[n + ".bias"].grad
    return {k: v.detach().double() for k, v in st.items()}


def hip(params, pts_o, pts_d, view, z_lo, z_hi, jitter, d_rgb, d_dist, D, products):
    dev = torch.device("cuda")
    prev = L.set_fp32_products(products)
    try:
        R, N = jitter.shape
        lib = L.load()
        cfg = L.make_cfg(R, N, D, train=True)
        cu = lambda t: t.to(dev).contiguous()
        w_d = [cu(params[n + ".weight"]) for n in L.LAYER_NAMES]
        b_d = [cu(params[n + ".bias"]) for n in L.LAYER_NAMES]
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        packed = torch.empty(lib.nnr_packed_floats(C.byref(cfg)), device=dev)
        ps = L.params_struct(w_d, b_d)
        L.check(lib.nnr_pack_weights(C.byref(cfg), C.byref(ps), L.ptr(packed), st), "pack")
        ws = torch.zeros(lib.nnr_workspace_floats(C.byref(cfg)), device=dev)
        a = [cu(t) for t in (pts_o, pts_d, view, z_lo, z_hi, jitter)]
        rgb, dist = torch.empty(R, 3, device=dev), torch.empty(R, device=dev)
        alpha, zv = torch.empty(R, N, device=dev), torch.empty(R, N, device=dev)
        L.check(lib.nnr_mlp_fwd(C.byref(cfg), *[L.ptr(t) for t in a], L.ptr(packed), L.ptr(ws), st), "mlp_fwd")
        L.check(lib.nnr_composite_fwd(C.byref(cfg), L.ptr(rgb), L.ptr(dist), L.ptr(alpha), L.ptr(zv), L.ptr(ws), st), "composite_fwd")
        torch.cuda.synchronize()
        plane = lambda i: ops.workspace_plane(cfg, ws, i).clone()
        out = {"z": plane(1)[:, 0], "posenc": plane(10)[:, :63], "direnc": plane(19)[:, :27]}
        for i in range(8):
            out["h%d" % (i + 1)] = plane(11 + i)
        o4 = plane(0)
        out.update({"sigma_raw": o4[:, 3], "colour hidden g": plane(20), "rgb (per sample)": o4[:, :3], "alpha": alpha.reshape(-1).clone(),
                    "RGB (composited)": rgb.clone(), "dist (composited)": dist.clone()})
        d_rgb_d, d_dist_d = cu(d_rgb), cu(d_dist)
        L.check(lib.nnr_composite_bwd(C.byref(cfg), L.ptr(d_rgb_d), L.ptr(d_dist_d), L.ptr(ws), st), "composite_bwd")
        L.check(lib.nnr_mlp_dgrad(C.byref(cfg), L.ptr(packed), L.ptr(ws), st), "mlp_dgrad")
        gw, gb = [torch.zeros_like(w) for w in w_d], [torch.zeros_like(b) for b in b_d]
        gs = L.params_struct(gw, gb)
        host = np.zeros(lib.nnr_plan_bytes(C.byref(cfg)), dtype=np.uint8)
        L.check(lib.nnr_plan_build(C.byref(cfg), host.ctypes.data_as(C.c_void_p)), "plan")
        plan = torch.from_numpy(host).to(dev)
        L.check(lib.nnr_mlp_wgrad(C.byref(cfg), L.ptr(packed), C.byref(gs), L.ptr(plan), L.ptr(ws), st), "wgrad")
        d_o, d_d, d_v = (torch.empty(R, 3, device=dev) for _ in range(3))
        L.check(lib.nnr_ray_reduce(C.byref(cfg), L.ptr(d_o), L.ptr(d_d), L.ptr(d_v), L.ptr(ws), st), "ray_reduce")
        torch.cuda.synchronize()
        dout = plane(2)
        out.update({"d rgb_pre": dout[:, :3], "d sigma_raw": dout[:, 3], "d g_pre": plane(40)})
        for i in range(7, -1, -1):
            out["d pre%d" % (i + 1)] = plane(31 + i)
        out.update({"d point": plane(3)[:, :3], "d pts_o": d_o, "d pts_d": d_d, "d view": d_v})
        for i, n in enumerate(L.LAYER_NAMES):
            out["dW " + n], out["db " + n] = gw[i], gb[i]
        return {k: v.detach().cpu().double() for k, v in out.items()}
    finally:
        L.set_fp32_products(prev)


def main():
    D, R, N, seed = (int(x) for x in (sys.argv[1:5] + ["256", "256", "64", "333"][len(sys.argv) - 1:]))
    g = torch.Generator().manual_seed(seed)
    params = orc.init_params(D, seed + 1)
    pts_o = torch.randn(R, 3, generator=g) * 0.1
    d = torch.randn(R, 3, generator=g)
    pts_d = d / d.norm(dim=-1, keepdim=True)
    view = -pts_d
    z = torch.linspace(0, 1, N)
    z = 0.01 * (1 - z) + 10.0 * z
    mid = 0.5 * (z[1:] + z[:-1])
    z_lo, z_hi = torch.cat([z[:1], mid]), torch.cat([mid, z[-1:]])
    jitter = torch.rand(R, N, generator=g)
    d_rgb = torch.randn(R, 3, generator=g) / R
    d_dist = torch.randn(R, generator=g) / R * 0.04
    args = (params, pts_o, pts_d, view, z_lo, z_hi, jitter, d_rgb, d_dist)
    ref = trace(*args, torch.float64)
    cols = {"cpu fp32": trace(*args, torch.float32)}
    if torch.cuda.is_available():
        for kind in os.environ.get("NNR_BISECT_KINDS", "split2,split3,mfma").split(","):      # (round 6: + the two-term fp16 products)
            cols["hip " + kind] = hip(*args, D, kind)
    names = list(cols)
    print("render operator, D=%d, %d rays x %d samples, seed %d: error against the fp64 trace, stage by stage" % (D, R, N, seed))
    print("%-22s %-11s" % ("stage", "max|ref|") + "".join("| %-34s" % (n + ": rel-L2  max/|ref|max  gates") for n in names))
    for k, r in ref.items():
        line = "%-22s %-11.3e" % (k, float(r.abs().max()))
        for n in names:
            gk = cols[n][k].reshape(r.shape)
            den = float(r.norm())
            rl2 = float((gk - r).norm()) / den if den > 0 else float((gk - r).norm())
            mx = float((gk - r).abs().max()) / max(float(r.abs().max()), 1e-300)
            flips = int(((gk > 0) != (r > 0)).sum()) if (k.startswith("h") or k.startswith("colour")) else -1
            line += "| %-9.2e %-11.2e %-12s" % (rl2, mx, ("%d" % flips) if flips >= 0 else "")
        print(line)
    # the encodings in ulps of fp32 (device sin / cos against libm's, both against fp64)
    for k in ("posenc", "direnc"):
        r = ref[k]
        ulp = np.spacing(np.abs(r.numpy()).astype(np.float32)).astype(np.float64)
        for n in names:
            e = (cols[n][k].reshape(r.shape) - r).abs().numpy() / ulp
            print("%s, %s: error in fp32 ulps of the value: max %.2f, mean %.3f, share above 0.5 ulp %.4f" % (k, n, e.max(), e.mean(), (e > 0.5).mean()))


if __name__ == "__main__":
    main()
