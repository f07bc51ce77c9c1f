"""GPU diagnostic (not a pytest): drives libnnr.so stage by stage through the C ABI on synthetic rays and compares
every workspace plane with the traced CPU oracle, printing max-abs errors per stage/layer.  Run on the GPU box:

    python tests/gpu_diag.py > gpurun_out/diag.txt

Exit code is 0 unless the library cannot run at all; FAIL markers flag planes above 1e-4 (relative to the plane's scale)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("nope-nerf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

import numpy as np
import torch

import layout_ref as lr
import nerf_oracle as orc
import trace_util
from nnr import lib as L
from nnr import ops

TOL = 1e-4


def report(name, got, ref):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    if got.shape != ref.shape:
        print(f"  {name:28s} SHAPE {tuple(got.shape)} vs {tuple(ref.shape)}  FAIL")
        return False
    err = (got - ref).abs().max().item() if got.numel() else 0.0
    scale = max(1.0, ref.abs().max().item()) if ref.numel() else 1.0
    bad = not np.isfinite(err) or err / scale > TOL
    mx = ref.abs().max().item() if ref.numel() else 0.0
    rms = ((got - ref).pow(2).mean().sqrt() / max(ref.pow(2).mean().sqrt().item(), 1e-30)).item() if ref.numel() else 0.0
    print(f"  {name:28s} max|err| {err:10.3e}  scale {scale:9.3e}  max|ref| {mx:9.3e}  rel-rms {rms:8.2e}  {'FAIL' if bad else 'ok'}")
    return not bad


def run(D, R, N, dist_alpha, white_bg, seed=0):
    print(f"== D={D} R={R} N={N} dist_alpha={dist_alpha} white_bg={white_bg}")
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(seed)
    params = orc.init_params(D, seed + 1, white_bg)
    weights = [params[n + ".weight"] for n in L.LAYER_NAMES]
    biases = [params[n + ".bias"] for n in L.LAYER_NAMES]
    pts_o = torch.randn(R, 3, generator=g) * 0.1
    d = torch.randn(R, 3, generator=g)
    pts_d = d / d.norm(dim=-1, keepdim=True)
    view = -pts_d
    near, far = (0.0, 1.0) if dist_alpha else (0.01, 10.0)
    z = torch.linspace(0, 1, N)
    z = near * (1 - z) + far * z
    mid = 0.5 * (z[1:] + z[:-1])
    z_lo, z_hi = torch.cat([z[:1], mid]), torch.cat([mid, z[-1:]])
    jitter = torch.rand(R, N, generator=g)
    d_rgb = torch.randn(R, 3, generator=g) / R
    d_dist = torch.randn(R, generator=g) / R * 0.04

    lib = L.load()
    relu_sigma = bool(int(os.environ.get('NNR_DIAG_RELU', '0')))
    cfg = L.make_cfg(R, N, D, dist_alpha=dist_alpha, white_bg=white_bg, relu_sigma=relu_sigma, train=True,
                     bf16=bool(int(os.environ.get('NNR_DIAG_BF16', '0'))))
    cu = lambda t: t.to(dev).contiguous()
    w_d, b_d = [cu(w) for w in weights], [cu(b) for b in biases]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    packed = torch.empty(lib.nnr_packed_floats(C.byref(cfg)), device=dev)
    ps = L.params_struct(w_d, b_d)
    L.check(lib.nnr_pack_weights(C.byref(cfg), C.byref(ps), L.ptr(packed), st), "pack")
    ref_pack = lr.pack_all([w.numpy() for w in weights], [b.numpy() for b in biases], D)
    ok = np.array_equal(packed.cpu().numpy(), ref_pack)
    print(f"  pack kernel bit-exact vs numpy layout: {'ok' if ok else 'FAIL'}")

    ws = torch.zeros(lib.nnr_workspace_floats(C.byref(cfg)), device=dev)
    a = [cu(t) for t in (pts_o, pts_d, view, z_lo, z_hi, jitter)]
    rgb = torch.empty(R, 3, device=dev); dist = torch.empty(R, device=dev)
    alpha = torch.empty(R, N, device=dev); zv = torch.empty(R, N, device=dev)
    L.check(lib.nnr_mlp_fwd(C.byref(cfg), *[L.ptr(t) for t in a], L.ptr(packed), L.ptr(ws), st), "mlp_fwd")
    L.check(lib.nnr_composite_fwd(C.byref(cfg), L.ptr(rgb), L.ptr(dist), L.ptr(alpha), L.ptr(zv), L.ptr(ws), st), "composite_fwd")
    torch.cuda.synchronize()

    # oracle with trace
    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    po, pd, pv = (t.clone().requires_grad_(True) for t in (pts_o, pts_d, view))
    orgb, odist, t = trace_util.traced_render(P, po, pd, pv, z_lo, z_hi, jitter, dist_alpha=dist_alpha, white_bg=white_bg,
                                              relu_sigma=bool(int(os.environ.get('NNR_DIAG_RELU', '0'))))
    plane = lambda i: ops.workspace_plane(cfg, ws, i)
    S = R * N
    report("z", plane(1)[:, 0], t["z"].reshape(-1))
    e_pad = torch.cat([t["e"], torch.zeros(S, 1)], dim=-1)
    report("posenc stash", plane(10), e_pad)
    for i in range(8):
        report(f"hidden {i + 1} activations", plane(11 + i), t[f"h{i + 1}"])
    out4 = plane(0)
    report("sigma_raw", out4[:, 3], t["raw"].reshape(-1))
    report("direnc stash", plane(19)[:, :27], t["dir"])   # the feature vector is never formed (merged into the colour layer)
    report("colour hidden", plane(20), t["g"])
    report("rgb (per sample)", out4[:, :3], t["rgb"])
    if os.environ.get("NNR_DIAG_RGB"):
        e = (out4[:, :3].detach().cpu() - t["rgb"].detach()).abs()
        idx = torch.arange(S)
        print("    per channel max", e.max(0).values.tolist())
        print("    tile 0 / tile 1 max", float(e[(idx // 32) % 2 == 0].max()), float(e[(idx // 32) % 2 == 1].max()))
        print("    per sample-in-chunk max (first 8 cols)", e.view(-1, 32, 3).amax((0, 2))[:8].tolist())
        pre_ref = torch.logit(t["rgb"].detach().double()); pre_got = torch.logit(out4[:, :3].detach().cpu().double())
        print("    pre-activation diff max", float((pre_got - pre_ref).abs().max()), "ref max", float(pre_ref.abs().max()))
    report("alpha", alpha, t["alpha"])
    report("RGB (composited)", rgb, orgb)
    report("dist (composited)", dist, odist)

    # backward
    loss = (orgb * d_rgb).sum() + (odist * d_dist).sum()
    loss.backward()
    d_rgb_d, d_dist_d = cu(d_rgb), cu(d_dist)      # keep the device copies alive until the kernels ran
    L.check(lib.nnr_composite_bwd(C.byref(cfg), L.ptr(d_rgb_d), L.ptr(d_dist_d), L.ptr(ws), st), "composite_bwd")
    L.check(lib.nnr_mlp_dgrad(C.byref(cfg), L.ptr(packed), L.ptr(ws), st), "mlp_dgrad")
    gw = [torch.zeros_like(w) for w in w_d]
    gb = [torch.zeros_like(b) for b in b_d]
    gs = L.params_struct(gw, gb)
    nbytes = lib.nnr_plan_bytes(C.byref(cfg))
    host = np.zeros(nbytes, dtype=np.uint8)
    L.check(lib.nnr_plan_build(C.byref(cfg), host.ctypes.data_as(C.c_void_p)), "plan")
    plan = torch.from_numpy(host).to(dev)
    L.check(lib.nnr_mlp_wgrad(C.byref(cfg), L.ptr(packed), C.byref(gs), L.ptr(plan), L.ptr(ws), st), "wgrad")
    d_o, d_d, d_v = (torch.empty(R, 3, device=dev) for _ in range(3))
    L.check(lib.nnr_ray_reduce(C.byref(cfg), L.ptr(d_o), L.ptr(d_d), L.ptr(d_v), L.ptr(ws), st), "ray_reduce")
    torch.cuda.synchronize()
    dout = plane(2)
    report("d rgb_pre", dout[:, :3], t["rgbpre"].grad)
    report("d sigma_raw", dout[:, 3], t["raw"].grad.reshape(-1))
    report("d colour hidden pre", plane(40), t["gpre"].grad)
    for i in range(7, -1, -1):
        report(f"d hidden {i + 1} pre", plane(31 + i), t[f"pre{i + 1}"].grad)
    report("d point", plane(3)[:, :3], t["pts"].grad)
    report("d view (per sample)", plane(4)[:, :3].view(R, N, 3).sum(1), pv.grad)
    report("d pts_o", d_o, po.grad)
    report("d pts_d", d_d, pd.grad)
    report("d view", d_v, pv.grad)
    for i, n in enumerate(L.LAYER_NAMES):
        report(f"dW {n}", gw[i], P[n + ".weight"].grad)
        report(f"db {n}", gb[i], P[n + ".bias"].grad)


if __name__ == "__main__":
    torch.manual_seed(0)
    print(torch.cuda.get_device_name(0))
    run(128, 8, 64, False, False)
    run(256, 6, 192, False, False)
    run(128, 5, 33, True, True)
    run(256, 4, 40, True, False)
