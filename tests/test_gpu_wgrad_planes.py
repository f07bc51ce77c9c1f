"""GPU test (-m gpu) of the bf16 weight-gradient kernel ALONE (nnr_wgrad_bf16.hip), against the operands it is given.

The whole-step parity tests compare gradients with the bf16 oracle at a few 1e-2 (bf16 rounding of activations that differ in the
last bit moves them that far); here the bar is the kernel's own arithmetic.  After forward + input-gradient passes through the C ABI
the workspace holds the very bf16 operands the kernel streams (tile-major planes, nnr_layout.h; nnr_ws_plane): decoded on the host,
    dW_l = Dlt_l^T X_l,  db_l = sum_s Dlt_l[s]
in float64 must agree with what nnr_mlp_wgrad returns to within fp32 accumulation error -- every bf16 x bf16 product is exact in
fp32, so the only freedom is the order of the additions.  This pins the two-plane units (skip layer: hidden | position encoding,
colour-hidden layer: hidden | direction encoding), the 4 x 5 tiling whose MFMAs are asm statements the compiler's hazard recogniser
does not see, the per-wave DMA bookkeeping and the slot reduction.  The workspace starts as NaN: a kernel that read a byte it or its
producers did not write would return NaN."""
import ctypes as C

import pytest
import torch

import nerf_oracle as orc

pytestmark = pytest.mark.gpu

P_XH1, P_XG, P_XE16, P_XF16, P_DH1, P_DG = 11, 20, 21, 22, 31, 40


def _plane(lib, cfg, ws, plane, S_pad, groups):
    """(S_pad, 16 * groups) float64 from a tile-major bf16 plane: blocks [chunk][group], block = [lane 32 h + c][8 bf16] holding,
    for sample 32 chunk + c, features 16 g + 4 h + k and 16 g + 8 + 4 h + k (k < 4)."""
    pitch = C.c_int32(0)
    off = lib.nnr_ws_plane(C.byref(cfg), plane, C.byref(pitch))
    assert off >= 0 and pitch.value == 8 * groups, (plane, off, pitch.value)
    raw = ws[off: off + S_pad * 8 * groups].view(torch.bfloat16)
    t = raw.view(S_pad // 32, groups, 2, 32, 2, 4)           # chunk, g, h, c, j, k   (feature = 16 g + 8 j + 4 h + k)
    return t.permute(0, 3, 1, 4, 2, 5).reshape(S_pad, 16 * groups).double()


@pytest.mark.parametrize("D,R,N", [(256, 37, 64), (256, 300, 128), (128, 50, 33), (256, 64, 192)])
def test_bf16_weight_gradient_is_the_exact_product_of_the_stashed_operands(D, R, N):
    from nnr import lib as L
    from nnr import ops
    lib = L.load()
    dev = torch.device("cuda")
    params = orc.init_params(D, 11)
    w = [params[n + ".weight"].to(dev) for n in L.LAYER_NAMES]
    b = [(params[n + ".bias"] + 0.05 * torch.randn_like(params[n + ".bias"])).to(dev) for n in L.LAYER_NAMES]
    cfg = L.make_cfg(R, N, D, train=True, bf16=True)
    g = torch.Generator().manual_seed(5)
    d = torch.randn(R, 3, generator=g)
    d = (d / d.norm(dim=-1, keepdim=True)).to(dev)
    o = (0.3 * torch.randn(R, 3, generator=g)).to(dev)
    z = torch.linspace(0, 1, N)
    z = 0.01 * (1 - z) + 4 * z
    mid = 0.5 * (z[1:] + z[:-1])
    lo, hi = torch.cat([z[:1], mid]).to(dev), torch.cat([mid, z[-1:]]).to(dev)
    jit = torch.rand(R, N, generator=g).to(dev)
    view = (-d).contiguous()
    packed = ops._packed_for(cfg, w, b)
    ws = torch.full((lib.nnr_workspace_floats(C.byref(cfg)),), float("nan"), device=dev)
    rgb, dst = torch.empty(R, 3, device=dev), torch.empty(R, device=dev)
    d_rgb, d_dst = torch.randn(R, 3, generator=g).to(dev), torch.randn(R, generator=g).to(dev)
    gw, gb = [torch.zeros_like(x) for x in w], [torch.zeros_like(x) for x in b]
    gs = L.params_struct(gw, gb)
    plan = ops._plan_for(cfg, dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.nnr_mlp_fwd(C.byref(cfg), L.ptr(o), L.ptr(d), L.ptr(view), L.ptr(lo), L.ptr(hi), L.ptr(jit), L.ptr(packed), L.ptr(ws), st), "fwd")
    L.check(lib.nnr_composite_fwd(C.byref(cfg), L.ptr(rgb), L.ptr(dst), None, None, L.ptr(ws), st), "composite")
    L.check(lib.nnr_composite_bwd(C.byref(cfg), L.ptr(d_rgb), L.ptr(d_dst), L.ptr(ws), st), "composite_bwd")
    L.check(lib.nnr_mlp_dgrad(C.byref(cfg), L.ptr(packed), L.ptr(ws), st), "dgrad")
    L.check(lib.nnr_mlp_wgrad(C.byref(cfg), L.ptr(packed), C.byref(gs), L.ptr(plan), L.ptr(ws), st), "wgrad")
    torch.cuda.synchronize()

    S_pad = (R * N + 127) // 128 * 128
    G, Gh, Dh = D // 16, D // 32, D // 2
    X = {l: _plane(lib, cfg, ws, P_XH1 + l - 1, S_pad, G) for l in range(1, 9)}      # hidden 1..8
    Dl = {l: _plane(lib, cfg, ws, P_DH1 + l, S_pad, G) for l in range(8)}            # d pre-activation of hidden l + 1 = of layer l
    E = _plane(lib, cfg, ws, P_XE16, S_pad, 4)[:, :63]
    F = _plane(lib, cfg, ws, P_XF16, S_pad, 2)[:, :27]
    Gc = _plane(lib, cfg, ws, P_XG, S_pad, Gh)
    DG = _plane(lib, cfg, ws, P_DG, S_pad, Gh + 1)
    dg, dout = DG[:, :Dh], DG[:, Dh:Dh + 4]                                          # d colour hidden; d rgb_pre[0..2], d sigma_raw
    for t in list(X.values()) + list(Dl.values()) + [E, F, Gc, DG]:
        assert torch.isfinite(t).all()
    assert float(dg.abs().max()) > 0 and float(Dl[0].abs().max()) > 0               # the backward did reach the first layer

    def prod(dlt, x):          # expected product and the sum of |terms| (the scale of the fp32 accumulation error)
        return dlt.T @ x, dlt.abs().T @ x.abs()

    exp_w, scale_w, exp_b, scale_b = {}, {}, {}, {}
    exp_w[0], scale_w[0] = prod(Dl[0], E)
    for l in (1, 2, 3, 5, 6, 7):
        exp_w[l], scale_w[l] = prod(Dl[l], X[l])
    exp_w[4], scale_w[4] = prod(Dl[4], torch.cat([X[4], E], dim=1))
    for l in range(8):
        exp_b[l], scale_b[l] = Dl[l].sum(0), Dl[l].abs().sum(0)
    exp_w[8], scale_w[8] = prod(dout[:, 3:4], X[8])
    exp_b[8], scale_b[8] = dout[:, 3:4].sum(0), dout[:, 3:4].abs().sum(0)
    exp_w[11], scale_w[11] = prod(dout[:, :3], Gc)
    exp_b[11], scale_b[11] = dout[:, :3].sum(0), dout[:, :3].abs().sum(0)
    dWm, _ = prod(dg, X[8])                                                          # the merged matrix W' = Wg[:, :D] Wf
    dbm = dg.sum(0)
    dir_w, dir_scale = prod(dg, F)

    tol = 5e-6      # x sum of |terms| (measured: <= 5e-7): fp32 roundings of the chained sums stay inside, a wrong or missing term does not
    for l in sorted(exp_w):
        got = gw[l].double()
        err = (got - exp_w[l]).abs()
        assert bool((err <= tol * scale_w[l] + 1e-30).all()), (l, float(err.max()), float(exp_w[l].abs().max()))
        assert float(exp_w[l].abs().max()) > 0
        errb = (gb[l].double() - exp_b[l]).abs()
        assert bool((errb <= tol * scale_b[l] + 1e-30).all()), (l, float(errb.max()), float(exp_b[l].abs().max()))
    err = (gw[10][:, D:].double() - dir_w).abs()
    assert bool((err <= tol * dir_scale + 1e-30).all()), float(err.max())
    # the un-merge step (wgrad_unmerge_kernel: fp32 fma chains over the fp32 weights):
    #   dWf = Wg1^T dW'   dWg[:, :D] = dW' Wf^T + db' bf^T   dbf = Wg1^T db'   dbg = db'
    Wg1, Wf, bf = w[10][:, :D].double(), w[9].double(), b[9].double()
    for got, exp in ((gw[9], Wg1.T @ dWm), (gw[10][:, :D], dWm @ Wf.T + torch.outer(dbm, bf)), (gb[9], Wg1.T @ dbm), (gb[10], dbm)):
        assert float((got.double() - exp).abs().max()) <= 1e-4 * float(exp.abs().max()), (float((got.double() - exp).abs().max()), float(exp.abs().max()))
