"""GPU tests (-m gpu) of the three MLP kernels -- bf16 mode and fp32 -- against THEIR OWN operands, layer by layer.

The whole-step parity tests compare the bf16 mode with the bf16 oracle at a few 1e-3 (outputs) / 1e-2 (gradients): an activation that
rounds the other way in one layer moves everything downstream that far, so those bars cannot show a rare wrong term.  Here every
layer is checked against the operands the kernels themselves left in the workspace (tile-major bf16 planes, nnr_layout.h, located
with nnr_ws_plane and decoded on the host) -- no error propagates from layer to layer:
  * forward:         hidden_{l+1} = bf16(relu(bf16(W_l) in_l + b_l))   -- every stored value is the correctly rounded bf16 of the
                     float64 pre-activation to within the fp32 accumulation slack; the heads likewise in fp32
  * input gradient:  Dlt_l = bf16(relu'(hidden_{l+1}) .* (Dlt_{l+1} bf16(W_{l+1})))   -- the same statement for the transposed chain
  * weight gradient: dW_l = Dlt_l^T in_l, db_l = sum_s Dlt_l[s] in float64 -- bf16 x bf16 products are exact in fp32, so the only
                     freedom is the order of the additions.  This pins the two-plane units (skip layer: hidden | position encoding,
                     colour-hidden layer: hidden | direction encoding), the 4 x 5 tiling whose MFMAs are asm statements the
                     compiler's hazard recogniser does not see, the per-wave DMA bookkeeping and the slot reduction.
The workspace starts as NaN: a kernel that read a byte it or its producers did not write would return NaN."""
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


P_OUT4, P_DOUT4 = 0, 2
SHAPES = [(256, 37, 64), (256, 300, 128), (128, 50, 33), (256, 64, 192),
          (256, 4096, 128)]      # the last: BASELINE configs[2] (the full 256-workgroup plans; ~25 GB of float64 operands on the GPU)


def run_passes(D, R, N, bf16=True):
    """forward (training) -> compositing -> its backward -> input gradient -> weight gradient through the C ABI on seeded inputs;
    returns what the checks need.  bf16=False: the fp32 kernels, whose planes are row-major fp32."""
    from nnr import lib as L
    from nnr import ops
    lib = L.load()
    dev = torch.device("cuda")
    params = orc.init_params(D, 11)
    w = [params[n + ".weight"].to(dev) for n in L.LAYER_NAMES]
    g = torch.Generator().manual_seed(5)
    b = [(params[n + ".bias"] + 0.05 * torch.randn(params[n + ".bias"].shape, generator=g)).to(dev) for n in L.LAYER_NAMES]
    cfg = L.make_cfg(R, N, D, train=True, bf16=bf16)
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
    # the weight-gradient stage OVERWRITES its outputs: poisoned buffers prove that every element of all 24 tensors is written
    gw, gb = [torch.full_like(x, float("nan")) for x in w], [torch.full_like(x, float("nan")) for x in b]
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
    G, Gh = D // 16, D // 32
    pl = (lambda plane, groups: _plane(lib, cfg, ws, plane, S_pad, groups)) if bf16 else \
         (lambda plane, groups: _rows(lib, cfg, ws, plane, S_pad, 16 * groups))
    if not bf16:       # fp32 mode: the encodings are planes 10 / 19 themselves, the colour-hidden gradient has no extra group
        return dict(w=w, b=b, gw=gw, gb=gb, S=R * N, S_pad=S_pad, ws=ws, lib=lib, cfg=cfg,
                    X={l: pl(P_XH1 + l - 1, G) for l in range(1, 9)}, Dl={l: pl(P_DH1 + l, G) for l in range(8)},
                    E=pl(10, 4)[:, :63], F=pl(19, 2)[:, :27], Gc=pl(P_XG, Gh), DG=pl(P_DG, Gh))
    out = dict(w=w, b=b, gw=gw, gb=gb, S=R * N, S_pad=S_pad, ws=ws, lib=lib, cfg=cfg,
               X={l: pl(P_XH1 + l - 1, G) for l in range(1, 9)},           # hidden 1..8
               Dl={l: pl(P_DH1 + l, G) for l in range(8)},                 # d pre-activation of layer l (= of hidden l + 1)
               E=pl(P_XE16, 4)[:, :63], F=pl(P_XF16, 2)[:, :27], Gc=pl(P_XG, Gh), DG=pl(P_DG, Gh + 1))
    for t in list(out["X"].values()) + list(out["Dl"].values()) + [out["E"], out["F"], out["Gc"], out["DG"]]:
        assert torch.isfinite(t).all()
    return out


def _rows(lib, cfg, ws, plane, S, width):
    """(S, width) float64 of an fp32 plane: row-major, or tile-major (the gradient planes of the three-term mode: nnr_ws_plane_layout 2,
    1 KiB blocks [chunk][octet j] of [half h][sample c][4 floats], feature 8 j + 4 h + i -- include/nnr.h)"""
    pitch = C.c_int32(0)
    off = lib.nnr_ws_plane(C.byref(cfg), plane, C.byref(pitch))
    assert off >= 0 and pitch.value == width, (plane, off, pitch.value)
    if lib.nnr_ws_plane_layout(C.byref(cfg), plane) == 2:
        chunks = (S + 31) // 32
        t = ws[off: off + chunks * 32 * width].view(chunks, width // 8, 2, 32, 4)
        return t.permute(0, 3, 1, 2, 4).reshape(chunks * 32, width)[:S].double()
    return ws[off: off + S * width].view(S, width).double()


@pytest.mark.parametrize("D,R,N", SHAPES)
def test_bf16_weight_gradient_is_the_exact_product_of_the_stashed_operands(D, R, N):
    r = run_passes(D, R, N)
    w, b, gw, gb, X, Dl, E, F, Gc, DG = (r[k] for k in ("w", "b", "gw", "gb", "X", "Dl", "E", "F", "Gc", "DG"))
    Dh = D // 2
    dg, dout = DG[:, :Dh], DG[:, Dh:Dh + 4]                                          # d colour hidden; d rgb_pre[0..2], d sigma_raw
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


def _q(t):
    """bf16 rounding (nearest even) of an fp32 tensor, as float64: what the pack kernel stores for the MFMA operands"""
    return t.to(torch.bfloat16).double()


def _merged(Wg1, Wf):
    """W' = Wg[:, :D] Wf as the pack kernel forms it (nnr_pack.hip merge_kernel): per element one fp32 fma chain in index order.
    Emulated in float64 (a product of two fp32 values is exact there; the sum is rounded to fp32 after every step): a plain fp32
    matmul adds in another order, and an element that rounds to the other bf16 neighbour moves a whole output column by far more than
    half an ulp."""
    acc = torch.zeros(Wg1.shape[0], Wf.shape[1], dtype=torch.float64, device=Wf.device)
    a, b = Wg1.double(), Wf.double()
    for j in range(Wf.shape[0]):
        acc = (a[:, j:j + 1] * b[j:j + 1, :] + acc).float().double()
    return acc.float()


def _off_by(got, exact, scale, slack):
    mag = torch.maximum(got.abs(), exact.abs()).clamp_min(1e-30)
    ulp = torch.exp2(torch.floor(torch.log2(mag)) - 7)
    return (got - exact).abs() > 0.5 * ulp + slack * scale + 1e-30


def _assert_rounded(got, exact, scale, what, slack=4e-6, max_inexact=0.03):
    """`got` (values read from a bf16 plane) is the nearest bf16 of `exact` up to the fp32 accumulation error of the kernel's sum
    (`slack` x sum of |terms|): |got - exact| <= half a bf16 ulp + slack.  Also counts how often got differs from the bf16 rounding
    of the float64 value at all (a sum that lands on the other side of a rounding boundary): a small fraction."""
    bad = _off_by(got, exact, scale, slack)
    assert not bool(bad.any()), (what, int(bad.sum()))
    inexact = float((got != exact.float().to(torch.bfloat16).double()).double().mean())
    assert inexact <= max_inexact, (what, inexact)


@pytest.mark.parametrize("D,R,N", SHAPES)
def test_bf16_forward_layers_round_the_exact_products_of_their_stashed_inputs(D, R, N):
    """model/official_nerf.py:60-96 layer by layer, each from the input the kernel itself stashed."""
    r = run_passes(D, R, N)
    w, b, X, E, F, Gc, S = (r[k] for k in ("w", "b", "X", "E", "F", "Gc", "S"))

    def layer(inp, W, bias):
        Wq = _q(W)
        return inp @ Wq.T + bias.double(), inp.abs() @ Wq.abs().T + bias.double().abs()

    ins = {0: E, 1: X[1], 2: X[2], 3: X[3], 4: torch.cat([X[4], E], dim=1), 5: X[5], 6: X[6], 7: X[7]}
    for l in range(8):                       # hidden l + 1 = relu(layer l)
        pre, scale = layer(ins[l][:S], w[l], b[l])
        _assert_rounded(X[l + 1][:S], pre.clamp_min(0), scale, "hidden %d" % (l + 1))
    # colour-hidden layer on the merged matrix W' = Wg[:, :D] Wf (formed in fp32 by the pack kernel, then rounded), b' = Wg[:, :D] bf + bg
    Wg, Wf = w[10], w[9]
    Wm = _merged(Wg[:, :D], Wf)
    bm = Wg[:, :D].double() @ b[9].double() + b[10].double()
    Wmq, Wdq = _q(Wm), _q(Wg[:, D:].contiguous())
    pre = X[8][:S] @ Wmq.T + F[:S] @ Wdq.T + bm
    scale = X[8][:S].abs() @ Wmq.abs().T + F[:S].abs() @ Wdq.abs().T + bm.abs()
    _assert_rounded(Gc[:S], pre.clamp_min(0), scale, "colour hidden")
    # the two heads: bf16 products, fp32 results in plane 0 = (rgb after the sigmoid, raw density)
    out4 = _rows(r["lib"], r["cfg"], r["ws"], P_OUT4, S, 4)
    raw = X[8][:S] @ _q(w[8]).T + b[8].double()
    raw_scale = X[8][:S].abs() @ _q(w[8]).abs().T + b[8].double().abs()
    assert bool(((out4[:, 3:4] - raw).abs() <= 4e-6 * raw_scale).all()), float((out4[:, 3:4] - raw).abs().max())
    rgb = torch.sigmoid(Gc[:S] @ _q(w[11]).T + b[11].double())
    assert float((out4[:, :3] - rgb).abs().max()) <= 2e-6


@pytest.mark.parametrize("D,R,N", SHAPES)
def test_bf16_input_gradient_layers_round_the_exact_products_of_their_stashed_gradients(D, R, N):
    """The transposed chain, each layer from the gradient the kernel itself stashed for the layer above and the ReLU pattern of the
    stashed activations (gate = stored activation > 0, torch's relu backward -- also where an fp32 accumulator cancelled to exactly
    +0.0: 2 of 6.7e7 values at 4096 x 128; a gate taken from the accumulator's sign bit lets those pass and fails here); the two heads
    feed their input in fp32 (oracle/nerf_oracle.py _Bf16Head)."""
    r = run_passes(D, R, N)
    w, X, Dl, Gc, DG, S = (r[k] for k in ("w", "X", "Dl", "Gc", "DG", "S"))
    Dh = D // 2
    dout = _rows(r["lib"], r["cfg"], r["ws"], P_DOUT4, S, 4)            # fp32: d rgb_pre[0..2], d sigma_raw
    # the bf16 copies of the output gradients the weight-gradient kernel reads (P_DG's extra group)
    assert torch.equal(DG[:S, Dh:Dh + 4], dout.float().to(torch.bfloat16).double())
    assert float(DG[:S, Dh + 4:].abs().max()) == 0.0
    # d colour hidden = relu'(g) .* (d rgb_pre Wc), fp32 weights
    Wc = w[11].double()
    full = dout[:, :3] @ Wc
    _assert_rounded(DG[:S, :Dh], full * (Gc[:S] > 0), dout[:, :3].abs() @ Wc.abs(), "d colour hidden")
    # d pre-activation of layer 7 (hidden 8) = relu'(h8) .* (d g bf16(W') + d sigma_raw w_sigma)
    Wmq = _q(_merged(w[10][:, :D], w[9]))
    dg = DG[:S, :Dh]
    full = dg @ Wmq + dout[:, 3:4] @ w[8].double()
    _assert_rounded(Dl[7][:S], full * (X[8][:S] > 0), dg.abs() @ Wmq.abs() + dout[:, 3:4].abs() @ w[8].double().abs(), "d layer 7")
    for l in range(6, -1, -1):              # d pre-activation of layer l = relu'(hidden l + 1) .* (Dlt_{l+1} bf16(W_{l+1})[:, :D])
        Wq = _q(w[l + 1])[:, :D]
        full = Dl[l + 1][:S] @ Wq
        _assert_rounded(Dl[l][:S], full * (X[l + 1][:S] > 0), Dl[l + 1][:S].abs() @ Wq.abs(), "d layer %d" % l)


@pytest.mark.parametrize("D,R,N", [(256, 37, 64), (128, 50, 33), (256, 64, 192), (256, 1024, 192)])      # the last: the headline shape
def test_fp32_kernels_layer_by_layer_against_their_stashed_operands(D, R, N):
    """The same three statements for the fp32 kernels (row-major fp32 planes): each layer of the forward and of the input-gradient
    chain from the operands the kernel stashed, and the weight gradients as their products, all to fp32 accumulation error
    (no rounding step in between: a plain bound relative to the sum of |terms|)."""
    r = run_passes(D, R, N, bf16=False)
    w, b, gw, gb, X, Dl, E, F, Gc, DG, S = (r[k] for k in ("w", "b", "gw", "gb", "X", "Dl", "E", "F", "Gc", "DG", "S"))
    tol = 3e-6          # measured: <= 7e-7
    worst = 0.0

    def check(got, exact, scale, what, floor=0.0):
        nonlocal worst
        ratio = float((((got - exact).abs() - floor).clamp_min(0) / (scale + 1e-30)).max())
        worst = max(worst, ratio)
        assert ratio <= tol, (what, ratio)

    dbl = lambda t: t.double()
    # forward
    ins = {0: E, 1: X[1], 2: X[2], 3: X[3], 4: torch.cat([X[4], E], dim=1), 5: X[5], 6: X[6], 7: X[7]}
    for l in range(8):
        pre = ins[l][:S] @ dbl(w[l]).T + dbl(b[l])
        check(X[l + 1][:S], pre.clamp_min(0), ins[l][:S].abs() @ dbl(w[l]).abs().T + dbl(b[l]).abs(), "hidden %d" % (l + 1))
    Wm = dbl(_merged(w[10][:, :D], w[9]))
    Wd = dbl(w[10][:, D:])
    bm = dbl(w[10][:, :D]) @ dbl(b[9]) + dbl(b[10])
    pre = X[8][:S] @ Wm.T + F[:S] @ Wd.T + bm
    check(Gc[:S], pre.clamp_min(0), X[8][:S].abs() @ Wm.abs().T + F[:S].abs() @ Wd.abs().T + bm.abs(), "colour hidden")
    out4 = _rows(r["lib"], r["cfg"], r["ws"], P_OUT4, S, 4)
    check(out4[:, 3:4], X[8][:S] @ dbl(w[8]).T + dbl(b[8]), X[8][:S].abs() @ dbl(w[8]).abs().T + dbl(b[8]).abs(), "raw density")
    assert float((out4[:, :3] - torch.sigmoid(Gc[:S] @ dbl(w[11]).T + dbl(b[11]))).abs().max()) <= 2e-6
    # input gradient
    dout = _rows(r["lib"], r["cfg"], r["ws"], P_DOUT4, S, 4)
    dg = DG[:S]
    check(dg, (dout[:, :3] @ dbl(w[11])) * (Gc[:S] > 0), dout[:, :3].abs() @ dbl(w[11]).abs(), "d colour hidden")
    check(Dl[7][:S], (dg @ Wm + dout[:, 3:4] @ dbl(w[8])) * (X[8][:S] > 0), dg.abs() @ Wm.abs() + dout[:, 3:4].abs() @ dbl(w[8]).abs(), "d layer 7")
    for l in range(6, -1, -1):
        Wl = dbl(w[l + 1])[:, :D]
        check(Dl[l][:S], (Dl[l + 1][:S] @ Wl) * (X[l + 1][:S] > 0), Dl[l + 1][:S].abs() @ Wl.abs(), "d layer %d" % l)
    # weight gradient (all S_pad rows: what the kernel streams)
    Sp = r["S_pad"]
    doutp = _rows(r["lib"], r["cfg"], r["ws"], P_DOUT4, Sp, 4)
    pairs = {0: (Dl[0], E), 4: (Dl[4], torch.cat([X[4], E], dim=1)), 8: (doutp[:, 3:4], X[8]), 11: (doutp[:, :3], Gc)}
    pairs.update({l: (Dl[l], X[l]) for l in (1, 2, 3, 5, 6, 7)})
    # The fp16-term weight gradient (nnr_wgrad.hip wgrad_group_split2) scales each PLANE by one power of two (its maximum -> [2^13, 2^14))
    # and keeps h + 2^-11 m' in fp16: a value below 2^-49 of its plane's maximum is below half the last subnormal step of m' and is
    # dropped.  Relative to the sum of |terms| of an element that is unbounded (a unit that fires only on occluded samples has a whole
    # column of such values), in absolute terms it is S 2^-48 max|d| max|x| -- ~1e-11 of a typical gradient element here -- and that
    # absolute floor is subtracted before the relative bound is applied.
    for l, (dl, x) in sorted(pairs.items()):
        floor = Sp * 2.0 ** -48 * float(dl.abs().max()) * float(x.abs().max())
        check(dbl(gw[l]), dl.T @ x, dl.abs().T @ x.abs(), "dW %d" % l, floor)
        check(dbl(gb[l]), dl.sum(0), dl.abs().sum(0), "db %d" % l)
    check(dbl(gw[10][:, D:]), DG.T @ F, DG.abs().T @ F.abs(), "dW colour hidden, direction columns")
    print("worst error / sum of |terms|: %.2e" % worst)
