"""GPU parity tests (-m gpu): the HIP path, called through the C ABI via the `model` package exactly as train.py would,
against (a) the golden vectors minted from the real reference and (b) the CPU oracle on the same seeded inputs.
Tolerance: 1e-4 max-abs for rgb / depth / gradients (gradients normalised by the golden tensor's max-abs when that
exceeds 1) -- the fp32 bar of BASELINE.json; indices, masks and z-samples must agree to 1e-6."""
import ctypes as C

import numpy as np
import pytest
import torch

import golden_util as gu
import layout_ref as lr
import nerf_oracle as orc
import trace_util

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _loaded_native():
    maps = open("/proc/self/maps").read()
    return "libnnr.so" in maps


def build(case, device):
    import model as mdl
    from test_host_logic import make_cfg
    rc = gu.render_cfg(case)
    cfg = make_cfg(int(case["cfg.hidden"]), **{k: rc[k] for k in ('num_points', 'dist_alpha', 'sample_option', 'depth_range',
                                                                  'normalise_ray', 'white_background', 'use_ray_dir')})
    cfg['model']['occ_activation'] = rc['occ_activation']
    net = mdl.OfficialStaticNerf(cfg)
    net.load_state_dict(case["weights"])
    model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device=device), cfg, device=device)
    pose = mdl.LearnPose(gu.N_CAMS, True, True, cfg).to(device)
    dist = mdl.Learn_Distortion(gu.N_CAMS, True, True, cfg).to(device)
    t = gu.tensors(case)
    with torch.no_grad():
        pose.r.copy_(t["pose_r"]); pose.t.copy_(t["pose_t"])
        dist.global_scales.copy_(t["scales"]); dist.global_shifts.copy_(t["shifts"])
    return net, model, pose, dist, t


def run_hip(case, eval_=False, mfma_dtype=None):
    dev = torch.device("cuda")
    net, model, pose, dist, t = build(case, dev)
    if mfma_dtype is not None:
        model.renderer.cfg['mfma_dtype'] = mfma_dtype
    from model.common import arange_pixels
    from model.losses import Loss
    h, w, cam = int(case["cfg.h"]), int(case["cfg.w"]), int(case["cfg.cam"])
    ray_idx = t["ray_idx"].to(dev)
    world_mat = torch.inverse(pose(cam)).unsqueeze(0)
    sc, sh = dist(cam)
    depth_in = t["depth_img"].to(dev) * sc + sh
    p = arange_pixels((h, w), device=dev)[1][:, ray_idx]
    renderer = model.renderer
    jit = t["jitter"]
    if jit is not None:
        # feed the fixture's jitter instead of a fresh torch.rand draw (GPU and CPU generators differ)
        orig = torch.rand
        torch.rand = lambda *a, **k: jit.to(dev)
    try:
        ctx = torch.no_grad() if eval_ else torch.enable_grad()
        with ctx:
            out = model(p, ray_idx, t["K"].to(dev), world_mat, torch.eye(4, device=dev)[None], 'nope_nerf', it=0,
                        eval_mode=eval_, depth_img=depth_in, add_noise=jit is not None, img_size=(h, w))
    finally:
        if jit is not None:
            torch.rand = orig
    grads = {}
    if not eval_:
        rgb_gt = t["img"].to(dev).view(1, 3, h * w).permute(0, 2, 1)[:, ray_idx]
        crit = Loss({'depth_loss_type': 'l1'})
        loss = crit.get_rgb_full_loss(out['rgb'], rgb_gt, 'l1') + 0.04 * crit.get_depth_loss(out['depth_pred'], out['depth_gt'])
        loss.backward()
        out["loss"] = loss.detach()
        grads = {"w." + k: v.grad for k, v in net.named_parameters()}
        grads.update(pose_r=pose.r.grad, pose_t=pose.t.grad, scales=dist.global_scales.grad, shifts=dist.global_shifts.grad)
    return out, grads


@pytest.mark.parametrize("mode", [0, 2, 3])
@pytest.mark.parametrize("D", [128, 256])
def test_pack_kernel_is_bit_exact(D, mode):
    """mode 0: fp32 fragments; mode 2: three bf16 terms per weight (NNR_F_SPLIT3); mode 3: two fp16 terms of the scaled weight in three
    fragment classes, the scale table, the scaled biases (NNR_F_SPLIT2) -- against tests/layout_ref.py, bit for bit."""
    from nnr import lib as L
    dev = torch.device("cuda")
    params = orc.init_params(D, 3)
    w = [params[n + ".weight"] for n in L.LAYER_NAMES]
    b = [params[n + ".bias"] for n in L.LAYER_NAMES]
    cfg = L.make_cfg(1, 1, D)
    cfg = L.Cfg(1, 1, D, (cfg.flags & ~(L.NNR_F_SPLIT3 | L.NNR_F_SPLIT2)) | (L.NNR_F_SPLIT3 if mode >= 2 else 0) | (L.NNR_F_SPLIT2 if mode == 3 else 0))
    lib = L.load()
    packed = torch.empty(lib.nnr_packed_floats(C.byref(cfg)), device=dev)
    wd, bd = [x.to(dev) for x in w], [x.to(dev) for x in b]
    ps = L.params_struct(wd, bd)
    L.check(lib.nnr_pack_weights(C.byref(cfg), C.byref(ps), L.ptr(packed),
                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), "pack")
    ref, exact = lr.pack_all([x.numpy() for x in w], [x.numpy() for x in b], D, with_exact_mask=True, mode=mode)
    got = packed.cpu().numpy()
    assert got.size == ref.size
    assert np.array_equal(got[exact].view(np.uint32), ref[exact].view(np.uint32))   # pure re-ordering (and exact term splits): bit-exact
    if mode == 0:
        np.testing.assert_allclose(got[~exact], ref[~exact], rtol=0, atol=2e-6)   # merged feature/colour matrix: an fp32 fma chain
    else:
        # the merged matrix differs from the host's by an fp32 rounding, so its terms may differ in the m / l fragments: compare the
        # values the fragments stand for (sum of the three terms) in the fp32 tail, and the fragments of the merged parts loosely
        tail = ref.size - (D // 2) * D - D // 2 - D * D - (D // 2) * D - D
        np.testing.assert_allclose(got[tail:][~exact[tail:]], ref[tail:][~exact[tail:]], rtol=0, atol=2e-6)
    assert _loaded_native()


@pytest.mark.parametrize("name", gu.GRAD_CASES)
def test_training_step_matches_reference_golden(name):
    case = gu.load_case(name)
    out, grads = run_hip(case)
    assert _loaded_native()
    np.testing.assert_allclose(out["z_vals"].cpu().numpy(), case["out.z_vals"], rtol=0, atol=1e-6)
    for k in ("rgb", "depth_pred", "depth_gt", "alpha"):
        got = out[k].detach().cpu().numpy()
        assert got.shape == case["out." + k].shape, k         # same mask -> same number of valid depths
        np.testing.assert_allclose(got, case["out." + k], rtol=0, atol=TOL, err_msg=k)
    assert abs(float(out["loss"]) - float(case["out.loss"])) <= TOL
    for k, (kind, ref, norm) in gu.golden_grads(case).items():
        gu.compare_grad(k, grads[k], kind, ref, norm, TOL)


@pytest.mark.parametrize("name", gu.EVAL_CASES)
def test_eval_render_matches_reference_golden(name):
    case = gu.load_case(name)
    out, _ = run_hip(case, eval_=True)
    for k in ("rgb", "depth_pred", "depth_gt", "alpha", "z_vals"):
        got = out[k].detach().cpu().numpy()
        assert got.shape == case["out." + k].shape, k
        np.testing.assert_allclose(got, case["out." + k], rtol=0, atol=TOL, err_msg=k)


def test_against_live_oracle_full_gradients():
    """Same check against the oracle run live on this host (full tensors for the D=256 case, not the subsample)."""
    case = gu.load_case("tanks_d256_n192")
    out, grads = run_hip(case)
    oout, ograds = gu.run_oracle(case)
    np.testing.assert_allclose(out["rgb"].detach().cpu().numpy(), oout["rgb"].detach().numpy(), rtol=0, atol=TOL)
    for k, g in ograds.items():
        ref = g.detach().double().numpy()
        got = grads[k].detach().cpu().double().numpy()
        scale = max(1.0, np.abs(ref).max())
        assert np.abs(got - ref).max() / scale <= TOL, k
        rl2 = gu.rel_l2(got, ref)        # the max-abs bar is absolute for tensors below 1 (all of them): this one is relative
        gu.parity_log("live oracle tanks_d256_n192 %s rel-L2 %.3e ref-max %.3e" % (k, rl2, float(np.abs(ref).max())))
        assert rl2 <= gu.REL_L2_TOL_SMALL, (k, rl2)      # 64 rays x 192 samples: the small-case bar (golden_util)


def _synthetic(D, R, N, seed=0, dist_alpha=False):
    g = torch.Generator().manual_seed(seed)
    params = orc.init_params(D, seed + 1)
    d = torch.randn(R, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    o = (torch.randn(3, generator=g) * 0.1).expand(R, 3).contiguous()
    z = torch.linspace(0, 1, N)
    z = 0.01 * (1 - z) + 10 * z
    mid = 0.5 * (z[1:] + z[:-1])
    return params, o, d, torch.cat([z[:1], mid]), torch.cat([mid, z[-1:]]), torch.rand(R, N, generator=g)


def _hip_render(params, o, d, lo, hi, jit, D, d_rgb=None, d_dist=None, dist_alpha=False, bf16=False):
    import nnr
    dev = torch.device("cuda")
    from nnr import lib as L
    w = [params[n + ".weight"].to(dev).requires_grad_(True) for n in L.LAYER_NAMES]
    b = [params[n + ".bias"].to(dev).requires_grad_(True) for n in L.LAYER_NAMES]
    oo, dd = o.to(dev).requires_grad_(True), d.to(dev).requires_grad_(True)
    vv = (-d).to(dev).requires_grad_(True)
    rgb, dist, alpha, z = nnr.render_rays(oo, dd, vv, lo.to(dev), hi.to(dev), jit.to(dev) if jit is not None else None, w, b,
                                          hidden=D, dist_alpha=dist_alpha, white_bg=False, relu_sigma=False, bf16=bf16)
    grads = None
    if d_rgb is not None:
        (rgb * d_rgb.to(dev)).sum().add((dist * d_dist.to(dev)).sum()).backward()
        grads = [oo.grad, dd.grad, vv.grad] + [x.grad for x in w] + [x.grad for x in b]
    return rgb.detach(), dist.detach(), alpha, grads


@pytest.mark.parametrize("R,N,bf16", [(1024, 192, False), (4096, 128, True)])
def test_full_size_properties(R, N, bf16):
    """BASELINE.json config 2 (1024 rays x 192 samples, D=256, fp32) and config 3 (4096 x 128, bf16 products): size-independent
    properties of the path -- determinism, ray-permutation equivariance, alpha/weight bounds, linearity of the backward in the
    upstream gradient (exact in bf16 too: a factor 2 commutes with the rounding), and shard additivity (sum of half-batch
    gradients == full-batch gradient: the data-parallel identity)."""
    D = 256
    import functools
    global _hip_render
    plain, _hip_render = _hip_render, functools.partial(_hip_render, bf16=bf16)
    try:
        _full_size_properties(D, R, N)
    finally:
        _hip_render = plain


def _full_size_properties(D, R, N):
    params, o, d, lo, hi, jit = _synthetic(D, R, N)
    g = torch.Generator().manual_seed(5)
    d_rgb, d_dist = torch.randn(R, 3, generator=g) / R, torch.randn(R, generator=g) / R
    rgb, dist, alpha, grads = _hip_render(params, o, d, lo, hi, jit, D, d_rgb, d_dist)
    rgb2, dist2, _, _ = _hip_render(params, o, d, lo, hi, jit, D)
    assert torch.equal(rgb, rgb2) and torch.equal(dist, dist2)                      # forward is deterministic
    assert torch.all((alpha >= 0) & (alpha <= 1)) and torch.all((rgb > 0) & (rgb < 1.0 + 1e-5))
    assert torch.all(dist >= 0) and torch.all(dist <= 10.0 * (1 + 1e-4))
    perm = torch.randperm(R, generator=g)
    rgb_p, dist_p, _, _ = _hip_render(params, o[perm], d[perm], lo, hi, jit[perm], D)
    assert torch.equal(rgb_p, rgb[perm.cuda()]) and torch.equal(dist_p, dist[perm.cuda()])   # rays are independent
    # linearity: grads(2 g) == 2 grads(g)
    _, _, _, grads2 = _hip_render(params, o, d, lo, hi, jit, D, 2 * d_rgb, 2 * d_dist)
    for a, b2 in zip(grads, grads2):
        scale = max(1.0, float(a.abs().max()))
        assert float((2 * a - b2).abs().max()) / scale <= 2e-5
    # shard additivity
    half = R // 2
    _, _, _, ga = _hip_render(params, o[:half], d[:half], lo, hi, jit[:half], D, d_rgb[:half], d_dist[:half])
    _, _, _, gb = _hip_render(params, o[half:], d[half:], lo, hi, jit[half:], D, d_rgb[half:], d_dist[half:])
    for full, x, y in zip(grads[3:], ga[3:], gb[3:]):
        scale = max(1.0, float(full.abs().max()))
        assert float((full - (x + y)).abs().max()) / scale <= 2e-5
    assert torch.allclose(torch.cat([ga[0], gb[0]]), grads[0], atol=1e-6) and torch.allclose(torch.cat([ga[1], gb[1]]), grads[1], atol=1e-5)


def test_full_size_against_oracle_sample():
    """Config-2 shape on the HIP path vs the oracle on a 16-ray subset (rays are independent, so the subset's forward
    must agree exactly with the same rays inside the big batch)."""
    D, R, N = 256, 1024, 192
    params, o, d, lo, hi, jit = _synthetic(D, R, N, seed=3)
    rgb, dist, _, _ = _hip_render(params, o, d, lo, hi, jit, D)
    sel = torch.arange(0, R, 64)
    P = {k: v for k, v in params.items()}
    with torch.no_grad():
        orgb, odist, _ = trace_util.traced_render(P, o[sel], d[sel], -d[sel], lo, hi, jit[sel], dist_alpha=False, white_bg=False)
    assert float((rgb[sel.cuda()].cpu() - orgb).abs().max()) <= TOL
    assert float((dist[sel.cuda()].cpu() - odist).abs().max()) <= TOL * 10      # distances reach 10


@pytest.mark.parametrize("D,R,N,dist_alpha,white_bg,relu_sigma,jittered", [
    (128, 1, 2, False, False, False, True),      # the smallest render: one ray, two samples (126 of the 128 padded samples are filler)
    (128, 7, 17, True, False, False, True),      # ragged sizes, dist_alpha
    (256, 33, 65, False, True, True, True),      # white background, ReLU density
    (256, 5, 130, True, True, False, False),     # no jitter (eval sampling), long rays
    (128, 129, 3, False, False, True, True),     # more rays than a wave, samples not a multiple of anything
])
def test_ragged_shapes_and_every_flag_against_live_oracle(D, R, N, dist_alpha, white_bg, relu_sigma, jittered):
    """Forward and full backward on sizes that are not multiples of the 32-sample wave / 128-sample workgroup / 16-sample
    weight-gradient granule, with every rendering switch, against the oracle evaluated on this host."""
    import nnr
    from nnr import lib as L
    dev = torch.device("cuda")
    params, o, d, lo, hi, jit = _synthetic(D, R, N, seed=11 + R, dist_alpha=dist_alpha)
    if not jittered:
        jit = None
    g = torch.Generator().manual_seed(5)
    d_rgb, d_dist = torch.randn(R, 3, generator=g) / R, torch.randn(R, generator=g) / R
    w = [params[n + ".weight"].to(dev).requires_grad_(True) for n in L.LAYER_NAMES]
    b = [params[n + ".bias"].to(dev).requires_grad_(True) for n in L.LAYER_NAMES]
    oo, dd, vv = o.to(dev).requires_grad_(True), d.to(dev).requires_grad_(True), (-d).to(dev).requires_grad_(True)
    rgb, dist, alpha, z = nnr.render_rays(oo, dd, vv, lo.to(dev), hi.to(dev), jit.to(dev) if jit is not None else None, w, b,
                                          hidden=D, dist_alpha=dist_alpha, white_bg=white_bg, relu_sigma=relu_sigma)
    (rgb * d_rgb.to(dev)).sum().add((dist * d_dist.to(dev)).sum()).backward()
    # The oracle in fp64 is the reference; the oracle in fp32 tells how well-conditioned each quantity is in fp32 at all
    # (with a ReLU density and 2^9-fold position frequencies d(point) of some rays is only good to 1e-3 in ANY fp32 evaluation).
    def oracle(dt):
        P = {k: v.clone().to(dt).requires_grad_(True) for k, v in params.items()}
        po, pd, pv = (t.clone().to(dt).requires_grad_(True) for t in (o, d, -d))
        orgb, odist, _ = trace_util.traced_render(P, po, pd, pv, lo.to(dt), hi.to(dt), None if jit is None else jit.to(dt),
                                                  dist_alpha=dist_alpha, white_bg=white_bg, relu_sigma=relu_sigma)
        ((orgb * d_rgb.to(dt)).sum() + (odist * d_dist.to(dt)).sum()).backward()
        out = {"rgb": orgb, "dist": odist, "d pts_o": po.grad, "d pts_d": pd.grad, "d view": pv.grad}
        for n in L.LAYER_NAMES:
            out["dW " + n], out["db " + n] = P[n + ".weight"].grad, P[n + ".bias"].grad
        return {k: v.detach().double() for k, v in out.items()}
    ref, ref32 = oracle(torch.float64), oracle(torch.float32)
    got = {"rgb": rgb, "dist": dist, "d pts_o": oo.grad, "d pts_d": dd.grad, "d view": vv.grad}
    for i, n in enumerate(L.LAYER_NAMES):
        got["dW " + n], got["db " + n] = w[i].grad, b[i].grad
    for k, r in ref.items():
        scale = max(1.0, float(r.abs().max()))
        tol = max(TOL * scale, 4.0 * float((ref32[k] - r).abs().max()))
        err = float((got[k].detach().cpu().double() - r).abs().max())
        assert err <= tol, (k, D, R, N, err, tol)


def test_full_image_inference_matches_oracle(tmp_path):
    """The evaluation / visualisation drivers (Extract_Images -> forward-only kernel in ray chunks) against the oracle."""
    import model as mdl
    from model.extracting_images import Extract_Images
    from test_host_logic import make_cfg
    case = gu.load_case("tanks_eval_d128")
    t = gu.tensors(case)
    dev = torch.device("cuda")
    cfg = make_cfg(128, num_points=48)
    cfg['extract_images'] = {'resolution': [21, 34]}
    net = mdl.OfficialStaticNerf(cfg)
    net.load_state_dict(case["weights"])
    model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device=dev), cfg, device=dev)
    c2w = orc.pose_c2w(t["pose_r"][1], t["pose_t"][1])
    ex = Extract_Images(model.renderer, cfg, use_learnt_poses=True, use_learnt_focal=False, device=dev, render_type='nope_nerf')
    ex.points_batch_size = 300          # several ragged chunks
    data = {'img.camera_mat': t["K"], 'img.scale_mat': torch.eye(4)[None], 'img.idx': 0}
    out = ex.generate_images(data, str(tmp_path), [c2w.to(dev)], None, 0, output_geo=False)
    rc = dict(gu.render_cfg(case), num_points=48)
    with torch.no_grad():
        ref = orc.render(case["weights"], orc.pixel_grid(21, 34), torch.ones(1, 21 * 34, 1), t["K"],
                         torch.inverse(c2w)[None], torch.eye(4)[None], rc, jitter=None, eval_=True)
    img_ref = (ref["rgb"].view(21, 34, 3).numpy() * 255)
    assert np.abs(out['img'].astype(np.float64) - img_ref).max() <= 1.0          # uint8 quantisation of a 1e-4 match
    depth = np.load(str(tmp_path / "depth_out" / "0.npy"))
    np.testing.assert_allclose(depth, ref["depth_pred"].view(21, 34).numpy(), rtol=0, atol=2e-4)


@pytest.mark.parametrize("name", ["tanks_d128", "tanks_d256_n192", "llff_ndc_d128"])
def test_bf16_mfma_mode_against_fp32_golden(name, capsys):
    """BASELINE configs[2] arithmetic: bf16 MFMA products with fp32 accumulation in the MLP forward and input-gradient
    kernels (rendering.mfma_dtype: bf16), everything else fp32.  Compared with the fp32 reference golden; the measured errors are
    printed.  Outputs: 2e-3 absolute (measured ~1e-4).  Gradients: relative L2 error of every tensor <= 0.2 (measured 0.04-0.1):
    activations agree to 0.3 %, but a ReLU whose pre-activation is below the bf16 noise flips, and each flipped unit
    contributes its whole gradient to an element-wise difference -- ~1 % of the units, i.e. ~10 % in L2, while the element-wise
    maximum can reach 25 % of the tensor's largest entry.  That is the nature of reduced-precision training, not a defect of
    the kernels (tests/gpu_diag.py with NNR_DIAG_BF16=1 shows the per-layer numbers)."""
    case = gu.load_case(name)
    out, grads = run_hip(case, mfma_dtype='bf16')
    errs = {}
    for k in ("rgb", "depth_pred"):
        got = out[k].detach().cpu().numpy()
        errs[k] = float(np.abs(got - case["out." + k]).max())
        assert errs[k] <= 2e-3, (k, errs[k])
    np.testing.assert_allclose(out["z_vals"].cpu().numpy(), case["out.z_vals"], rtol=0, atol=1e-6)   # sampling is not affected
    worst_l2, worst_max = ("", 0.0), ("", 0.0)
    for k, (kind, ref, norm) in gu.golden_grads(case).items():
        g = grads[k].detach().cpu().numpy().reshape(-1).astype(np.float64)
        if kind != "full":
            g = g[::g.size // gu.SUBSAMPLE]
        r = ref.reshape(-1).astype(np.float64)
        if np.abs(r).max() == 0:
            continue
        l2 = float(np.linalg.norm(g - r) / np.linalg.norm(r))
        mx = float(np.abs(g - r).max() / np.abs(r).max())
        worst_l2 = max(worst_l2, (k, l2), key=lambda t: t[1])
        worst_max = max(worst_max, (k, mx), key=lambda t: t[1])
        assert l2 <= 0.2, (k, l2)
    with capsys.disabled():
        print("\nbf16 mode vs fp32 golden [%s]: rgb %.2e depth %.2e; gradients: worst relative L2 %.3f (%s), worst element / max %.3f (%s)"
              % (name, errs["rgb"], errs["depth_pred"], worst_l2[1], worst_l2[0], worst_max[1], worst_max[0]))


def _oracle_bf16(case):
    """The oracle with every MFMA-shaped product in bf16 x bf16 -> fp32 (nerf_oracle.mlp_bf16: the arithmetic BASELINE configs[2]
    names, restated op by op as the bf16 kernels carry it out), on a golden case's inputs."""
    t = gu.tensors(case)
    cfg = gu.render_cfg(case)
    cfg["mfma_dtype"] = "bf16"
    h, w, cam = int(case["cfg.h"]), int(case["cfg.w"]), int(case["cfg.cam"])
    params = {k: v.clone().requires_grad_(True) for k, v in case["weights"].items()}
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
    loss, out = orc.train_step_scope(params, leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], cam, t["K"],
                                     t["depth_img"], t["img"], (h, w), t["ray_idx"], t["jitter"], cfg)
    loss.backward()
    grads = {"w." + k: v.grad for k, v in params.items()}
    grads.update({k: v.grad for k, v in leaves.items()})
    return out, grads


@pytest.mark.parametrize("name", ["tanks_d128", "tanks_d256_n192", "llff_ndc_d128", "uniform_distalpha_masked_d128"])
def test_bf16_kernels_match_the_bf16_arithmetic_oracle(name, capsys):
    """The TIGHT parity statement of the bf16 mode.  Against the fp32 reference the bf16 gradients are 5-15 % off in relative L2
    (test above) -- that is what bf16 products do to a ReLU network, not a property of these kernels: the CPU oracle run with the
    same arithmetic (operands of every hidden-layer product rounded to bf16, fp32 accumulation, fp32 everything else) lands on
    the same numbers.  HIP kernels vs that oracle: outputs 1e-4, every gradient tensor within 2.5e-2 in relative L2 (measured
    1e-3 .. 1e-2, against 5e-2 .. 1.5e-1 for bf16 vs fp32).  What is left: the device's sincosf and the host's sin / cos differ in
    the last bit for some encodings, and the order of the fp32 additions differs; a last-bit difference that straddles a bf16
    rounding boundary becomes a 0.4 % difference of that operand (2^16 ulps) and moves ReLU decisions downstream of it."""
    case = gu.load_case(name)
    out, grads = run_hip(case, mfma_dtype='bf16')
    ref, rgrads = _oracle_bf16(case)
    for k in ("rgb", "depth_pred"):
        err = float((out[k].detach().cpu() - ref[k].detach()).abs().max())
        assert err <= 1e-4, (k, err)
    worst = ("", 0.0)
    for k, r in rgrads.items():
        g = grads[k].detach().cpu().double()
        r = (r if r is not None else torch.zeros_like(g)).double()
        if float(r.abs().max()) == 0:
            assert float(g.abs().max()) == 0, k
            continue
        l2 = float((g - r).norm() / r.norm())
        worst = max(worst, (k, l2), key=lambda t: t[1])
        assert l2 <= (5e-2 if k.startswith("pose") else 2.5e-2), (k, l2)      # 6 pose numbers: a handful of rays carry their gradient
    with capsys.disabled():
        print("\nbf16 kernels vs bf16-arithmetic oracle [%s]: worst gradient relative L2 %.2e (%s)" % (name, worst[1], worst[0]))


@pytest.mark.parametrize("bf16", [False, True])
@pytest.mark.parametrize("D,R,N,dist_alpha,white_bg,relu_sigma,jittered", [
    (256, 8, 192, False, False, False, False),    # the evaluation shape class: whole chunks per ray, no jitter
    (128, 12, 64, True, True, False, True),       # dist_alpha (needs the next sample's z across the chunk boundary), white background
    (256, 4, 128, True, False, True, False),      # ReLU density
])
def test_inference_composites_in_the_kernel_epilogue(D, R, N, dist_alpha, white_bg, relu_sigma, jittered, bf16):
    """Forward-only renders that do not ask for per-sample outputs (samples=False) take the fused path of nnr_render_fwd: the MLP kernel
    composites its ray in the epilogue and writes 16 bytes per ray.  Same numbers as MLP kernel + composite kernel (the product scan
    associates differently: 32 lanes + carry instead of 64), and the per-sample planes of the workspace stay untouched."""
    import ctypes as C
    import nnr
    from nnr import lib as L
    from nnr import ops
    dev = torch.device("cuda")
    params, o, d, lo, hi, jit = _synthetic(D, R, N, seed=3 + R, dist_alpha=dist_alpha)
    w = [params[n + ".weight"].to(dev) for n in L.LAYER_NAMES]
    b = [params[n + ".bias"].to(dev) for n in L.LAYER_NAMES]
    args = (o.to(dev), d.to(dev), (-d).to(dev), lo.to(dev), hi.to(dev), jit.to(dev) if jittered else None, w, b)
    kw = dict(hidden=D, dist_alpha=dist_alpha, white_bg=white_bg, relu_sigma=relu_sigma, bf16=bf16)
    with torch.no_grad():
        rgb0, dist0, alpha0, z0 = nnr.render_rays(*args, **kw)                       # unfused: per-sample outputs requested
        cfg = L.make_cfg(R, N, D, dist_alpha=dist_alpha, white_bg=white_bg, relu_sigma=relu_sigma, bf16=bf16)
        ws = ops._take_workspace(cfg, dev)
        ws.fill_(-7.0)
        ops._give_workspace(cfg, dev, ws)                                          # the next forward-only render takes this one
        rgb1, dist1, alpha1, z1 = nnr.render_rays(*args, samples=False, **kw)
        torch.cuda.synchronize()
    assert alpha1 is None and z1 is None and alpha0.shape == (R, N)
    assert float((rgb1 - rgb0).abs().max()) <= 2e-6 and float((dist1 - dist0).abs().max()) <= 2e-5 * max(1.0, float(dist0.abs().max()))
    assert bool((ws == -7.0).all())                                                # nothing per-sample went to HBM


def _render_bf16_oracle(params, o, d, v, lo, hi, jit, *, dist_alpha, white_bg, relu_sigma):
    """nnr.render_rays' contract on the CPU with the MLP of the bf16 mode (oracle mlp_bf16): sampling and compositing as in
    oracle/nerf_oracle.py::render."""
    R, N = o.shape[0], lo.shape[0]
    z = lo.view(1, N).expand(R, N)
    if jit is not None:
        z = lo + (hi - lo) * jit.view(R, N)
    pts = (o.unsqueeze(1) + d.unsqueeze(1) * z.unsqueeze(-1)).reshape(-1, 3)
    view = v.unsqueeze(1).expand(R, N, 3).reshape(-1, 3)
    rgb, occ = orc.mlp_bf16(params, pts, view, dist_alpha=dist_alpha, occ_activation="relu" if relu_sigma else "softplus")
    alpha = occ.view(R, N)
    if dist_alpha:
        delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full((R, 1), 1e10)], dim=-1)
        alpha = 1 - torch.exp(-1.0 * alpha * delta)
        alpha = torch.cat([alpha[:, :-1], torch.ones(R, 1)], dim=-1)
    trans = torch.cumprod(torch.cat([torch.ones(R, 1), 1.0 - alpha + orc.EPS_T], -1), -1)[:, :-1]
    w = alpha * trans
    out = (w.unsqueeze(-1) * rgb.view(R, N, 3)).sum(-2)
    dist = (w * z).sum(-1)
    if white_bg:
        out = out + (1.0 - w.sum(-1, keepdim=True))
    return out, dist


@pytest.mark.parametrize("D,R,N,dist_alpha,white_bg,relu_sigma,jittered", [
    (128, 1, 2, False, False, False, True),      # one ray, two samples: one chunk pair, three of the four waves recompute clamped chunks
    (128, 7, 17, True, False, False, True),      # flat decomposition (N % 64 != 0), dist_alpha
    (256, 33, 65, False, True, True, True),      # white background, ReLU density; 2145 samples = 17 workgroups, the last one ragged
    (256, 5, 130, True, True, False, False),     # no jitter, long rays
    (256, 8, 128, False, False, False, True),    # ray mode (N % 64 == 0, R % 4 == 0): two passes per wave, the weight stream wraps
    (128, 8, 192, True, True, False, True),      # ray mode at D = 128: three passes per wave (one panel per pass in that layout)
    (128, 129, 3, False, False, True, True),     # more rays than a wave, three samples each
])
def test_bf16_ragged_shapes_and_every_flag_against_the_bf16_oracle(D, R, N, dist_alpha, white_bg, relu_sigma, jittered):
    """The bf16 kernels (two chunks per wave, 256 samples per workgroup, 64-sample passes in ray mode) on sizes that are not
    multiples of any of those, with every rendering switch: forward to 3e-3 (3e-4 on average), every gradient tensor to 3e-2 relative L2 (the
    tolerance of test_bf16_kernels_match_the_bf16_arithmetic_oracle, which explains what is left) against the CPU oracle with the
    same arithmetic; and the forward-only kernel (no stash) returns the training kernel's outputs bit for bit."""
    import nnr
    from nnr import lib as L
    dev = torch.device("cuda")
    params, o, d, lo, hi, jit = _synthetic(D, R, N, seed=31 + R, dist_alpha=dist_alpha)
    if not jittered:
        jit = None
    g = torch.Generator().manual_seed(5)
    d_rgb, d_dist = torch.randn(R, 3, generator=g) / R, torch.randn(R, generator=g) / R
    w = [params[n + ".weight"].to(dev).requires_grad_(True) for n in L.LAYER_NAMES]
    b = [params[n + ".bias"].to(dev).requires_grad_(True) for n in L.LAYER_NAMES]
    oo, dd, vv = o.to(dev).requires_grad_(True), d.to(dev).requires_grad_(True), (-d).to(dev).requires_grad_(True)
    kw = dict(hidden=D, dist_alpha=dist_alpha, white_bg=white_bg, relu_sigma=relu_sigma, bf16=True)
    jd = jit.to(dev) if jit is not None else None
    rgb, dist, _, _ = nnr.render_rays(oo, dd, vv, lo.to(dev), hi.to(dev), jd, w, b, **kw)
    (rgb * d_rgb.to(dev)).sum().add((dist * d_dist.to(dev)).sum()).backward()
    with torch.no_grad():
        rgb_i, dist_i, _, _ = nnr.render_rays(oo, dd, vv, lo.to(dev), hi.to(dev), jd, w, b, **kw)
    assert torch.equal(rgb_i, rgb.detach()) and torch.equal(dist_i, dist.detach())

    P = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    po, pd, pv = (t.clone().requires_grad_(True) for t in (o, d, -d))
    orgb, odist = _render_bf16_oracle(P, po, pd, pv, lo, hi, jit, dist_alpha=dist_alpha, white_bg=white_bg, relu_sigma=relu_sigma)
    ((orgb * d_rgb).sum() + (odist * d_dist).sum()).backward()
    # Outputs: 3e-3 for every ray, 3e-4 on average.  Kernel and oracle add the exact bf16 x bf16 products in different orders, so
    # their fp32 pre-activations differ in the last bits; where such a value sits on a bf16 rounding boundary the NEXT layer sees an
    # input that differs by 2^-9 of its size (about one activation in 40 000: a few per cent of the samples have one somewhere in
    # their nine layers), which moves that sample's colour / density by 1e-4 .. 1e-3.  Rays of 64+ samples average it away (the
    # golden scenes agree to 1e-4); a ray of three samples shows it undamped -- measured worst case 1.6e-3 on (128, 129, 3).
    for got_o, ref_o in ((rgb, orgb), (dist, odist)):
        err = (got_o.detach().cpu() - ref_o.detach()).abs().flatten() / max(1.0, float(ref_o.detach().abs().max()))
        assert float(err.max()) <= 3e-3 and float(err.mean()) <= 3e-4, (D, R, N, float(err.max()), float(err.mean()))
    ref = {"d pts_o": po.grad, "d pts_d": pd.grad, "d view": pv.grad}
    got = {"d pts_o": oo.grad, "d pts_d": dd.grad, "d view": vv.grad}
    for i, n in enumerate(L.LAYER_NAMES):
        ref["dW " + n], ref["db " + n] = P[n + ".weight"].grad, P[n + ".bias"].grad
        got["dW " + n], got["db " + n] = w[i].grad, b[i].grad
    for k, r in ref.items():
        r = r.double()
        if float(r.abs().max()) == 0:
            assert float(got[k].abs().max()) == 0, k
            continue
        l2 = float((got[k].detach().cpu().double() - r).norm() / r.norm())
        # gradients of the sampling points / directions: a handful of samples with 2^9-fold frequencies carry them
        assert l2 <= (1e-1 if k.startswith("d ") else 3e-2), (k, D, R, N, l2)


def test_config4_eight_shards_of_4096_rays_equal_one_32768_ray_pass():
    """BASELINE.json config 4 (8 ranks x 4096 rays of one image, Ballroom settings: uniform sampling, no dist_alpha, D=256, N=128):
    the gradients of the eight shards sum to the gradients of a single 32 768-ray pass -- the data-parallel identity at the
    full size (4.2 M samples, a 38 GB workspace) -- and a handful of its rays agree with the oracle."""
    D, R, N, W = 256, 32768, 128, 8
    params, o, d, lo, hi, jit = _synthetic(D, R, N, seed=11)
    g = torch.Generator().manual_seed(6)
    d_rgb, d_dist = torch.randn(R, 3, generator=g) / R, torch.randn(R, generator=g) / R
    rgb, dist, _, full = _hip_render(params, o, d, lo, hi, jit, D, d_rgb, d_dist)
    sums = None
    per = R // W
    for k in range(W):
        sl = slice(k * per, (k + 1) * per)
        r_k, dist_k, _, g_k = _hip_render(params, o[sl], d[sl], lo, hi, jit[sl], D, d_rgb[sl], d_dist[sl])
        assert torch.equal(r_k, rgb[sl.start:sl.stop]) and torch.equal(dist_k, dist[sl.start:sl.stop])      # rays are independent
        w_k = [x.clone() for x in g_k[3:]]
        sums = w_k if sums is None else [a + b for a, b in zip(sums, w_k)]
        assert torch.allclose(g_k[0], full[0][sl], atol=1e-6) and torch.allclose(g_k[1], full[1][sl], atol=1e-5)
    for a, b in zip(sums, full[3:]):
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) / scale <= 2e-5
    sel = torch.arange(0, R, 4096)
    with torch.no_grad():
        orgb, odist, _ = trace_util.traced_render(dict(params), o[sel], d[sel], -d[sel], lo, hi, jit[sel], dist_alpha=False, white_bg=False)
    assert float((rgb[sel.cuda()].cpu() - orgb).abs().max()) <= TOL
    assert float((dist[sel.cuda()].cpu() - odist).abs().max()) <= TOL * 10
