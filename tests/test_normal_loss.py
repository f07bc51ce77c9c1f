"""rendering.normal_loss -- the normal-consistency vector of reference model/rendering.py:133-143 through
OfficialStaticNerf.gradient (official_nerf.py:46-58, second-order autograd).  The fused kernels render; the normal term runs in
stock autograd over the same parameters.  Golden: tests/golden/normal_loss_d128.npz from the REFERENCE (oracle/gen_golden_normal.py):
out['normal'] and the gradients of render loss + sum(out['normal']) w.r.t. every parameter, tolerance 1e-4 (gradients normalised by
the golden tensor's max-abs when that exceeds 1)."""
import os

import numpy as np
import pytest
import torch

import golden_util as gu
import nerf_oracle as orc

GOLD = np.load(os.path.join(gu.GOLDEN, "normal_loss_d128.npz"))
WEIGHTS = np.load(os.path.join(gu.GOLDEN, "weights_d128.npz"))
R, N, H, W, CAM, M = (int(GOLD[k]) for k in ("cfg.R", "cfg.N", "cfg.h", "cfg.w", "cfg.cam", "cfg.M"))


def _inp():
    return {k[3:]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith("in.")}


def _check_grads(got):
    for k in GOLD.files:
        if not k.startswith("g."):
            continue
        ref = GOLD[k].astype(np.float64)
        g = got[k[2:]]
        g = np.zeros_like(ref) if g is None else g.detach().cpu().double().numpy()
        err = np.abs(g - ref).max() / max(1.0, np.abs(ref).max())
        assert err <= 1e-4, (k, err)


def test_oracle_normal_term_matches_reference_golden():
    t = _inp()
    params = {k: torch.from_numpy(WEIGHTS[k]).clone().requires_grad_(True) for k in WEIGHTS.files}
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
    rcfg = {"num_points": N, "dist_alpha": False, "sample_option": "uniform", "depth_range": [0.01, 10], "normalise_ray": True,
            "white_background": False, "use_ray_dir": True, "normal_loss": True, "occ_activation": "softplus"}
    loss, out = orc.train_step_scope(params, leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], CAM, t["K"],
                                     t["depth_img"], t["img"], (H, W), t["ray_idx"], t["jitter"], rcfg, normal_noise=t["noise"])
    (loss + out["normal"].sum()).backward()
    assert out["normal"].shape == (M,)
    np.testing.assert_allclose(out["normal"].detach().numpy(), GOLD["out.normal"], rtol=0, atol=1e-6)
    got = {"w." + k: v.grad for k, v in params.items()}
    got.update({k: v.grad for k, v in leaves.items()})
    _check_grads(got)


def _product_step(dev, monkeypatch, world=1, rank=0):
    """model.Renderer with normal_loss on, fed the reference's draws; returns (out, grads of render loss + sum(normal))."""
    import model as mdl
    from model.common import arange_pixels
    from model.losses import Loss
    from test_host_logic import make_cfg
    t = _inp()
    cfg = make_cfg(128, num_points=N, normal_loss=True)
    net = mdl.OfficialStaticNerf(cfg)
    net.load_state_dict({k: torch.from_numpy(WEIGHTS[k]) for k in WEIGHTS.files})
    model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device=dev), cfg, device=dev)
    pose = mdl.LearnPose(gu.N_CAMS, True, True, cfg).to(dev)
    dist = mdl.Learn_Distortion(gu.N_CAMS, True, True, cfg).to(dev)
    with torch.no_grad():
        pose.r.copy_(t["pose_r"]); pose.t.copy_(t["pose_t"])
        dist.global_scales.copy_(t["scales"]); dist.global_shifts.copy_(t["shifts"])
    ray_idx = t["ray_idx"].to(dev)
    c2w = pose(CAM)
    world_mat = (torch.inverse(c2w) if dev.type == "cpu" else mdl.training.camera.inverse4(c2w)).unsqueeze(0)
    sc, sh = dist(CAM)
    depth_in = t["depth_img"].to(dev) * sc + sh
    p = arange_pixels((H, W), device=dev)[1][:, ray_idx]
    draws = [t["jitter"].to(dev)]
    monkeypatch.setattr(torch, "rand", lambda *a, **k: draws[0])
    monkeypatch.setattr(torch, "rand_like", lambda x, **k: t["noise"].to(dev))
    out = model(p, ray_idx, t["K"].to(dev), world_mat, torch.eye(4, device=dev)[None], 'nope_nerf', it=0, eval_mode=False,
                depth_img=depth_in, add_noise=True, img_size=(H, W))
    rgb_gt = t["img"].to(dev).view(1, 3, H * W).permute(0, 2, 1)[:, ray_idx]
    crit = Loss({'depth_loss_type': 'l1'})
    loss = crit.get_rgb_full_loss(out['rgb'], rgb_gt, 'l1') + 0.04 * crit.get_depth_loss(out['depth_pred'], out['depth_gt']) + out['normal'].sum()
    loss.backward()
    grads = {"w." + k: v.grad for k, v in net.named_parameters()}
    grads.update(pose_r=pose.r.grad, pose_t=pose.t.grad, scales=dist.global_scales.grad, shifts=dist.global_shifts.grad)
    return out, grads


def test_normal_term_on_the_cpu_stand_in_matches_reference_golden(monkeypatch):
    import oracle_backend
    from model import rendering
    monkeypatch.setattr(rendering.nnr, "render_rays", oracle_backend.render_rays)
    out, grads = _product_step(torch.device("cpu"), monkeypatch)
    # not 1e-6: the host glue forms the rays with ONE pixel->world matrix where the reference chains three inverses; the 1e-7
    # difference in the surface points is amplified by the 2^9 encoding frequency inside d(sigma)/dp
    np.testing.assert_allclose(out["normal"].detach().numpy(), GOLD["out.normal"], rtol=0, atol=1e-4)
    _check_grads(grads)


# Conditioning (measured with the oracle on this very case): scaling the mono-depth map by 1 + 1.2e-7 -- ONE ulp -- moves out['normal'] by
# up to 1.5e-3 and the weight gradients of sum(normal) by 1.9e-2 of their largest entry: the MLP is piecewise linear, so d(sigma)/dp is
# piecewise CONSTANT in p and the term jumps whenever a surface point crosses a ReLU boundary of a 2^9-frequency encoding.  No two
# devices (nor two BLAS libraries) agree on it to 1e-4 -- the reference on a GPU would not reproduce its own CPU numbers either.  The
# GPU tests therefore pin (a) the end-to-end values against the reference golden at the tolerance the conditioning allows, and (b)
# the double-backward machinery itself tightly, at IDENTICAL points against an fp64 evaluation of the same formula.
COND_TOL_NORMAL, COND_TOL_GRAD = 2e-2, 1e-1


@pytest.mark.gpu
def test_normal_term_with_the_hip_render_matches_reference_golden(monkeypatch):
    """normal_loss: True RUNS on the GPU: fused kernels for the render, stock autograd (rocBLAS GEMMs, double backward) for the
    2 M surface points."""
    out, grads = _product_step(torch.device("cuda"), monkeypatch)
    assert out["normal"].is_cuda and out["normal"].shape == (M,)
    np.testing.assert_allclose(out["rgb"].detach().cpu().numpy(), GOLD["out.rgb"], rtol=0, atol=1e-4)     # the render itself: fp32 bar
    err = np.abs(out["normal"].detach().cpu().numpy() - GOLD["out.normal"])
    print("normal term on the GPU vs reference golden: max %.2e, median %.2e" % (err.max(), np.median(err)))
    assert err.max() <= COND_TOL_NORMAL and np.median(err) <= 1e-4
    worst = 0.0
    for k in GOLD.files:
        if k.startswith("g."):
            ref = GOLD[k].astype(np.float64)
            g = grads[k[2:]]
            g = np.zeros_like(ref) if g is None else g.detach().cpu().double().numpy()
            worst = max(worst, np.abs(g - ref).max() / max(1.0, np.abs(ref).max()))
    print("gradients of render loss + sum(normal): worst error / max = %.2e" % worst)
    assert worst <= COND_TOL_GRAD


@pytest.mark.gpu
def test_density_gradient_on_the_gpu_matches_fp64_at_identical_points():
    """OfficialStaticNerf.gradient on cuda (the stock-autograd route the normal term takes) against the same formula in fp64 on the
    CPU at the SAME points: first derivative AND its derivative with respect to the weights (the double backward), 1e-4 of scale."""
    import model as mdl
    from test_host_logic import make_cfg
    dev = torch.device("cuda")
    net = mdl.OfficialStaticNerf(make_cfg(128)).to(dev)
    net.load_state_dict({k: torch.from_numpy(WEIGHTS[k]) for k in WEIGHTS.files})
    p = torch.randn(64, 3, generator=torch.Generator().manual_seed(5))
    g = net.gradient(p.to(dev), 0)
    probe = torch.randn(64, 1, 3, generator=torch.Generator().manual_seed(6))
    (g * probe.to(dev)).sum().backward()
    params = {k: torch.from_numpy(WEIGHTS[k]).double().requires_grad_(True) for k in WEIGHTS.files}
    ref = orc.density_gradient(params, p.double())
    (ref * probe.double()).sum().backward()
    assert float((g.detach().cpu().double() - ref.detach()).abs().max()) <= 1e-4 * float(ref.detach().abs().max())
    for name, prm in net.named_parameters():
        r = params[name].grad
        if r is None:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, name
            continue
        err = float((prm.grad.detach().cpu().double() - r).abs().max()) / max(1.0, float(r.abs().max()))
        assert err <= 1e-4, (name, err)


def test_mlp_gradient_method_equals_the_oracle():
    """OfficialStaticNerf.gradient / infer_occ (the reference's public methods, official_nerf.py:46-67) in isolation."""
    import model as mdl
    from test_host_logic import make_cfg
    net = mdl.OfficialStaticNerf(make_cfg(128))
    net.load_state_dict({k: torch.from_numpy(WEIGHTS[k]) for k in WEIGHTS.files})
    p = torch.randn(17, 3, generator=torch.Generator().manual_seed(3))
    g = net.gradient(p.clone(), 0)
    params = {k: torch.from_numpy(WEIGHTS[k]) for k in WEIGHTS.files}
    ref = orc.density_gradient(params, p.clone())
    assert g.shape == (17, 1, 3) and g.requires_grad
    assert float((g - ref).abs().max()) <= 1e-5 * float(ref.abs().max())       # same function, autograd sums in another order
    x, raw = net.infer_occ(p)
    assert x.shape == (17, 128) and raw.shape == (17, 1)
