"""Per-image losses of the first training phase (point cloud + surface reprojection).
CPU: the oracle (train_step_scope + aux_scope) against the golden vectors minted from the reference Trainer.train_step with
pc_weight = rgb_s_weight = 1 (oracle/gen_golden_aux.py).  GPU (-m gpu): the product Trainer.train_step -- render kernels,
per-image terms, every fused op in the loop -- against the same golden: loss parts to 1e-5, pose / distortion gradients to
1e-4 of their scale."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "aux_terms.npz"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
H, W, R, N, N_CAMS = 48, 64, 64, 32, 6


def _inp(name):
    return {k: torch.from_numpy(GOLD[f"{name}.in.{k}"]) for k in
            ("K", "img", "ref_img", "dpt", "ref_dpt", "pose_r", "pose_t", "scales", "shifts")}


@pytest.mark.parametrize("name", ["mid", "last"])
def test_oracle_aux_scope_matches_reference_golden(name):
    import nerf_oracle as orc
    inp = _inp(name)
    cam, ref = int(GOLD[f"{name}.cam"]), int(GOLD[f"{name}.ref"])
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
    aux, l_pc, l_rgbs = orc.aux_scope(leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], cam, ref, inp["K"],
                                      inp["dpt"].unsqueeze(1), inp["ref_dpt"].unsqueeze(1), inp["img"], inp["ref_img"])
    aux.backward()
    np.testing.assert_allclose(l_pc.item(), float(GOLD[f"{name}.out.loss_pc"]), rtol=0, atol=1e-6)
    np.testing.assert_allclose(l_rgbs.item(), float(GOLD[f"{name}.out.loss_rgb_s"]), rtol=0, atol=1e-6)
    for k in leaves:
        g = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(inp[k])
        np.testing.assert_allclose(g.numpy(), GOLD[f"{name}.gaux.{k}"], rtol=0, atol=2e-6, err_msg=k)


def test_oracle_aux_scope_with_ssim_matches_reference_golden():
    """with_ssim: the reference's recorded step (mid_ssim.*) differs from the plain one (mid.*) only in the surface re-projection term, so
    its gradient minus the render part of the plain step (mid.g - mid.gaux: same weights, same draws) is the reference's gradient of the
    per-image terms with SSIM."""
    import nerf_oracle as orc
    inp = _inp("mid")
    cam, ref = int(GOLD["mid.cam"]), int(GOLD["mid.ref"])
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
    aux, l_pc, l_rgbs = orc.aux_scope(leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], cam, ref, inp["K"],
                                      inp["dpt"].unsqueeze(1), inp["ref_dpt"].unsqueeze(1), inp["img"], inp["ref_img"], with_ssim=True)
    aux.backward()
    np.testing.assert_allclose(l_pc.item(), float(GOLD["mid_ssim.out.loss_pc"]), rtol=0, atol=1e-6)
    np.testing.assert_allclose(l_rgbs.item(), float(GOLD["mid_ssim.out.loss_rgb_s"]), rtol=0, atol=1e-6)
    assert abs(l_rgbs.item() - float(GOLD["mid.out.loss_rgb_s"])) > 1e-3          # and it is not the plain term
    for k in leaves:
        g = leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(inp[k])
        want = GOLD[f"mid_ssim.g.{k}"] - (GOLD[f"mid.g.{k}"] - GOLD[f"mid.gaux.{k}"])
        np.testing.assert_allclose(g.numpy(), want, rtol=0, atol=5e-6, err_msg=k)


def _trainer(inp, dev, adam=False, rendering_overrides=None, focal=None, **training_overrides):
    import model as mdl
    if True:
        cfg = {
            'model': {'hidden_dim': 128, 'pos_enc_levels': 10, 'dir_enc_levels': 4, 'occ_activation': 'softplus'},
            'rendering': {'type': 'nope_nerf', 'n_max_network_queries': 64000, 'white_background': False, 'radius': 4.0,
                          'num_points': N, 'depth_range': [0.01, 10], 'dist_alpha': False, 'use_ray_dir': True,
                          'normalise_ray': True, 'normal_loss': False, 'sample_option': 'uniform', 'outside_steps': 0},
            'depth': {'type': 'None'}, 'distortion': {'fix_scaleN': True},
            'training': {
                'type': 'nope_nerf', 'n_training_points': R, 'vis_geo': False, 'detach_gt_depth': False, 'pc_ratio': 4,
                'match_method': 'dense', 'shift_first': False, 'detach_ref_img': True, 'scale_pcs': True,
                'detach_rgbs_scale': False, 'vis_reprojection_every': 10 ** 9, 'nearest_limit': 0.01, 'annealing_epochs': 2000,
                'rgb_weight': [1.0, 1.0], 'depth_weight': [0.04, 0.0], 'pc_weight': [1.0, 0.0], 'rgb_s_weight': [1.0, 0.0],
                'depth_consistency_weight': [0.0, 0.0], 'weight_dist_2nd_loss': [0.0, 0.0], 'weight_dist_1st_loss': [0.0, 0.0],
                'depth_loss_type': 'l1', 'with_ssim': False, 'with_auto_mask': False},
        }
    cfg['training'].update(training_overrides)
    cfg['rendering'].update(rendering_overrides or {})
    net = mdl.OfficialStaticNerf(cfg)
    wts = np.load(os.path.join(HERE, "golden", "weights_d128.npz"))
    net.load_state_dict({k: torch.from_numpy(wts[k]) for k in wts.files})
    model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device=dev), cfg, device=dev)
    pose = mdl.LearnPose(N_CAMS, True, True, cfg).to(dev)
    dist = mdl.Learn_Distortion(N_CAMS, True, True, cfg).to(dev)
    with torch.no_grad():
        pose.r.copy_(inp["pose_r"]); pose.t.copy_(inp["pose_t"])
        dist.global_scales.copy_(inp["scales"]); dist.global_shifts.copy_(inp["shifts"])
    # lr-0 SGD for the single-step gradient checks; the reference's three Adam instances (configs/default.yaml:79-82) otherwise
    opt = (lambda m, lr: torch.optim.Adam(m.parameters(), lr=lr)) if adam else (lambda m, lr: torch.optim.SGD(m.parameters(), lr=0.0))
    extra = dict(optimizer_focal=opt(focal, 1e-3), focal_net=focal) if focal is not None else {}
    tr = mdl.Trainer(model, opt(model, 1e-3), cfg['training'], device=dev, optimizer_pose=opt(pose, 5e-4), pose_param_net=pose,
                     optimizer_distortion=opt(dist, 5e-4), distortion_net=dist, **extra)
    return tr, pose, dist


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mid", "last"])
def test_trainer_step_with_per_image_losses_matches_reference_golden(name, monkeypatch):
    dev = torch.device("cuda")
    inp = _inp(name)
    cam, ref = int(GOLD[f"{name}.cam"]), int(GOLD[f"{name}.ref"])
    tr, pose, dist = _trainer(inp, dev)
    # replay the reference's draws (its generator is the CPU one): pixel permutation and jitter
    ray_idx = torch.from_numpy(GOLD[f"{name}.ray_idx"])
    jitter = torch.from_numpy(GOLD[f"{name}.jitter"])
    monkeypatch.setattr(torch, "randperm", lambda n, device=None, **kw: torch.cat([ray_idx, torch.zeros(n - R, dtype=torch.int64)]).to(device))
    real_rand = torch.rand
    monkeypatch.setattr(torch, "rand", lambda *s, device=None, **kw: jitter.to(device) if tuple(s) == (1, R, N) else real_rand(*s, device=device, **kw))
    data = {"img": inp["img"].to(dev), "img.idx": cam, "img.dpt": inp["dpt"].to(dev), "img.camera_mat": inp["K"].to(dev),
            "img.scale_mat": torch.eye(4).unsqueeze(0).to(dev), "img.ref_imgs": inp["ref_img"].to(dev),
            "img.ref_dpts": inp["ref_dpt"].to(dev), "img.ref_idxs": ref}
    ld = tr.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
    for k in ("loss", "loss_pc", "loss_rgb_s", "loss_rgb", "loss_depth"):
        np.testing.assert_allclose(float(ld[k]), float(GOLD[f"{name}.out.{k}"]), rtol=0, atol=1e-5, err_msg=k)
    got = {"pose_r": pose.r.grad, "pose_t": pose.t.grad, "scales": dist.global_scales.grad, "shifts": dist.global_shifts.grad}
    for k, g in got.items():
        ref_g = GOLD[f"{name}.g.{k}"]
        g = g.cpu().numpy() if g is not None else np.zeros_like(ref_g)
        scale = max(1.0, float(np.abs(ref_g).max()))
        assert float(np.abs(g - ref_g).max()) / scale <= 1e-4, (k, float(np.abs(g - ref_g).max()), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("name,flags", [
    ("mid", dict(scale_pcs=False)), ("mid", dict(detach_rgbs_scale=True)), ("mid", dict(pc_weight=[0.0, 0.0])),
    ("mid", dict(rgb_s_weight=[0.0, 0.0])), ("last", dict(detach_rgbs_scale=True, scale_pcs=False)), ("mid", dict(shift_first=True)),
    ("mid", dict(with_ssim=True)), ("last", dict(with_ssim=True, detach_rgbs_scale=True)), ("last", dict(with_ssim=True, pc_weight=[0.0, 0.0])),
])
def test_fused_per_image_terms_match_oracle_for_every_flag(name, flags):
    """Only the per-image terms (render weights 0), every configuration switch of reference training.py:325-357, against
    the oracle restatement (itself pinned to the reference for the default switches)."""
    import nerf_oracle as orc
    dev = torch.device("cuda")
    inp = _inp(name)
    cam, ref = int(GOLD[f"{name}.cam"]), int(GOLD[f"{name}.ref"])
    over = dict(rgb_weight=[0.0, 0.0], depth_weight=[0.0, 0.0])
    over.update(flags)
    tr, pose, dist = _trainer(inp, dev, **over)
    data = {"img": inp["img"].to(dev), "img.idx": cam, "img.dpt": inp["dpt"].to(dev), "img.camera_mat": inp["K"].to(dev),
            "img.scale_mat": torch.eye(4).unsqueeze(0).to(dev), "img.ref_imgs": inp["ref_img"].to(dev),
            "img.ref_dpts": inp["ref_dpt"].to(dev), "img.ref_idxs": ref}
    ld = tr.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
    kw = dict(pc_weight=over.get("pc_weight", [1.0])[0], rgb_s_weight=over.get("rgb_s_weight", [1.0])[0],
              scale_pcs=over.get("scale_pcs", True), detach_rgbs_scale=over.get("detach_rgbs_scale", False),
              shift_first=over.get("shift_first", False), with_ssim=over.get("with_ssim", False))
    aux, l_pc, l_rgbs = orc.aux_scope(leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], cam, ref, inp["K"],
                                      inp["dpt"].unsqueeze(1), inp["ref_dpt"].unsqueeze(1), inp["img"], inp["ref_img"], **kw)
    aux.backward()
    np.testing.assert_allclose(float(ld["loss_pc"]), float(l_pc), rtol=0, atol=1e-5)
    np.testing.assert_allclose(float(ld["loss_rgb_s"]), float(l_rgbs), rtol=0, atol=1e-5)
    np.testing.assert_allclose(float(ld["loss"]), float(aux), rtol=0, atol=1e-5)
    got = {"pose_r": pose.r.grad, "pose_t": pose.t.grad, "scales": dist.global_scales.grad, "shifts": dist.global_shifts.grad}
    for k, g in got.items():
        ref_g = leaves[k].grad.numpy() if leaves[k].grad is not None else np.zeros_like(inp[k].numpy())
        g = g.cpu().numpy() if g is not None else np.zeros_like(ref_g)
        scale = max(1.0, float(np.abs(ref_g).max()))
        assert float(np.abs(g - ref_g).max()) / scale <= 1e-4, (k, flags, float(np.abs(g - ref_g).max()), scale)


@pytest.mark.gpu
@pytest.mark.parametrize("name,flags", [
    ("mid", {}), ("mid", dict(with_ssim=True)), ("last", dict(detach_rgbs_scale=True)), ("last", dict(with_ssim=True, scale_pcs=False)),
    ("mid", dict(pc_weight=[0.0, 0.0], with_ssim=True)), ("mid", dict(rgb_s_weight=[0.0, 0.0])),
])
def test_fused_per_image_terms_with_a_learnable_focal_match_oracle(name, flags):
    """optimizer_focal (reference training.py:247-252): camera_mat is built from the focal parameters, so the per-image terms also
    return d/dK (the re-projection) and d/dKinv (both back-projections).  fx / fy gradients and everything else against the oracle with
    the same camera_mat as an autograd leaf chain."""
    import model as mdl
    import nerf_oracle as orc
    dev = torch.device("cuda")
    inp = _inp(name)
    cam, ref = int(GOLD[f"{name}.cam"]), int(GOLD[f"{name}.ref"])
    f0 = [float(inp["K"][0, 0, 0]) * 1.07, float(-inp["K"][0, 1, 1]) * 0.94]      # not the data's intrinsics: a focal mid-optimisation
    focal = mdl.LearnFocal(True, False, order=2, init_focal=f0).to(dev)
    over = dict(rgb_weight=[0.0, 0.0], depth_weight=[0.0, 0.0])
    over.update(flags)
    tr, pose, dist = _trainer(inp, dev, focal=focal, **over)
    data = {"img": inp["img"].to(dev), "img.idx": cam, "img.dpt": inp["dpt"].to(dev), "img.camera_mat": inp["K"].to(dev),
            "img.scale_mat": torch.eye(4).unsqueeze(0).to(dev), "img.ref_imgs": inp["ref_img"].to(dev),
            "img.ref_dpts": inp["ref_dpt"].to(dev), "img.ref_idxs": ref}
    ld = tr.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
    focal_cpu = mdl.LearnFocal(True, False, order=2, init_focal=f0)
    fxfy = focal_cpu(0)
    pad = torch.zeros(4)
    k_mat = torch.cat([fxfy[0:1], pad, -fxfy[1:2], pad, -torch.ones(1), pad, torch.ones(1)]).view(1, 4, 4)     # training.py:248-252
    kw = dict(pc_weight=over.get("pc_weight", [1.0])[0], rgb_s_weight=over.get("rgb_s_weight", [1.0])[0],
              scale_pcs=over.get("scale_pcs", True), detach_rgbs_scale=over.get("detach_rgbs_scale", False),
              with_ssim=over.get("with_ssim", False))
    aux, l_pc, l_rgbs = orc.aux_scope(leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], cam, ref, k_mat,
                                      inp["dpt"].unsqueeze(1), inp["ref_dpt"].unsqueeze(1), inp["img"], inp["ref_img"], **kw)
    aux.backward()
    np.testing.assert_allclose(float(ld["loss_pc"]), float(l_pc), rtol=0, atol=1e-5)
    np.testing.assert_allclose(float(ld["loss_rgb_s"]), float(l_rgbs), rtol=0, atol=1e-5)
    got = {"fx": focal.fx.grad, "fy": focal.fy.grad, "pose_r": pose.r.grad, "pose_t": pose.t.grad, "scales": dist.global_scales.grad,
           "shifts": dist.global_shifts.grad}
    want = dict(leaves, fx=focal_cpu.fx, fy=focal_cpu.fy)
    assert float(focal_cpu.fx.grad.abs()) > 1e-4 and float(focal_cpu.fy.grad.abs()) > 1e-4      # the case does exercise d/dK
    for k, g in got.items():
        ref_g = want[k].grad.numpy() if want[k].grad is not None else np.zeros(tuple(want[k].shape), np.float32)
        g = g.cpu().numpy() if g is not None else np.zeros_like(ref_g)
        scale = max(1.0, float(np.abs(ref_g).max()))
        assert float(np.abs(g - ref_g).max()) / scale <= 1e-4, (k, flags, g, ref_g)


@pytest.mark.gpu
@pytest.mark.parametrize("with_ssim", [False, True])
def test_fused_path_is_the_one_that_runs(monkeypatch, with_ssim):
    """The trainer must reach libnnr.so for the per-image terms on the GPU (no silent torch fallback), with_ssim included."""
    from nnr import aux as nnr_aux
    calls = []
    real = nnr_aux.aux_terms
    monkeypatch.setattr(nnr_aux, "aux_terms", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    dev = torch.device("cuda")
    inp = _inp("mid")
    tr, _, _ = _trainer(inp, dev, with_ssim=with_ssim)
    data = {"img": inp["img"].to(dev), "img.idx": 2, "img.dpt": inp["dpt"].to(dev), "img.camera_mat": inp["K"].to(dev),
            "img.scale_mat": torch.eye(4).unsqueeze(0).to(dev), "img.ref_imgs": inp["ref_img"].to(dev),
            "img.ref_dpts": inp["ref_dpt"].to(dev), "img.ref_idxs": 3}
    tr.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
    assert calls == [1]


def _ssim_step(dev, monkeypatch):
    inp = _inp("mid")
    cam, ref = int(GOLD["mid.cam"]), int(GOLD["mid.ref"])
    tr, pose, dist = _trainer(inp, dev, with_ssim=True)
    ray_idx, jitter = torch.from_numpy(GOLD["mid.ray_idx"]), torch.from_numpy(GOLD["mid.jitter"])
    monkeypatch.setattr(torch, "randperm", lambda n, device=None, **kw: torch.cat([ray_idx, torch.zeros(n - R, dtype=torch.int64)]).to(device))
    real_rand = torch.rand
    monkeypatch.setattr(torch, "rand", lambda *s, device=None, **kw: jitter.to(device) if tuple(s) == (1, R, N) else real_rand(*s, device=device, **kw))
    data = {"img": inp["img"].to(dev), "img.idx": cam, "img.dpt": inp["dpt"].to(dev), "img.camera_mat": inp["K"].to(dev),
            "img.scale_mat": torch.eye(4).unsqueeze(0).to(dev), "img.ref_imgs": inp["ref_img"].to(dev),
            "img.ref_dpts": inp["ref_dpt"].to(dev), "img.ref_idxs": ref}
    ld = tr.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
    for k in ("loss", "loss_pc", "loss_rgb_s", "loss_rgb", "loss_depth"):
        np.testing.assert_allclose(float(ld[k].detach()), float(GOLD[f"mid_ssim.out.{k}"]), rtol=0, atol=1e-5, err_msg=k)
    got = {"pose_r": pose.r.grad, "pose_t": pose.t.grad, "scales": dist.global_scales.grad, "shifts": dist.global_shifts.grad}
    for k, g in got.items():
        ref_g = GOLD[f"mid_ssim.g.{k}"]
        g = g.cpu().numpy() if g is not None else np.zeros_like(ref_g)
        assert float(np.abs(g - ref_g).max()) / max(1.0, float(np.abs(ref_g).max())) <= 1e-4, k


def test_step_with_ssim_term_matches_reference_on_the_cpu_stand_in(monkeypatch):
    """training.with_ssim: True (the surface re-projection term mixes in the 3x3 SSIM dissimilarity, losses.py:150-157,222-252)."""
    import oracle_backend
    from model import rendering
    monkeypatch.setattr(rendering.nnr, "render_rays", oracle_backend.render_rays)
    _ssim_step(torch.device("cpu"), monkeypatch)


@pytest.mark.gpu
def test_step_with_ssim_term_matches_reference_on_the_hip_kernels(monkeypatch):
    _ssim_step(torch.device("cuda"), monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("name,cam_ref,flags", [
    ("mid", None, {}), ("last", None, {}),
    ("mid", None, dict(detach_ref_img=False)),                          # gradients into the reference camera's pose and distortion rows
    ("last", None, dict(detach_ref_img=False, shift_first=True)),       # the roles swapped, (depth + shift) * scale
    ("mid", (2, 3), dict(detach_ref_img=False)),                        # the reference camera is the gauge (its scale is the constant 1)
    ("mid", None, dict(scale_pcs=False, detach_rgbs_scale=True)), ("mid", None, dict(with_ssim=True, detach_ref_img=False)),
    ("mid", None, dict(pc_weight=[0.0, 0.0])), ("last", None, dict(rgb_s_weight=[0.0, 0.0], detach_ref_img=False)),
])
def test_first_phase_step_on_the_fused_front_end_equals_the_separate_launches(name, cam_ref, flags, monkeypatch):
    """Round 5: the first-phase step takes the fused front end too -- the frame pair (reference pose, inverses, relative transform, the
    reference's distortion) rides along in nnr_step_rays_fwd / _bwd and the per-image kernels distort their own sampled depths
    (nnr_aux_terms_*, `aff`).  Against the same step on the separate launches (training.fuse_pair: False): the same losses, the same
    gradients of poses, distortions and network, also where the reference side is not detached, for the last camera (roles swapped), for
    a reference camera that is the gauge, and under every per-image switch."""
    dev = torch.device("cuda")
    inp = _inp(name)
    cam, ref = cam_ref if cam_ref is not None else (int(GOLD[f"{name}.cam"]), int(GOLD[f"{name}.ref"]))
    ray_idx, jitter = torch.from_numpy(GOLD[f"{name}.ray_idx"]), torch.from_numpy(GOLD[f"{name}.jitter"])
    monkeypatch.setattr(torch, "randperm", lambda n, device=None, **kw: torch.cat([ray_idx, torch.zeros(n - R, dtype=torch.int64)]).to(device))
    real_rand = torch.rand
    monkeypatch.setattr(torch, "rand", lambda *s, device=None, **kw: jitter.to(device) if tuple(s) == (1, R, N) else real_rand(*s, device=device, **kw))
    from nnr import camera
    out = {}
    for fused in (False, True):
        calls = []
        real_step = camera.step_rays
        monkeypatch.setattr(camera, "step_rays", lambda *a, **k: (calls.append(k.get("ref", -1)), real_step(*a, **k))[1])
        tr, pose, dist = _trainer(inp, dev, fuse_pair=fused, **flags)
        data = {"img": inp["img"].to(dev), "img.idx": cam, "img.dpt": inp["dpt"].to(dev), "img.camera_mat": inp["K"].to(dev),
                "img.scale_mat": torch.eye(4).unsqueeze(0).to(dev), "img.ref_imgs": inp["ref_img"].to(dev),
                "img.ref_dpts": inp["ref_dpt"].to(dev), "img.ref_idxs": ref}
        ld = tr.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
        monkeypatch.setattr(camera, "step_rays", real_step)
        assert calls == ([ref] if fused else []), calls            # the fused step went through the pair front end, the other one did not
        grads = {"pose_r": pose.r.grad, "pose_t": pose.t.grad, "scales": dist.global_scales.grad, "shifts": dist.global_shifts.grad}
        grads.update({"net." + k: p.grad for k, p in tr.model.named_parameters()})
        out[fused] = ({k: float(ld[k]) for k in ("loss", "loss_pc", "loss_rgb_s", "loss_rgb", "loss_depth")},
                      {k: (None if g is None else g.detach().clone()) for k, g in grads.items()})
    (l0, g0), (l1, g1) = out[False], out[True]
    for k in l0:
        assert abs(l0[k] - l1[k]) <= 2e-6 * max(1.0, abs(l0[k])), (k, l0[k], l1[k])
    for k in g0:
        assert (g0[k] is None) == (g1[k] is None), k
        if g0[k] is None:
            continue
        scale = max(1e-6, float(g0[k].abs().max()))
        assert float((g0[k] - g1[k]).abs().max()) <= 2e-5 * scale, (k, float((g0[k] - g1[k]).abs().max()) / scale)
    if flags.get("detach_ref_img") is False and flags.get("pc_weight", [1.0])[0] != 0.0:
        assert float(g1["pose_r"][ref].abs().max()) > 0.0           # the reference camera's pose row is live
