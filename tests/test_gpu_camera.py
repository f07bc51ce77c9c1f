"""GPU tests (-m gpu) of the camera front end / loss-head kernels (nnr_camera.hip) against plain PyTorch fp32 autograd
of the same formulas, and of the whole Trainer.train_step (every fused op in the loop) against the reference golden."""
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, b, tol=1e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    err = float((a - b).abs().max()) / scale if b.numel() else 0.0
    assert err <= tol, err


@pytest.mark.parametrize("zero", [False, True])
def test_se3_exp_matches_torch(zero):
    from model.common import make_c2w
    from nnr import camera
    g = torch.Generator().manual_seed(0)
    r = (torch.zeros(5, 3) if zero else 0.3 * torch.randn(5, 3, generator=g))
    t = torch.randn(5, 3, generator=g)
    G = torch.randn(4, 4, generator=g)
    r1, t1 = r.clone().requires_grad_(True), t.clone().requires_grad_(True)
    ref = make_c2w(r1[2], t1[2])
    (ref * G).sum().backward()
    r2, t2 = r.to(DEV).requires_grad_(True), t.to(DEV).requires_grad_(True)
    out = camera.se3_exp(r2, t2, 2)
    (out * G.to(DEV)).sum().backward()
    close(out, ref, 1e-6)
    close(r2.grad, r1.grad, 2e-5)
    close(t2.grad, t1.grad, 1e-6)
    assert torch.isfinite(r2.grad).all()


def test_inverse4_matches_torch():
    from nnr import camera
    g = torch.Generator().manual_seed(1)
    a = torch.randn(3, 4, 4, generator=g) + 3 * torch.eye(4)
    G = torch.randn(3, 4, 4, generator=g)
    a1 = a.clone().requires_grad_(True)
    ref = torch.inverse(a1)
    (ref * G).sum().backward()
    a2 = a.to(DEV).requires_grad_(True)
    out = camera.inverse4(a2)
    (out * G.to(DEV)).sum().backward()
    close(out, ref, 5e-5)        # cofactor formula vs LU: different rounding, same answer
    close(a2.grad, a1.grad, 5e-5)


def torch_ray_setup(pixels, depth, K, W, S, normalise, use_dir):
    from model.common import pixel_to_world_matrix
    m = pixel_to_world_matrix(K, W, S)[0]
    R = pixels.shape[1]
    pix_h = torch.cat([pixels[0], torch.ones(R, 1)], dim=-1)
    ray = pix_h @ m[:3, :3].t()
    norm = ray.norm(2, -1)
    d_gt = (ray * depth[0]).norm(2, -1)
    if normalise:
        ray = ray / norm.unsqueeze(-1)
    else:
        d_gt = d_gt / norm
    mask = torch.isfinite(d_gt) & (d_gt != 0)
    view = -ray if use_dir else torch.ones_like(ray)
    return m[:3, 3].unsqueeze(0).expand(R, 3), ray, view, norm, d_gt, mask


@pytest.mark.parametrize("normalise,use_dir", [(True, True), (False, True), (True, False)])
def test_ray_setup_matches_torch(normalise, use_dir):
    from model.common import make_c2w
    from nnr import camera
    g = torch.Generator().manual_seed(2)
    R = 300
    pixels = torch.rand(1, R, 2, generator=g) * 2 - 1
    depth = 1 + 2 * torch.rand(1, R, 1, generator=g)
    depth[0, 5] = 0.0
    K = torch.diag(torch.tensor([1.4, -2.49, -1.0, 1.0]))[None]
    W = torch.inverse(make_c2w(0.2 * torch.randn(3, generator=g), torch.randn(3, generator=g)))[None]
    S = (torch.eye(4) + 0.05 * torch.randn(4, 4, generator=g))[None]
    ups = [torch.randn(R, 3, generator=g) for _ in range(3)] + [torch.randn(R, generator=g) for _ in range(2)]

    def run(fn, dev):
        leaves = [t.clone().to(dev).requires_grad_(True) for t in (depth, K, W, S)]
        out = fn(pixels.to(dev), *leaves, normalise, use_dir)
        loss = sum((o * u.to(dev)).sum() for o, u in zip(out[:5], ups))
        loss.backward()
        return out, [l.grad for l in leaves]

    ref, gref = run(torch_ray_setup, "cpu")
    out, gout = run(camera.ray_setup, DEV)
    for a, b in zip(out[:5], ref[:5]):
        close(a, b, 1e-5)
    assert torch.equal(out[5].cpu(), ref[5])
    for a, b in zip(gout, gref):
        close(a, b, 5e-5)
    # non-finite depth: masked out, no NaN anywhere in the gradients (the reference's autograd gives NaN here)
    depth2 = depth.clone()
    depth2[0, 7] = float("inf")
    d = depth2.to(DEV).requires_grad_(True)
    o = camera.ray_setup(pixels.to(DEV), d, K.to(DEV), W.to(DEV), S.to(DEV), normalise, use_dir)
    assert not bool(o[5][7]) and not bool(o[5][5])
    (o[1].sum() + (o[4] * o[5]).nan_to_num(posinf=0.0).sum()).backward()


@pytest.mark.parametrize("dst,src", [((60, 80), (30, 40)), ((540, 960), (384, 672)), ((75, 100), (756, 1008))])
def test_depth_gather_matches_interpolate(dst, src):
    from nnr import camera
    h, w = dst
    g = torch.Generator().manual_seed(3)
    img = torch.rand(1, 1, *src, generator=g)
    idx = torch.randperm(h * w, generator=g)[:777]
    up = torch.randn(1, 777, 1, generator=g)
    a = img.clone().requires_grad_(True)
    ref = F.interpolate(a, dst, mode="nearest").view(1, 1, -1).permute(0, 2, 1)[:, idx]
    (ref * up).sum().backward()
    b = img.to(DEV).requires_grad_(True)
    out = camera.depth_gather(b, idx.to(DEV), h, w)
    (out * up.to(DEV)).sum().backward()
    assert torch.equal(out.cpu(), ref.detach())
    close(b.grad, a.grad, 1e-6)


def test_pixels_from_index_is_bit_exact():
    from model.common import arange_pixels
    from nnr import camera
    h, w = 540, 960
    idx = torch.randperm(h * w, generator=torch.Generator().manual_seed(9))[:4096]
    ref = arange_pixels((h, w))[1][:, idx]
    assert torch.equal(camera.pixels_from_index(idx.to(DEV), h, w).cpu(), ref)


@pytest.mark.parametrize("l2,ndc,detach", [(False, False, False), (True, False, False), (False, True, False), (False, False, True)])
def test_render_loss_matches_torch(l2, ndc, detach):
    from model.losses import Loss
    from nnr import camera
    g = torch.Generator().manual_seed(4)
    R = 1500
    rgb, gt = torch.rand(1, R, 3, generator=g), torch.rand(1, R, 3, generator=g)
    dist, dgt = torch.rand(R, generator=g) * 5, 1 + torch.rand(R, generator=g) * 5
    mask = torch.rand(R, generator=g) > 0.1
    crit = Loss({'depth_loss_type': 'l1'})
    a, b, c = rgb.clone().requires_grad_(True), dist.clone().requires_grad_(True), dgt.clone().requires_grad_(True)
    tgt = c[mask]
    if ndc:
        tgt = 1 - 1 / tgt
    if detach:
        tgt = tgt.detach()
    lr = crit.get_rgb_full_loss(a, gt, 'l2' if l2 else 'l1')
    ld = crit.get_depth_loss(b[mask], tgt)
    ref = 1.0 * lr + 0.04 * ld
    ref.backward()
    a2, b2, c2 = (t.to(DEV).requires_grad_(True) for t in (rgb, dist, dgt))
    loss, aux = camera.render_loss(a2, gt.to(DEV), b2, c2, mask.to(DEV), r_total=R, w_rgb=1.0, w_depth=0.04, rgb_l2=l2, ndc=ndc,
                                   detach_gt=detach)
    loss.backward()
    close(loss, ref, 1e-6)
    close(aux[0], lr, 1e-6), close(aux[1], ld, 1e-6), close(aux[2], F.mse_loss(rgb, gt), 1e-6)
    assert int(aux[3]) == int(mask.sum())
    close(a2.grad, a.grad, 1e-7), close(b2.grad, b.grad, 1e-7)
    close(c2.grad, c.grad if c.grad is not None else torch.zeros(R), 1e-7)


@pytest.mark.parametrize("name", ["tanks_d128", "uniform_distalpha_masked_d128", "zero_pose_d128", "tanks_d256_n192"])
def test_trainer_train_step_matches_reference_golden(name, monkeypatch):
    """model.Trainer.train_step on the GPU -- pose exp, inverses, ray setup, depth gather, fused render, fused loss heads,
    all backward kernels, one after the other exactly as train.py drives them -- against the reference's gradients."""
    import model as mdl
    from test_host_logic import make_cfg
    case = gu.load_case(name)
    t = gu.tensors(case)
    rc = gu.render_cfg(case)
    cfg = make_cfg(int(case["cfg.hidden"]), **{k: rc[k] for k in ('num_points', 'dist_alpha', 'sample_option', 'depth_range',
                                                                  'normalise_ray', 'white_background')})
    R = int(case["cfg.R"])
    tcfg = {'type': 'nope_nerf', 'n_training_points': R, 'vis_geo': False, 'detach_gt_depth': False, 'pc_ratio': 4,
            'match_method': 'dense', 'shift_first': False, 'detach_ref_img': True, 'scale_pcs': True, 'detach_rgbs_scale': False,
            'vis_reprojection_every': 5000, 'nearest_limit': 0.01, 'annealing_epochs': 2000, 'rgb_weight': [1.0, 1.0],
            'depth_weight': [0.04, 0.0], 'pc_weight': [0.0, 0.0], 'rgb_s_weight': [0.0, 0.0],
            'depth_consistency_weight': [0.0, 0.0], 'weight_dist_2nd_loss': [0.0, 0.0], 'weight_dist_1st_loss': [0.0, 0.0],
            'depth_loss_type': 'l1', 'with_ssim': False, 'with_auto_mask': False}
    net = mdl.OfficialStaticNerf(cfg)
    net.load_state_dict(case["weights"])
    model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device=DEV), cfg, device=DEV)
    pose = mdl.LearnPose(gu.N_CAMS, True, True, cfg).to(DEV)
    dist = mdl.Learn_Distortion(gu.N_CAMS, True, True, cfg).to(DEV)
    with torch.no_grad():
        pose.r.copy_(t["pose_r"]); pose.t.copy_(t["pose_t"])
        dist.global_scales.copy_(t["scales"]); dist.global_shifts.copy_(t["shifts"])
    sgd = lambda m: torch.optim.SGD(m.parameters(), lr=0.0)
    tr = mdl.Trainer(model, sgd(model), tcfg, device=DEV, optimizer_pose=sgd(pose), pose_param_net=pose,
                     optimizer_distortion=sgd(dist), distortion_net=dist)
    h, w, cam = int(case["cfg.h"]), int(case["cfg.w"]), int(case["cfg.cam"])
    perm = torch.cat([t["ray_idx"], torch.zeros(h * w - R, dtype=torch.long)]).to(DEV)
    monkeypatch.setattr(torch, "randperm", lambda n, device=None: perm)
    if t["jitter"] is not None:
        monkeypatch.setattr(torch, "rand", lambda *a, **k: t["jitter"].to(DEV))
    data = {'img': t["img"], 'img.idx': cam, 'img.dpt': t["depth_img"][:, 0], 'img.camera_mat': t["K"],
            'img.scale_mat': torch.eye(4)[None]}
    ld = tr.train_step(data, it=0, epoch=0, scheduling_start=10000, render_path=None)
    assert abs(float(ld['loss']) - float(case["out.loss"])) <= 1e-4
    got = {"w." + k: v.grad for k, v in net.named_parameters()}
    got.update(pose_r=pose.r.grad, pose_t=pose.t.grad, scales=dist.global_scales.grad, shifts=dist.global_shifts.grad)
    for k, (kind, ref, norm) in gu.golden_grads(case).items():
        gu.compare_grad(k, got[k], kind, ref, norm, 1e-4)


def test_ndc_rays_match_the_torch_expression():
    """nnr_ndc_rays_fwd/bwd vs get_ndc_rays_fxfy (reference model/common.py:632-675) under torch autograd."""
    from model.common import get_ndc_rays_fxfy
    from nnr import camera
    g = torch.Generator().manual_seed(4)
    R = 777
    o = (0.2 * torch.randn(R, 3, generator=g)).to(DEV)
    d = torch.randn(R, 3, generator=g)
    d[:, 2] = -(0.5 + d[:, 2].abs())                      # forward-facing: looking down -z
    d = (d / d.norm(dim=-1, keepdim=True)).to(DEV)
    K = torch.diag(torch.tensor([1.8, -2.4, -1.0, 1.0])).unsqueeze(0).to(DEV)
    go, gd = torch.randn(R, 3, generator=g).to(DEV), torch.randn(R, 3, generator=g).to(DEV)
    o1, d1 = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    ro, rd = get_ndc_rays_fxfy(torch.cat([K[:, 0, 0], K[:, 1, 1]]), 1.0, rays_o=o1, rays_d=d1)
    ((ro * go).sum() + (rd * gd).sum()).backward()
    o2, d2 = o.clone().requires_grad_(True), d.clone().requires_grad_(True)
    ho, hd = camera.ndc_rays(o2, d2, K, 1.0)
    ((ho * go).sum() + (hd * gd).sum()).backward()
    for a, b in ((ho, ro), (hd, rd)):      # same operations in the same order; ATen's device division may round differently
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
    for a, b in ((o2.grad, o1.grad), (d2.grad, d1.grad)):
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))
    # a single origin broadcast over the rays (what the camera centre is): its gradient is the sum over rays
    c1, c2 = o[0].clone().requires_grad_(True), o[0].clone().requires_grad_(True)
    ro, rd = get_ndc_rays_fxfy(torch.cat([K[:, 0, 0], K[:, 1, 1]]), 1.0, rays_o=c1.expand(R, 3), rays_d=d)
    ((ro * go).sum() + (rd * gd).sum()).backward()
    ho, hd = camera.ndc_rays(c2, d, K, 1.0)
    ((ho * go).sum() + (hd * gd).sum()).backward()
    assert float((c2.grad - c1.grad).abs().max()) <= 1e-4 * max(1.0, float(c1.grad.abs().max()))


@pytest.mark.parametrize("shift_first", [False, True])
def test_depth_gather_affine_matches_distort_then_gather(shift_first):
    """nnr_depth_gather_affine_* == distorting the whole map (training.py:240-245) and gathering (network.py:22-24)."""
    from nnr import camera
    g = torch.Generator().manual_seed(8)
    h, w, hd, wd, R = 54, 96, 40, 72, 500
    raw = (1 + 2 * torch.rand(1, 1, hd, wd, generator=g)).to(DEV)
    idx = torch.randperm(h * w, generator=g)[:R].to(DEV)
    up = torch.randn(1, R, 1, generator=g).to(DEV)
    s1, t1 = torch.tensor([1.07], device=DEV, requires_grad=True), torch.tensor([-0.03], device=DEV, requires_grad=True)
    full = (raw + t1) * s1 if shift_first else raw * s1 + t1
    ref = camera.depth_gather(full, idx, h, w)
    (ref * up).sum().backward()
    s2, t2 = s1.detach().clone().requires_grad_(True), t1.detach().clone().requires_grad_(True)
    got = camera.depth_gather_affine(raw, idx, s2, t2, h, w, shift_first)
    (got * up).sum().backward()
    assert float((got - ref).abs().max()) <= 2.5e-7 * float(ref.abs().max())      # one ulp: ATen's device arithmetic may contract
    for a, b in ((s2.grad, s1.grad), (t2.grad, t1.grad)):
        assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))


# ---------------------------------------------------------------------------------------------------------------------------------
# the fused front end of a training step (nnr_step_rays_fwd / _bwd) against the separate launches it replaces
# ---------------------------------------------------------------------------------------------------------------------------------
def _front_end_step(case_name, n_rays, fused, cam=None, scale_value=None, shift_first=False):
    import golden_util as gu
    from test_gpu_dp import _trainer
    case = gu.load_case(case_name)
    tr, mods, data = _trainer(case, n_rays)
    tr.fuse_front_end = fused
    tr.shift_first = shift_first
    if cam is not None:
        data['img.idx'] = cam
    if scale_value is not None:
        with torch.no_grad():
            mods[2].global_scales[data['img.idx']] = scale_value
    torch.manual_seed(77)
    torch.cuda.manual_seed(77)
    ld = tr.train_step(data, it=0, epoch=0, scheduling_start=10000, render_path=None)
    grads = [None if p.grad is None else p.grad.detach().clone() for m in mods for p in m.parameters()]
    return {k: ld[k].detach().clone().reshape(-1) for k in ('loss', 'loss_rgb', 'loss_depth', 'l2_mean', 'scale', 'shift')}, grads


@pytest.mark.parametrize("case_name,n_rays,cam,scale_value,shift_first", [
    ("tanks_d128", 96, None, None, False),
    ("llff_ndc_d128", 64, None, None, False),
    ("uniform_distalpha_masked_d128", 96, None, None, True),      # masked depths (0 in the map), (depth + shift) * scale
    ("tanks_d128", 50, 3, None, False),                            # the last camera: its scale is the gauge (pinned to 1, no gradient)
    ("tanks_d128", 50, 0, 0.004, False),                           # a scale below the floor: the constant 0.01, no gradient
    ("white_nonorm_d128", 40, 2, None, False),                     # rays not normalised
])
def test_fused_front_end_equals_the_separate_launches(case_name, n_rays, cam, scale_value, shift_first):
    """One launch each way instead of se3_exp / inv4 / distortion lookup / pixels / depth gather / colour gather / ray_setup and their
    backward: same arithmetic in the same order -> the same loss values and the same 28 gradient tensors, to the last bit or at most a
    few ulps (the compiler may contract multiply-adds differently in the two kernels)."""
    l_ref, g_ref = _front_end_step(case_name, n_rays, False, cam, scale_value, shift_first)
    l_fus, g_fus = _front_end_step(case_name, n_rays, True, cam, scale_value, shift_first)
    for k in l_ref:
        assert torch.allclose(l_ref[k], l_fus[k], rtol=1e-6, atol=1e-7), (k, l_ref[k], l_fus[k])
    assert len(g_ref) == len(g_fus) == 28
    for a, b in zip(g_ref, g_fus):
        assert (a is None) == (b is None)           # a table outside the step's graph keeps .grad None in both (Adam then skips it)
        if a is None:
            continue
        scale = max(1e-6, float(a.abs().max()))
        assert float((a - b).abs().max()) <= 2e-6 * scale, float((a - b).abs().max()) / scale
    g_scales = g_fus[-1]              # Learn_Distortion declares global_shifts first, global_scales second
    if cam == 3:                      # the gauge: effective scale exactly 1, the scale table is not in the graph
        assert float(l_fus['scale']) == 1.0 and g_scales is None
    elif scale_value is not None:     # below the floor: effective scale 0.01 and a zero scale gradient for that camera
        assert abs(float(l_fus['scale']) - 0.01) < 1e-9 and float(g_scales.abs().max()) == 0.0
    else:
        assert float(g_scales.abs().max()) > 0.0
