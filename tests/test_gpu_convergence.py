"""End-to-end sanity check on the GPU (-m gpu): a tiny synthetic scene (a textured plane seen from six nearby cameras, exact
mono depth), a few hundred Trainer.train_step iterations through every fused op in the loop -- render kernels, per-image
losses, fused Adam -- with poses and distortions learnable from their identity initialisation.  Asserts what any working
NoPe-NeRF step must deliver: the photometric loss falls substantially, nothing becomes non-finite, the depth distortions stay
near identity (the mono depth is exact)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H, W, N_CAMS = 48, 64, 6


def _scene(dev):
    g = torch.Generator().manual_seed(3)
    f = 0.9 * W
    K = torch.diag(torch.tensor([2 * f / W, -2 * f / H, -1.0, 1.0]))
    Kinv = torch.inverse(K)
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    frames = []
    for c in range(N_CAMS):
        t = torch.tensor([0.08 * (c - 2.5), 0.03 * math.sin(c), 0.0])            # cameras side by side, looking down -z
        # ray through each pixel: direction = Kinv [x', y', 1, 1] (z = -1), origin t; plane z = -3
        d = torch.stack([xs * Kinv[0, 0], ys * Kinv[1, 1], torch.full_like(xs, Kinv[2, 2])], -1)
        depth = (-3.0 - t[2]) / d[..., 2]                                           # multiples of the unit-|z| direction
        hit = t + depth[..., None] * d
        u, v = hit[..., 0], hit[..., 1]
        img = torch.stack([0.5 + 0.4 * torch.sin(3.0 * u) * torch.cos(2.0 * v), 0.5 + 0.4 * torch.sin(2.0 * u + 1.0),
                           0.5 + 0.4 * torch.cos(4.0 * v - 0.5)], 0)
        frames.append((img.unsqueeze(0).to(dev), depth.unsqueeze(0).to(dev)))
    return K.unsqueeze(0).to(dev), frames


def test_training_reduces_the_loss_on_a_synthetic_scene():
    import model as mdl
    dev = torch.device("cuda")
    torch.manual_seed(0)
    cfg = {
        'model': {'hidden_dim': 128, 'pos_enc_levels': 10, 'dir_enc_levels': 4, 'occ_activation': 'softplus'},
        'rendering': {'type': 'nope_nerf', 'n_max_network_queries': 64000, 'white_background': False, 'radius': 4.0,
                      'num_points': 64, 'depth_range': [0.5, 6.0], 'dist_alpha': False, 'use_ray_dir': True,
                      'normalise_ray': True, 'normal_loss': False, 'sample_option': 'uniform', 'outside_steps': 0},
        'depth': {'type': 'None'}, 'distortion': {'fix_scaleN': True},
        'training': {
            'type': 'nope_nerf', 'n_training_points': 512, 'vis_geo': False, 'detach_gt_depth': False, 'pc_ratio': 4,
            'match_method': 'dense', 'shift_first': False, 'detach_ref_img': True, 'scale_pcs': True,
            'detach_rgbs_scale': False, 'vis_reprojection_every': 10 ** 9, 'nearest_limit': 0.01, 'annealing_epochs': 2000,
            'rgb_weight': [1.0, 1.0], 'depth_weight': [0.04, 0.0], 'pc_weight': [1.0, 0.0], 'rgb_s_weight': [1.0, 0.0],
            'depth_consistency_weight': [0.0, 0.0], 'weight_dist_2nd_loss': [0.0, 0.0], 'weight_dist_1st_loss': [0.0, 0.0],
            'depth_loss_type': 'l1', 'with_ssim': False, 'with_auto_mask': False},
    }
    net = mdl.OfficialStaticNerf(cfg)
    model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device=dev), cfg, device=dev)
    pose = mdl.LearnPose(N_CAMS, True, True, cfg).to(dev)
    dist = mdl.Learn_Distortion(N_CAMS, True, True, cfg).to(dev)
    tr = mdl.Trainer(model, torch.optim.Adam(model.parameters(), lr=1e-3), cfg['training'], device=dev,
                     optimizer_pose=torch.optim.Adam(pose.parameters(), lr=5e-4), pose_param_net=pose,
                     optimizer_distortion=torch.optim.Adam(dist.parameters(), lr=5e-4), distortion_net=dist)
    K, frames = _scene(dev)
    eye = torch.eye(4, device=dev).unsqueeze(0)
    hist = []
    for it in range(360):
        c = it % N_CAMS
        r = c + 1 if c < N_CAMS - 1 else c - 1
        data = {"img": frames[c][0], "img.idx": c, "img.dpt": frames[c][1], "img.camera_mat": K, "img.scale_mat": eye,
                "img.ref_imgs": frames[r][0], "img.ref_dpts": frames[r][1], "img.ref_idxs": r}
        ld = tr.train_step(data, it=it + 1, epoch=0, scheduling_start=10000, render_path=None)
        hist.append(ld)
    vals = {k: torch.stack([h[k].detach().float().reshape(()) for h in hist]).cpu().numpy() for k in ("loss", "loss_rgb", "loss_depth", "loss_pc", "loss_rgb_s")}
    assert all(np.isfinite(v).all() for v in vals.values())
    first, last = vals["loss_rgb"][:12].mean(), vals["loss_rgb"][-12:].mean()
    print("loss_rgb %.4f -> %.4f, loss_depth %.4f -> %.4f, loss_pc %.4f -> %.4f, loss_rgb_s %.4f -> %.4f" % (
        first, last, vals["loss_depth"][:12].mean(), vals["loss_depth"][-12:].mean(), vals["loss_pc"][:12].mean(),
        vals["loss_pc"][-12:].mean(), vals["loss_rgb_s"][:12].mean(), vals["loss_rgb_s"][-12:].mean()))
    assert last < 0.45 * first, (first, last)                          # the photometric term (L1 sum / R) falls by more than half
    assert vals["loss"][-12:].mean() < vals["loss"][:12].mean()
    assert np.isfinite(pose.r.detach().cpu().numpy()).all() and float(pose.t.detach().abs().max()) < 0.5
    s = dist.global_scales.detach().cpu().numpy()
    assert np.all(np.abs(s - 1.0) < 0.2)                                # exact mono depth: scales stay near 1
