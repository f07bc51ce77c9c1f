"""GPU (-m gpu): the data-parallel code path of model.Trainer on the HIP kernels, with two *virtual* ranks executed one
after the other on the single GPU (rank()/world_size() patched, the all-reduce replaced by a recorder): the sum of the two
ranks' gradients and losses must equal the single-process step.  Plus an RCCL world-size-1 all-reduce through
nnr.parallel to exercise the real backend call."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _trainer(case, n_rays):
    import model as mdl
    from test_host_logic import make_cfg
    t = gu.tensors(case)
    rc = gu.render_cfg(case)
    cfg = make_cfg(int(case["cfg.hidden"]), **{k: rc[k] for k in ('num_points', 'dist_alpha', 'sample_option', 'depth_range',
                                                                  'normalise_ray', 'white_background', 'use_ray_dir')})
    cfg['model']['occ_activation'] = rc['occ_activation']
    tcfg = {'type': 'nope_nerf', 'n_training_points': n_rays, 'vis_geo': False, 'detach_gt_depth': False, 'pc_ratio': 4,
            'match_method': 'dense', 'shift_first': False, 'detach_ref_img': True, 'scale_pcs': True, 'detach_rgbs_scale': False,
            'vis_reprojection_every': 5000, 'nearest_limit': 0.01, 'annealing_epochs': 2000, 'rgb_weight': [1.0, 1.0],
            'depth_weight': [0.04, 0.0], 'pc_weight': [0.0, 0.0], 'rgb_s_weight': [0.0, 0.0],
            'depth_consistency_weight': [0.0, 0.0], 'weight_dist_2nd_loss': [0.5, 0.5], 'weight_dist_1st_loss': [0.1, 0.1],
            'depth_loss_type': 'l1', 'with_ssim': False, 'with_auto_mask': False}
    net = mdl.OfficialStaticNerf(cfg)
    net.load_state_dict(case["weights"])
    model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device=DEV), cfg, device=DEV)
    pose = mdl.LearnPose(gu.N_CAMS, True, True, cfg).to(DEV)
    distn = mdl.Learn_Distortion(gu.N_CAMS, True, True, cfg).to(DEV)
    with torch.no_grad():
        pose.r.copy_(t["pose_r"]); pose.t.copy_(t["pose_t"])
        distn.global_scales.copy_(t["scales"]); distn.global_shifts.copy_(t["shifts"])
    sgd = lambda m: torch.optim.SGD(m.parameters(), lr=0.0)
    tr = mdl.Trainer(model, sgd(model), tcfg, device=DEV, optimizer_pose=sgd(pose), pose_param_net=pose,
                     optimizer_distortion=sgd(distn), distortion_net=distn)
    data = {'img': t["img"], 'img.idx': int(case["cfg.cam"]), 'img.dpt': t["depth_img"][:, 0], 'img.camera_mat': t["K"],
            'img.scale_mat': torch.eye(4)[None]}
    return tr, [net, pose, distn], data


def _step(case, n_rays, monkeypatch, rank=0, world=1):
    from nnr import parallel
    monkeypatch.setattr(parallel, "rank", lambda: rank)
    monkeypatch.setattr(parallel, "world_size", lambda: world)
    monkeypatch.setattr(parallel.dist, "all_reduce", lambda t, op=None: t)        # keep this rank's share
    tr, mods, data = _trainer(case, n_rays)
    torch.manual_seed(321)
    torch.cuda.manual_seed(321)
    ld = tr.train_step(data, it=0, epoch=0, scheduling_start=10000, render_path=None)
    grads = [p.grad.detach().clone() for m in mods for p in m.parameters()]
    return {k: float(ld[k]) for k in ('loss', 'loss_rgb', 'loss_depth', 'l2_mean', 'loss_dist_1st')}, grads


@pytest.mark.parametrize("name,n_rays,world", [("tanks_d128", 61, 2), ("uniform_distalpha_masked_d128", 96, 2),
                                               ("tanks_d128", 61, 8), ("llff_ndc_d128", 50, 4)])
def test_virtual_ranks_sum_to_single_process(name, n_rays, world, monkeypatch):
    """W virtual ranks (uneven shards: 61 rays over 8 ranks = 8,8,8,8,8,7,7,7), one after the other on the one GPU."""
    case = gu.load_case(name)
    ref_l, ref_g = _step(case, n_rays, monkeypatch)
    parts = [_step(case, n_rays, monkeypatch, r, world) for r in range(world)]
    for k, v in ref_l.items():
        total = sum(p[0][k] for p in parts)
        assert abs(total - v) <= 1e-5 * max(1.0, abs(v)), (k, total, v)
    for i, r in enumerate(ref_g):
        scale = max(1.0, float(r.abs().max()))
        assert float((sum(p[1][i] for p in parts) - r).abs().max()) / scale <= 2e-5


def test_rccl_flat_allreduce_world1():
    from nnr import parallel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        a = torch.nn.Parameter(torch.ones(1000, device=DEV))
        b = torch.nn.Parameter(torch.ones(7, device=DEV))
        a.grad = torch.full_like(a, 3.0)
        ld = {'loss': torch.tensor(2.5, device=DEV)}
        parallel.allreduce_gradients([a, b], ld)
        torch.cuda.synchronize()
        assert float(a.grad.sum()) == 3000.0 and b.grad is None and float(ld['loss']) == 2.5      # no rank has a gradient for b: it stays without one (Adam skips it, as in a single process)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("mid", 2), ("last", 3), ("mid", 8), ("mid_ssim", 3)])
def test_virtual_ranks_shard_the_per_image_losses(name, world, monkeypatch):
    """First training phase (pc_weight = rgb_s_weight = 1): each virtual rank runs the fused per-image kernels on its shard of the
    source points (nnr_aux_cfg.shard_lo/hi); the SUM of the ranks' losses and gradients is the reference's single-process step
    (tests/golden/aux_terms.npz, minted from the reference Trainer.train_step)."""
    import numpy as np
    import test_aux_terms as ta
    from nnr import parallel
    G = ta.GOLD
    gold, ssim = name, name.endswith("_ssim")       # with_ssim: a window centre belongs to the rank that owns the point, its taps need not
    name = name.replace("_ssim", "")
    inp = ta._inp(name)
    cam, ref = int(G[f"{name}.cam"]), int(G[f"{name}.ref"])
    ray_idx, jitter = torch.from_numpy(G[f"{name}.ray_idx"]), torch.from_numpy(G[f"{name}.jitter"])
    monkeypatch.setattr(torch, "randperm", lambda n, device=None, **kw: torch.cat([ray_idx, torch.zeros(n - ta.R, dtype=torch.int64)]).to(device))
    real_rand = torch.rand
    monkeypatch.setattr(torch, "rand", lambda *s, device=None, **kw: jitter.to(device) if tuple(s) == (1, ta.R, ta.N) else real_rand(*s, device=device, **kw))
    from nnr import sampling      # (a shard draws its jitter rows through nnr.sampling.rand_rows, not torch.rand: hand it the golden rows too)
    monkeypatch.setattr(sampling, "rand_rows", lambda total, first, n, device: jitter.reshape(-1)[first:first + n].to(device))
    monkeypatch.setattr(parallel.dist, "all_reduce", lambda t, op=None: t)
    monkeypatch.setattr(parallel, "world_size", lambda: world)
    dev = torch.device(DEV)
    data = {"img": inp["img"].to(dev), "img.idx": cam, "img.dpt": inp["dpt"].to(dev), "img.camera_mat": inp["K"].to(dev),
            "img.scale_mat": torch.eye(4).unsqueeze(0).to(dev), "img.ref_imgs": inp["ref_img"].to(dev),
            "img.ref_dpts": inp["ref_dpt"].to(dev), "img.ref_idxs": ref}
    tot_l, tot_g = {}, {}
    for r in range(world):
        monkeypatch.setattr(parallel, "rank", lambda r=r: r)
        tr, pose, distn = ta._trainer(inp, dev, with_ssim=ssim)
        ld = tr.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
        for k in ("loss", "loss_pc", "loss_rgb_s", "loss_rgb", "loss_depth"):
            tot_l[k] = tot_l.get(k, 0.0) + float(ld[k])
        for k, p in (("pose_r", pose.r), ("pose_t", pose.t), ("scales", distn.global_scales), ("shifts", distn.global_shifts)):
            g = p.grad.detach().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
            tot_g[k] = tot_g.get(k, 0.0) + g
    for k, v in tot_l.items():
        np.testing.assert_allclose(v, float(G[f"{gold}.out.{k}"]), rtol=0, atol=2e-5, err_msg=k)
    for k, g in tot_g.items():
        ref_g = G[f"{gold}.g.{k}"]
        assert float(np.abs(g - ref_g).max()) / max(1.0, float(np.abs(ref_g).max())) <= 1e-4, k
