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
        assert float(a.grad.sum()) == 3000.0 and float(b.grad.abs().sum()) == 0.0 and float(ld['loss']) == 2.5
    finally:
        dist.destroy_process_group()
