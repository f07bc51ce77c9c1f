#!/usr/bin/env python
"""Joint pose + radiance-field training on a scene in the reference's on-disk layout, through this repository's `dataloading`,
`model` and `utils_poses` packages -- the loop of reference train.py:198-300 without its logging / checkpoint plumbing, used for
the offline convergence runs of SURVEY.md 8(d) (configs 4-5) and to time the loop with the host loader against the HBM-resident
one.  GPU only (the render path has no CPU fallback).

    python tools/scene_writer.py /tmp/scenes --frames 16 --size 120 160
    python tools/train_scene.py /tmp/scenes synthetic --epochs 200 [--style tanks|llff] [--host-loader] [--out result.json]

Reported per run: PSNR of the training rays (from l2_mean, as train.py:283-285 does), ATE / RPE of the learned poses against the
scene's ground truth after sim(3) alignment (train.py:270-281), loop throughput in steps/s and rays/s.  One host sync per epoch
(the reference's loop calls .item() three times per step, train.py:212-214)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))

import numpy as np
import torch


def scene_cfg(path, scene, style="tanks", n_rays=1024, n_samples=128, hidden=256, resident=True, aux=True, resize_factor=None,
              sample_rate=8, workers=0, mfma_dtype="fp32"):
    """The keys configs/default.yaml + configs/Tanks/*.yaml (style 'tanks') or configs/LLFF/fern.yaml (style 'llff') would set."""
    llff = style == "llff"
    on = [1.0, 0.0] if aux else [0.0, 0.0]
    return {
        "model": {"hidden_dim": hidden, "pos_enc_levels": 10, "dir_enc_levels": 4, "occ_activation": "softplus",
                  "network_type": "official"},
        "dataloading": {"dataset_name": "any", "path": path, "scene": [scene], "batchsize": 1, "n_workers": workers, "with_depth": False,
                        "spherify": True, "customized_poses": False, "customized_focal": False, "resize_factor": resize_factor,
                        "depth_net": "dpt", "crop_size": 0, "random_ref": 1, "norm_depth": False, "load_colmap_poses": True,
                        "shuffle": True, "sample_rate": sample_rate, "resident": resident},
        "rendering": {"type": "nope_nerf", "n_max_network_queries": 64000, "white_background": False, "radius": 4.0,
                      "num_points": n_samples, "depth_range": [0.0, 1.0] if llff else [0.01, 10], "dist_alpha": llff,
                      "use_ray_dir": True, "normalise_ray": True, "normal_loss": False,
                      "sample_option": "ndc" if llff else "uniform", "outside_steps": 0, "mfma_dtype": mfma_dtype},
        "depth": {"type": "None"},
        "pose": {"learn_pose": True, "learn_R": True, "learn_t": True, "init_pose": False, "learn_focal": False},
        "distortion": {"learn_distortion": True, "fix_scaleN": True, "learn_scale": True, "learn_shift": True},
        "training": {
            "type": "nope_nerf", "n_training_points": n_rays, "vis_geo": False, "detach_gt_depth": False, "pc_ratio": 4,
            "match_method": "dense", "shift_first": False, "detach_ref_img": True, "scale_pcs": True, "detach_rgbs_scale": False,
            "vis_reprojection_every": 10 ** 9, "nearest_limit": 0.01, "annealing_epochs": 2000, "rgb_weight": [1.0, 1.0],
            "depth_weight": [0.04, 0.0], "pc_weight": on, "rgb_s_weight": on, "depth_consistency_weight": [0.0, 0.0],
            "weight_dist_2nd_loss": [0.0, 0.0], "weight_dist_1st_loss": [0.0, 0.0], "depth_loss_type": "l1", "with_ssim": False,
            "with_auto_mask": False, "learning_rate": 1e-3, "pose_lr": 5e-4, "distortion_lr": 5e-4,
        },
    }


def build(cfg, device, n_views):
    import model as mdl
    net = mdl.OfficialStaticNerf(cfg)
    nope = mdl.get_model(mdl.Renderer(net, cfg["rendering"], device=device), cfg, device=device)
    pose = mdl.LearnPose(n_views, True, True, cfg, init_c2w=None).to(device)
    dist = mdl.Learn_Distortion(n_views, True, True, cfg).to(device)
    t = cfg["training"]
    opt = torch.optim.Adam(nope.parameters(), lr=t["learning_rate"])
    opt_pose = torch.optim.Adam(pose.parameters(), lr=t["pose_lr"])
    opt_dist = torch.optim.Adam(dist.parameters(), lr=t["distortion_lr"])
    trainer = mdl.Trainer(nope, opt, t, device=device, optimizer_pose=opt_pose, pose_param_net=pose, optimizer_distortion=opt_dist,
                          distortion_net=dist, cfg_all=cfg)
    return trainer, pose, dist


def pose_errors(pose_net, gt_poses, n_views):
    from utils_poses.align_traj import align_ate_c2b_use_a2b
    from utils_poses.comp_ate import compute_ATE, compute_rpe
    with torch.no_grad():
        learned = torch.stack([pose_net(i) for i in range(n_views)])
    aligned = align_ate_c2b_use_a2b(learned, gt_poses).cpu().numpy()
    gt = gt_poses.cpu().numpy()
    rpe_t, rpe_r = compute_rpe(gt, aligned)
    return {"ate": float(compute_ATE(gt, aligned)), "rpe_trans_x100": float(rpe_t * 100), "rpe_rot_deg": float(np.degrees(rpe_r))}


def novel_view_eval(cfg, nope, pose, train_field, device, epochs, out_dir, n_points=1024):
    """What reference evaluation/eval.py:54-186 does for the held-out views: initialise their poses from the learned training
    trajectory (init_method 'pre': the training pose just before each held-out frame), optimise them against the frozen field
    (Trainer_pose, Adam 1e-3 halved five times), then render every held-out frame in full and score it.  LPIPS needs a VGG
    checkpoint and is reported as 0."""
    import dataloading as dl
    import model as mdl
    from model.eval_images import Eval_Images
    loader, fields = dl.get_dataloader(cfg, mode="eval", shuffle=False)
    f = fields["img"]
    if f.N_imgs == 0:
        return None
    with torch.no_grad():
        learned = torch.stack([pose(i) for i in range(train_field.N_imgs)])
    sr = train_field.sample_rate
    init = learned[int(sr / 2) - 1::sr - 1][:f.N_imgs].clone()
    eval_pose = mdl.LearnPose(f.N_imgs, True, True, cfg, init_c2w=init).to(device)
    opt = torch.optim.Adam(eval_pose.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=list(range(0, epochs, max(1, epochs // 5))), gamma=0.5)
    tp = mdl.Trainer_pose(nope, {"n_points": n_points, "type": "nope_nerf"}, device=device, optimizer_pose=opt, pose_param_net=eval_pose)
    first = last = None
    for e in range(epochs):
        l2 = torch.stack([tp.train_step(batch)["loss"].detach() for batch in loader]).mean()
        sched.step()
        if e in (0, epochs - 1):
            v = float(l2)
            first, last = (v, last) if e == 0 else (first, v)
    eval_pose.eval()
    with torch.no_grad():
        c2ws = torch.stack([eval_pose(i) for i in range(f.N_imgs)])
    cfg = dict(cfg, extract_images={"resolution": [f.H, f.W]})
    gen = Eval_Images(nope.renderer, cfg, use_learnt_poses=True, use_learnt_focal=False, device=device, render_type="nope_nerf",
                      c2ws=c2ws, img_list=f.img_list)
    zero = lambda a, b, normalize=True: torch.zeros(())
    t0 = time.perf_counter()
    rows = [gen.eval_images(batch, out_dir, None, zero, logger=None) for batch in loader]
    dt = time.perf_counter() - t0
    return {"views": f.N_imgs, "pose_opt_epochs": epochs, "pose_opt_mse_first": first, "pose_opt_mse_last": last,
            "psnr": float(np.mean([r["psnr"] for r in rows])), "ssim": float(np.mean([r["ssim"] for r in rows])),
            "seconds_per_frame": dt / max(1, len(rows))}


def run(path, scene, style="tanks", epochs=100, seed=42, log_every=10, device=None, eval_epochs=0, eval_dir=None, **cfg_kw):
    """-> dict with the per-epoch PSNR / pose-error curve and the loop throughput (+ novel-view scores with eval_epochs > 0)."""
    import dataloading as dl
    from model.common import mse2psnr
    from nnr import parallel
    device = torch.device(device or "cuda")
    sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)
    np.random.seed(seed)
    torch.manual_seed(seed)
    cfg = scene_cfg(path, scene, style=style, **cfg_kw)
    loader, fields = dl.get_dataloader(cfg, mode="train", shuffle=True)
    field = fields["img"]
    n_views = field.N_imgs
    trainer, pose, dist = build(cfg, device, n_views)
    gt = field.c2ws.to(device)
    curve = [dict(epoch=-1, psnr=None, **pose_errors(pose, gt, n_views))]
    it, t_loop, steps_timed = -1, 0.0, 0
    for epoch in range(epochs):
        l2 = []
        sync()
        t0 = time.perf_counter()
        for batch in loader:
            it += 1
            losses = trainer.train_step(batch, it, epoch, 10 ** 6, None)
            l2.append(losses["l2_mean"])
        mse = float(torch.stack(l2).detach().mean())             # the epoch's only device->host sync
        if epoch > 0:                                            # epoch 0 pays the lazy initialisations
            t_loop += time.perf_counter() - t0
            steps_timed += len(l2)
        if epoch % log_every == 0 or epoch == epochs - 1:
            curve.append(dict(epoch=epoch, psnr=float(mse2psnr(mse)), **pose_errors(pose, gt, n_views)))
            if parallel.rank() == 0:
                print(json.dumps(curve[-1]), flush=True)
    trainer.flush_nan_check()        # the step's NaN check is one step late by design: look at the last one too
    n_rays = cfg["training"]["n_training_points"]
    novel = None
    if eval_epochs > 0:
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            novel = novel_view_eval(cfg, trainer.model, pose, field, device, eval_epochs, eval_dir or tmp, n_points=min(1024, n_rays))
    return {"novel_views": novel, "scene": scene, "style": style, "views": n_views, "image": [field.H, field.W], "rays_per_step": n_rays,
            "samples_per_ray": cfg["rendering"]["num_points"], "hidden": cfg["model"]["hidden_dim"], "epochs": epochs,
            "steps": it + 1, "loader": "resident" if cfg["dataloading"]["resident"] else "host(workers=%d)" % cfg["dataloading"]["n_workers"],
            "aux_losses": cfg["training"]["pc_weight"][0] != 0.0,
            "steps_per_s": steps_timed / t_loop if t_loop else None, "rays_per_s": steps_timed * n_rays / t_loop if t_loop else None,
            "curve": curve}


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("path")
    ap.add_argument("scene")
    ap.add_argument("--style", choices=("tanks", "llff"), default="tanks")
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--factor", type=int, default=None)
    ap.add_argument("--host-loader", action="store_true")
    ap.add_argument("--workers", type=int, default=0, help="DataLoader worker processes of the host loader (reference default: 1)")
    ap.add_argument("--no-aux", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="rendering.mfma_dtype: bf16 (bf16 MFMA products, fp32 accumulation)")
    ap.add_argument("--log-every", type=int, default=10)
    ap.add_argument("--eval-epochs", type=int, default=0, help="test-time pose optimisation epochs before scoring the held-out views")
    ap.add_argument("--eval-dir", default=None, help="where the rendered held-out frames go (default: a temporary directory)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    res = run(a.path, a.scene, style=a.style, epochs=a.epochs, log_every=a.log_every, eval_epochs=a.eval_epochs, eval_dir=a.eval_dir, n_rays=a.rays, n_samples=a.samples,
              hidden=a.hidden, resident=not a.host_loader, aux=not a.no_aux, resize_factor=a.factor, workers=a.workers,
              mfma_dtype="bf16" if a.bf16 else "fp32")
    print(json.dumps({k: v for k, v in res.items() if k != "curve"}))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
