"""Golden vectors for the two remaining callers of the render path that the single-step goldens did not cover (SURVEY 8(b) "what
calls it"): (a) the test-time pose optimisation step, reference Trainer_pose.train_step (model/eval_pose_one_epoch.py:28-98: frozen
network in eval mode, no jitter, MSE on rgb, poses initialised from given c2w matrices); (b) Trainer.train_step with a LEARNABLE
focal length (model/intrinsics.py:5-70, model/training.py:247-252: K rebuilt from LearnFocal every step, gradients to fx, fy).
Runs the REFERENCE classes on CPU and freezes inputs, draws, losses and gradients in tests/golden/eval_focal.npz.
Authoring container only:  python oracle/gen_golden_eval.py"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import gen_golden_aux as ga  # noqa: E402
from gen_golden_poses import trajectory  # noqa: E402
from gen_golden_steps import frames  # noqa: E402


def recorder():
    real = torch.randperm
    seen = {}

    def randperm(n, *a, **k):
        seen["perm"] = real(n, *a, **k)
        return seen["perm"]
    return real, randperm, seen


def main():
    ref = gg.import_reference()
    import model.eval_pose_one_epoch  # noqa: F401  (reference module, now importable)
    torch.set_num_threads(8)
    cfg = copy.deepcopy(gg.base_cfg(128))
    cfg["training"].update(n_training_points=ga.R, pc_weight=[1.0, 0.0], rgb_s_weight=[1.0, 0.0], vis_reprojection_every=10 ** 9)
    cfg["rendering"]["num_points"] = ga.N
    inp = ga.inputs(41)
    imgs, dpts = frames(42)
    dev = torch.device("cpu")
    blob = {"imgs": imgs.numpy(), "dpts": dpts.numpy(), "K": inp["K"].numpy()}
    for k in ("pose_r", "pose_t", "scales", "shifts"):
        blob["init." + k] = inp[k].numpy()

    def network():
        torch.manual_seed(42)
        net = ref.OfficialStaticNerf(cfg)
        return net, ref.get_model(ref.Renderer(net, cfg["rendering"], device=dev), cfg, device=dev)

    # ---- (a) test-time pose optimisation: 3 held-out views initialised from c2w matrices, small learned offsets on top
    net, model = network()
    c2w0, _ = trajectory(3, 7, 0.0)
    c2w0[:, :3, 3] *= 0.1
    pose = ref.LearnPose(3, True, True, cfg, init_c2w=c2w0)
    with torch.no_grad():
        pose.r.copy_(inp["pose_r"][:3]); pose.t.copy_(inp["pose_t"][:3])
    opt = torch.optim.SGD(pose.parameters(), lr=0.0)
    tp = ref.Trainer_pose(model, {"n_points": ga.R, "type": "nope_nerf"}, device=dev, optimizer_pose=opt, pose_param_net=pose)
    real, rec, seen = recorder()
    torch.randperm = rec
    try:
        torch.manual_seed(5)
        view = 1
        data = {"img": imgs[2:3], "img.idx": torch.tensor([view]), "img.camera_mat": inp["K"], "img.scale_mat": torch.eye(4).unsqueeze(0)}
        ld = tp.train_step(data)
    finally:
        torch.randperm = real
    blob.update({"pose_opt.c2w0": c2w0.numpy(), "pose_opt.view": view, "pose_opt.frame": 2, "pose_opt.ray_idx": seen["perm"][:ga.R].numpy(),
                 "pose_opt.loss": np.float64(float(ld["loss"])), "pose_opt.g.r": pose.r.grad.numpy(), "pose_opt.g.t": pose.t.grad.numpy()})
    print("pose optimisation step: loss %.6f  |g_r| %.3e  |g_t| %.3e" % (float(ld["loss"]), pose.r.grad.abs().max(), pose.t.grad.abs().max()))

    # ---- (b) one training step with a learnable focal length (order 2: the parameter is sqrt(f)), first-phase losses on
    net, model = network()
    pose = ref.LearnPose(ga.N_CAMS, True, True, cfg)
    dist = ref.Learn_Distortion(ga.N_CAMS, True, True, cfg)
    with torch.no_grad():
        pose.r.copy_(inp["pose_r"]); pose.t.copy_(inp["pose_t"])
        dist.global_scales.copy_(inp["scales"]); dist.global_shifts.copy_(inp["shifts"])
    K = inp["K"]
    init_focal = [float(K[0, 0, 0]) * 1.1, float(-K[0, 1, 1]) * 0.9]          # start off the true focal
    focal = ref.LearnFocal(True, False, order=2, init_focal=init_focal)
    sgd = lambda m: torch.optim.SGD(m.parameters(), lr=0.0)
    tr = ref.Trainer(model, sgd(model), cfg["training"], device=dev, optimizer_pose=sgd(pose), pose_param_net=pose,
                     optimizer_focal=sgd(focal), focal_net=focal, optimizer_distortion=sgd(dist), distortion_net=dist)
    real_rand = torch.rand
    drawn = {}

    def rand(*s, **k):
        out = real_rand(*s, **k)
        if tuple(s) == (1, ga.R, ga.N):
            drawn["jitter"] = out
        return out
    real, rec, seen = recorder()
    torch.randperm, torch.rand = rec, rand
    try:
        torch.manual_seed(6)
        cam, nb = 1, 2
        data = {"img": imgs[cam:cam + 1], "img.idx": cam, "img.dpt": dpts[cam:cam + 1], "img.camera_mat": K,
                "img.scale_mat": torch.eye(4).unsqueeze(0), "img.ref_imgs": imgs[nb:nb + 1], "img.ref_dpts": dpts[nb:nb + 1],
                "img.ref_idxs": nb}
        ld = tr.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
    finally:
        torch.randperm, torch.rand = real, real_rand
    z = lambda g, like: (g if g is not None else torch.zeros_like(like)).numpy()
    blob.update({"focal.init": np.array(init_focal), "focal.cam": cam, "focal.nb": nb, "focal.ray_idx": seen["perm"][:ga.R].numpy(),
                 "focal.jitter": drawn["jitter"].numpy(), "focal.g.fx": focal.fx.grad.numpy(), "focal.g.fy": focal.fy.grad.numpy(),
                 "focal.g.pose_r": z(pose.r.grad, pose.r), "focal.g.pose_t": z(pose.t.grad, pose.t),
                 "focal.g.scales": z(dist.global_scales.grad, dist.global_scales), "focal.g.shifts": z(dist.global_shifts.grad, dist.global_shifts)})
    for k in ("loss", "loss_rgb", "loss_depth", "loss_pc", "loss_rgb_s", "l2_mean", "focalx", "focaly"):
        blob["focal.out." + k] = np.float64(float(ld[k]))
    print("learnable-focal step: " + "  ".join(f"{k} {float(ld[k]):.6f}" for k in ("loss", "loss_pc", "loss_rgb_s", "focalx", "focaly")) +
          "  g_fx %.4e g_fy %.4e" % (float(focal.fx.grad), float(focal.fy.grad)))
    out = os.path.join(gg.OUT, "eval_focal.npz")
    np.savez_compressed(out, **blob)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
