"""Golden vectors for the loss-side switches of the training step that the other goldens leave at their defaults (reference
model/training.py:187-217 weight annealing and the L1 -> L2 switch, model/losses.py:34-64 depth_loss_type 'invariant', :92-112
trajectory smoothness, training.detach_gt_depth): the REFERENCE Trainer.train_step, one step per case on the same inputs and
draws, loss dictionary + pose / distortion gradients + two network gradient tensors frozen in tests/golden/loss_switches.npz.
Authoring container only:  python oracle/gen_golden_switches.py"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import gen_golden_aux as ga  # noqa: E402

# name -> (training overrides, epoch, scheduling_start)
CASES = {
    "mid_anneal": (dict(annealing_epochs=4), 1, 0),          # weights a quarter of the way from start to end, still L1
    "l2_phase": (dict(annealing_epochs=2), 3, 0),            # past the switch: L2 colour loss, depth / per-image weights at their end value 0
    "invariant": (dict(depth_loss_type="invariant"), 0, 10000),
    "smooth_traj": (dict(weight_dist_1st_loss=[0.1, 0.1], weight_dist_2nd_loss=[0.5, 0.5]), 0, 10000),
    "detach_gt": (dict(detach_gt_depth=True), 0, 10000),
}
LOGGED = ("loss", "loss_rgb", "loss_depth", "loss_pc", "loss_rgb_s", "l2_mean", "loss_dist_1st", "loss_dist_2nd")
NET = ("fc_rgb.weight", "layers0.0.weight", "fc_density.bias")


def main():
    ref = gg.import_reference()
    torch.set_num_threads(8)
    inp = ga.inputs(51)
    blob = {"in." + k: v.numpy() for k, v in inp.items()}
    dev = torch.device("cpu")
    cam, nb = 2, 3
    blob["cam"], blob["nb"] = cam, nb
    for name, (over, epoch, start) in CASES.items():
        cfg = copy.deepcopy(gg.base_cfg(128))
        cfg["training"].update(n_training_points=ga.R, pc_weight=[1.0, 0.0], rgb_s_weight=[1.0, 0.0], vis_reprojection_every=10 ** 9)
        cfg["training"].update(over)
        cfg["rendering"]["num_points"] = ga.N
        torch.manual_seed(42)
        net = ref.OfficialStaticNerf(cfg)
        model = ref.get_model(ref.Renderer(net, cfg["rendering"], device=dev), cfg, device=dev)
        pose, dist = ref.LearnPose(ga.N_CAMS, True, True, cfg), ref.Learn_Distortion(ga.N_CAMS, True, True, cfg)
        with torch.no_grad():
            pose.r.copy_(inp["pose_r"]); pose.t.copy_(inp["pose_t"])
            dist.global_scales.copy_(inp["scales"]); dist.global_shifts.copy_(inp["shifts"])
        sgd = lambda m: torch.optim.SGD(m.parameters(), lr=0.0)
        tr = ref.Trainer(model, sgd(model), cfg["training"], device=dev, optimizer_pose=sgd(pose), pose_param_net=pose,
                         optimizer_distortion=sgd(dist), distortion_net=dist)
        data = {"img": inp["img"], "img.idx": cam, "img.dpt": inp["dpt"], "img.camera_mat": inp["K"],
                "img.scale_mat": torch.eye(4).unsqueeze(0), "img.ref_imgs": inp["ref_img"], "img.ref_dpts": inp["ref_dpt"],
                "img.ref_idxs": nb}
        torch.manual_seed(7)
        ld = tr.train_step(data, it=1, epoch=epoch, scheduling_start=start, render_path=None)
        for k in LOGGED:
            blob[f"{name}.out.{k}"] = np.float64(float(ld[k]))
        for k, t in (("pose_r", pose.r), ("pose_t", pose.t), ("scales", dist.global_scales), ("shifts", dist.global_shifts)):
            blob[f"{name}.g.{k}"] = (t.grad if t.grad is not None else torch.zeros_like(t)).numpy()
        sd = dict(net.named_parameters())
        for k in NET:
            blob[f"{name}.g.net.{k}"] = sd[k].grad.numpy()
        print(f"{name:12s} " + "  ".join(f"{k} {float(ld[k]):.6f}" for k in LOGGED))
    torch.manual_seed(7)
    blob["ray_idx"] = torch.randperm(ga.H * ga.W)[:ga.R].numpy()
    blob["jitter"] = torch.rand(1, ga.R, ga.N).numpy()
    out = os.path.join(gg.OUT, "loss_switches.npz")
    np.savez_compressed(out, **blob)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
