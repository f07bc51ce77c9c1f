"""Golden vectors for SURVEY section 8 row a16: `rendering.normal_loss: True` -- the normal-consistency vector of reference
model/rendering.py:133-143 (OfficialStaticNerf.gradient, model/official_nerf.py:46-58: second-order autograd through the trunk).

TEST INFRASTRUCTURE; authoring container only (needs /root/reference):   python oracle/gen_golden_normal.py

Runs the REFERENCE renderer with normal_loss on (masked depths, so fewer surface points than rays), asserts the oracle restatement
(nerf_oracle.normal_consistency) reproduces it, and freezes inputs, the two random draws (jitter, surface perturbation),
out['normal'] and the gradients of sum(out['normal']) + the render loss with respect to every parameter in
tests/golden/normal_loss_d128.npz.  (No loss term of the reference consumes out['normal']; the gradient is pinned so that the
double-backward path is.)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg          # noqa: E402
import nerf_oracle as orc        # noqa: E402

CASE = dict(hidden=128, R=40, N=24, h=60, w=80, hd=30, wd=40, rend=dict(normal_loss=True), jitter=True, bad_depth=True)


def main():
    ref = gg.import_reference()
    torch.set_num_threads(8)
    c = CASE
    cfg = gg.base_cfg(c["hidden"])
    cfg["rendering"].update(c["rend"])
    cfg["rendering"]["num_points"] = c["N"]
    inp = gg.make_inputs(c)
    h, w, cam = c["h"], c["w"], inp["cam"]
    torch.manual_seed(42)
    net = ref.OfficialStaticNerf(cfg)
    model = ref.get_model(ref.Renderer(net, cfg["rendering"], device=torch.device("cpu")), cfg, device=torch.device("cpu"))
    pose = ref.LearnPose(gg.N_CAMS, True, True, cfg)
    dist = ref.Learn_Distortion(gg.N_CAMS, True, True, cfg)
    with torch.no_grad():
        pose.r.copy_(inp["pose_r"]); pose.t.copy_(inp["pose_t"])
        dist.global_scales.copy_(inp["scales"]); dist.global_shifts.copy_(inp["shifts"])
    from model.common import arange_pixels
    from model.losses import Loss
    c2w = pose(cam)
    world_mat = torch.inverse(c2w).unsqueeze(0)
    sc, sh = dist(cam)
    depth_in = inp["depth_img"] * sc + sh
    ray_idx = inp["ray_idx"]
    rgb_gt = inp["img"].view(1, 3, h * w).permute(0, 2, 1)[:, ray_idx]
    p = arange_pixels((h, w), 1)[1][:, ray_idx]
    torch.manual_seed(43)
    assert torch.equal(torch.rand(1, c["R"], c["N"]), inp["jitter"])
    torch.manual_seed(43)
    out = model(p, ray_idx, inp["K"], world_mat, torch.eye(4).unsqueeze(0), "nope_nerf", it=0, eval_mode=False,
                depth_img=depth_in, add_noise=True, img_size=(h, w))
    m = int(out["depth_pred"].shape[0])
    assert 0 < m < c["R"] and out["normal"].shape == (m,)
    torch.manual_seed(43)                      # replay: the jitter draw, then rand_like(surface_points)
    torch.rand(1, c["R"], c["N"])
    noise = torch.rand(m, 3)
    crit = Loss(cfg["training"])
    loss = crit.get_rgb_full_loss(out["rgb"], rgb_gt, "l1") + 0.04 * crit.get_depth_loss(out["depth_pred"], out["depth_gt"]) \
        + out["normal"].sum()
    loss.backward()
    grads = {"w." + n: (p_.grad.clone() if p_.grad is not None else torch.zeros_like(p_)) for n, p_ in net.named_parameters()}
    grads.update(pose_r=pose.r.grad.clone(), pose_t=pose.t.grad.clone(), scales=dist.global_scales.grad.clone(),
                 shifts=dist.global_shifts.grad.clone())
    weights = {k: v.detach().clone() for k, v in net.state_dict().items()}

    # the oracle restatement on the same inputs and draws
    params = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
    rcfg = dict(cfg["rendering"])
    rcfg["occ_activation"] = cfg["model"]["occ_activation"]
    oloss, oout = orc.train_step_scope(params, leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], cam, inp["K"],
                                       inp["depth_img"], inp["img"], (h, w), ray_idx, inp["jitter"], rcfg, normal_noise=noise)
    (oloss + oout["normal"].sum()).backward()
    worst = gg.check("normal", oout["normal"], out["normal"], 1e-6)
    worst = max(worst, gg.check("rgb", oout["rgb"], out["rgb"], 1e-6))
    for k, v in grads.items():
        og = params[k[2:]].grad if k.startswith("w.") else leaves[k].grad
        worst = max(worst, gg.check("grad." + k, og if og is not None else torch.zeros_like(v), v, 2e-5))
    old = np.load(os.path.join(gg.OUT, "weights_d128.npz"))
    assert all(np.array_equal(old[k], v.numpy()) for k, v in weights.items())     # the shared seed-42 reference network
    blob = {"cfg.hidden": c["hidden"], "cfg.R": c["R"], "cfg.N": c["N"], "cfg.h": h, "cfg.w": w, "cfg.cam": cam, "cfg.M": m,
            "in.noise": noise.numpy(), "in.jitter": inp["jitter"].numpy(), "out.normal": out["normal"].detach().numpy(),
            "out.rgb": out["rgb"].detach().numpy(), "out.loss": loss.detach().numpy()}
    for k in ("K", "pose_r", "pose_t", "scales", "shifts", "depth_img", "img", "ray_idx"):
        blob["in." + k] = inp[k].numpy()
    for k, v in grads.items():
        blob["g." + k] = v.numpy()
    np.savez_compressed(os.path.join(gg.OUT, "normal_loss_d128.npz"), **blob)
    print(f"normal_loss_d128: oracle == reference (worst deviation {worst:.2e}), M = {m} of {c['R']} rays; wrote normal_loss_d128.npz")


if __name__ == "__main__":
    main()
