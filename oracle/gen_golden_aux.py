"""Golden vectors for the per-image losses of the first training phase: drives the REFERENCE
Trainer.train_step (model/training.py:165-377) on CPU with pc_weight = rgb_s_weight = 1, asserts that the oracle
(train_step_scope + aux_scope) reproduces its loss parts and pose / distortion gradients, and freezes the inputs, loss parts
and gradients in tests/golden/aux_terms.npz.  Run in the authoring container only:  python oracle/gen_golden_aux.py"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import nerf_oracle as orc  # noqa: E402

N_CAMS = 6
H, W, R, N = 48, 64, 64, 32        # depth maps at image resolution: res = 12 x 16 = 192 points per cloud


def inputs(seed):
    g = torch.Generator().manual_seed(seed)
    f = 0.7 * W
    # a smooth surface + noise, so that re-projected points land inside the other image and the clouds overlap
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    surf = lambda: (2.0 + 0.4 * torch.sin(2 * xs + 0.3) * torch.cos(1.5 * ys) + 0.05 * torch.rand(H, W, generator=g)).view(1, H, W)
    return {
        "K": torch.diag(torch.tensor([2 * f / W, -2 * f / H, -1.0, 1.0])).unsqueeze(0),
        "img": torch.rand(1, 3, H, W, generator=g), "ref_img": torch.rand(1, 3, H, W, generator=g),
        "dpt": surf(), "ref_dpt": surf(),
        "pose_r": 0.02 * torch.randn(N_CAMS, 3, generator=g), "pose_t": 0.05 * torch.randn(N_CAMS, 3, generator=g),
        "scales": 1 + 0.05 * torch.randn(N_CAMS, 1, generator=g), "shifts": 0.05 * torch.randn(N_CAMS, 1, generator=g),
    }


def run(ref, name, cam, ref_idx, seed, blob):
    cfg = copy.deepcopy(gg.base_cfg(128))
    cfg["training"].update(n_training_points=R, pc_weight=[1.0, 0.0], rgb_s_weight=[1.0, 0.0], vis_reprojection_every=10 ** 9)
    cfg["rendering"]["num_points"] = N
    inp = inputs(seed)
    dev = torch.device("cpu")
    torch.manual_seed(42)
    net = ref.OfficialStaticNerf(cfg)
    model = ref.get_model(ref.Renderer(net, cfg["rendering"], device=dev), cfg, device=dev)
    pose, dist = ref.LearnPose(N_CAMS, True, True, cfg), ref.Learn_Distortion(N_CAMS, True, True, cfg)
    with torch.no_grad():
        pose.r.copy_(inp["pose_r"]); pose.t.copy_(inp["pose_t"])
        dist.global_scales.copy_(inp["scales"]); dist.global_shifts.copy_(inp["shifts"])
    sgd = lambda m: torch.optim.SGD(m.parameters(), lr=0.0)
    tr = ref.Trainer(model, sgd(model), cfg["training"], device=dev, optimizer_pose=sgd(pose), pose_param_net=pose,
                     optimizer_distortion=sgd(dist), distortion_net=dist)
    data = {"img": inp["img"], "img.idx": cam, "img.dpt": inp["dpt"], "img.camera_mat": inp["K"],
            "img.scale_mat": torch.eye(4).unsqueeze(0), "img.ref_imgs": inp["ref_img"], "img.ref_dpts": inp["ref_dpt"],
            "img.ref_idxs": ref_idx}
    torch.manual_seed(7)
    ld = tr.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
    # oracle: the render slice on the replayed draws + the per-image terms
    torch.manual_seed(7)
    ray_idx = torch.randperm(H * W)[:R]
    jitter = torch.rand(1, R, N)
    weights = {k: v.detach().clone() for k, v in net.state_dict().items()}
    params = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
    rc = {**cfg["rendering"]}
    loss_main, out = orc.train_step_scope(params, leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], cam,
                                          inp["K"], inp["dpt"].unsqueeze(1), inp["img"], (H, W), ray_idx, jitter, rc)
    aux, l_pc, l_rgbs = orc.aux_scope(leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], cam, ref_idx,
                                      inp["K"], inp["dpt"].unsqueeze(1), inp["ref_dpt"].unsqueeze(1), inp["img"], inp["ref_img"])
    (loss_main + aux).backward()
    gg.check(f"{name}.loss", loss_main + aux, ld["loss"], 1e-6)
    gg.check(f"{name}.loss_pc", l_pc, ld["loss_pc"], 1e-6)
    gg.check(f"{name}.loss_rgb_s", l_rgbs, ld["loss_rgb_s"], 1e-6)
    ref_g = {"pose_r": pose.r.grad, "pose_t": pose.t.grad, "scales": dist.global_scales.grad, "shifts": dist.global_shifts.grad}
    z = lambda g, k: g if g is not None else torch.zeros_like(inp[k])   # e.g. the last camera's scale is the constant 1
    ref_g = {k: z(v, k) for k, v in ref_g.items()}
    for k, v in ref_g.items():
        gg.check(f"{name}.grad.{k}", z(leaves[k].grad, k), v, 2e-5)
    # the per-image terms alone (what the fused kernels are compared with)
    leaves2 = {k: inp[k].clone().requires_grad_(True) for k in leaves}
    aux2, _, _ = orc.aux_scope(leaves2["pose_r"], leaves2["pose_t"], leaves2["scales"], leaves2["shifts"], cam, ref_idx, inp["K"],
                               inp["dpt"].unsqueeze(1), inp["ref_dpt"].unsqueeze(1), inp["img"], inp["ref_img"])
    aux2.backward()
    for k, v in inp.items():
        blob[f"{name}.in.{k}"] = v.numpy()
    blob[f"{name}.cam"], blob[f"{name}.ref"] = cam, ref_idx
    blob[f"{name}.ray_idx"], blob[f"{name}.jitter"] = ray_idx.numpy(), jitter.numpy()
    for k in ("loss", "loss_pc", "loss_rgb_s", "loss_rgb", "loss_depth"):
        blob[f"{name}.out.{k}"] = ld[k].detach().numpy()
    for k, v in ref_g.items():
        blob[f"{name}.g.{k}"] = v.numpy()                                   # full step (render + per-image terms)
        blob[f"{name}.gaux.{k}"] = (leaves2[k].grad if leaves2[k].grad is not None else torch.zeros_like(inp[k])).numpy()
    print(f"aux case {name}: cam {cam} ref {ref_idx}: loss_pc {float(ld['loss_pc']):.6f} loss_rgb_s {float(ld['loss_rgb_s']):.6f}; "
          "oracle == reference")
    return weights


def run_ssim(ref, name, cam, ref_idx, seed, blob):
    """The same step with training.with_ssim: True (model/losses.py:150-157,222-252: 0.15 |diff| + 0.85 SSIM dissimilarity in the
    surface re-projection term).  Same inputs and draws as case `name`; only the reference's outputs are recorded -- the SSIM
    branch is plain torch on both sides and is not part of the oracle restatement."""
    cfg = copy.deepcopy(gg.base_cfg(128))
    cfg["training"].update(n_training_points=R, pc_weight=[1.0, 0.0], rgb_s_weight=[1.0, 0.0], vis_reprojection_every=10 ** 9,
                           with_ssim=True)
    cfg["rendering"]["num_points"] = N
    inp = inputs(seed)
    dev = torch.device("cpu")
    torch.manual_seed(42)
    net = ref.OfficialStaticNerf(cfg)
    model = ref.get_model(ref.Renderer(net, cfg["rendering"], device=dev), cfg, device=dev)
    pose, dist = ref.LearnPose(N_CAMS, True, True, cfg), ref.Learn_Distortion(N_CAMS, True, True, cfg)
    with torch.no_grad():
        pose.r.copy_(inp["pose_r"]); pose.t.copy_(inp["pose_t"])
        dist.global_scales.copy_(inp["scales"]); dist.global_shifts.copy_(inp["shifts"])
    sgd = lambda m: torch.optim.SGD(m.parameters(), lr=0.0)
    tr = ref.Trainer(model, sgd(model), cfg["training"], device=dev, optimizer_pose=sgd(pose), pose_param_net=pose,
                     optimizer_distortion=sgd(dist), distortion_net=dist)
    data = {"img": inp["img"], "img.idx": cam, "img.dpt": inp["dpt"], "img.camera_mat": inp["K"],
            "img.scale_mat": torch.eye(4).unsqueeze(0), "img.ref_imgs": inp["ref_img"], "img.ref_dpts": inp["ref_dpt"],
            "img.ref_idxs": ref_idx}
    torch.manual_seed(7)
    ld = tr.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
    for k in ("loss", "loss_pc", "loss_rgb_s", "loss_rgb", "loss_depth"):
        blob[f"{name}_ssim.out.{k}"] = ld[k].detach().numpy()
    for k, t in (("pose_r", pose.r), ("pose_t", pose.t), ("scales", dist.global_scales), ("shifts", dist.global_shifts)):
        blob[f"{name}_ssim.g.{k}"] = (t.grad if t.grad is not None else torch.zeros_like(t)).numpy()
    print(f"aux case {name} with SSIM: loss_rgb_s {float(ld['loss_rgb_s']):.6f} (plain: {float(blob[name + '.out.loss_rgb_s']):.6f})")


def main():
    ref = gg.import_reference()
    torch.set_num_threads(8)
    blob = {}
    w = run(ref, "mid", 2, 3, 21, blob)          # ordinary frame: frame 1 = current, frame 2 = reference
    run_ssim(ref, "mid", 2, 3, 21, blob)
    run(ref, "last", N_CAMS - 1, N_CAMS - 2, 22, blob)   # last camera: roles swapped, scale fixed to 1 (distortions.py:24-25)
    np.savez_compressed(os.path.join(gg.OUT, "aux_terms.npz"), **blob)
    base = np.load(os.path.join(gg.OUT, "weights_d128.npz"))   # same seed-42 D=128 network as the render cases
    assert all(np.array_equal(base[k], v.numpy()) for k, v in w.items())
    print("wrote tests/golden/aux_terms.npz")


if __name__ == "__main__":
    main()
