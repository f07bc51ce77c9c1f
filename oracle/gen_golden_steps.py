"""Golden vectors for a SEQUENCE of training steps (the loop of reference train.py:205-210 around model/training.py:71-98): the
REFERENCE Trainer with its three Adam optimisers (lr 1e-3 / 5e-4 / 5e-4, configs/default.yaml:79-82) runs K steps on CPU over
changing frames, first-phase losses on; every step's random draws (pixel permutation, jitter) are recorded so that another
implementation can replay them.  Frozen in tests/golden/train_steps.npz: inputs, draws, the per-step loss dictionaries and the
parameters after the first and the last step.  What single-step goldens cannot see -- stale packed weights, optimiser state,
gradient accumulation across steps -- shows up here.  Authoring container only:  python oracle/gen_golden_steps.py"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
import gen_golden_aux as ga  # noqa: E402

STEPS = [(2, 3), (0, 1), (5, 4), (3, 4), (2, 3), (4, 5)]       # (frame, neighbour); (5, 4) is the last-camera role swap
LOGGED = ("loss", "loss_rgb", "loss_depth", "loss_pc", "loss_rgb_s", "l2_mean")


def frames(seed):
    """Six frames of a smooth surface: image + mono depth per camera."""
    g = torch.Generator().manual_seed(seed)
    H, W = ga.H, ga.W
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, H), torch.linspace(-1, 1, W), indexing="ij")
    imgs, dpts = [], []
    for c in range(ga.N_CAMS):
        imgs.append(torch.stack([0.5 + 0.4 * torch.sin(3 * xs + 0.3 * c) * torch.cos(2 * ys), 0.5 + 0.4 * torch.sin(2 * xs + ys + 0.2 * c),
                                 0.5 + 0.4 * torch.cos(xs - 2 * ys)], 0) + 0.02 * torch.rand(3, H, W, generator=g))
        dpts.append(2.0 + 0.4 * torch.sin(2 * xs + 0.3 + 0.1 * c) * torch.cos(1.5 * ys) + 0.05 * torch.rand(H, W, generator=g))
    return torch.stack(imgs).clamp(0, 1), torch.stack(dpts)


NDC_STEPS = [(2, 3), (4, 5), (5, 4)]
NDC_RENDERING = dict(sample_option="ndc", dist_alpha=True, depth_range=[0.0, 1.0])       # configs/LLFF/fern.yaml
NDC_TENSORS = ("layers0.0.weight", "layers1.0.weight", "fc_density.weight", "rgb_layers.0.weight", "fc_rgb.weight", "layers1.6.bias")


def main():
    ref = gg.import_reference()
    torch.set_num_threads(8)
    blob = run(ref, STEPS, {}, 31, 32)
    ndc = run(ref, NDC_STEPS, NDC_RENDERING, 31, 32)          # same frames and initial parameters, stored once
    base = np.load(os.path.join(gg.OUT, "weights_d128.npz"))
    # the LLFF-style sequence rides along under "ndc.": draws, losses, small parameters, the moves of a few network tensors
    for k, v in ndc.items():
        if k.startswith("final.net.") or k.startswith("after1.") or k.startswith("init.") or k in ("imgs", "dpts", "K"):
            continue
        blob["ndc." + k] = v
    for k in NDC_TENSORS:
        blob["ndc.final.delta_f16." + k] = (ndc["final.net." + k] - base[k]).astype(np.float16)
    finish(blob, base)


def run(ref, STEPS, rendering, seed_inputs, seed_frames):
    cfg = copy.deepcopy(gg.base_cfg(128))
    cfg["training"].update(n_training_points=ga.R, pc_weight=[1.0, 0.0], rgb_s_weight=[1.0, 0.0], vis_reprojection_every=10 ** 9)
    cfg["rendering"]["num_points"] = ga.N
    cfg["rendering"].update(rendering)
    inp = ga.inputs(seed_inputs)
    imgs, dpts = frames(seed_frames)
    dev = torch.device("cpu")
    torch.manual_seed(42)
    net = ref.OfficialStaticNerf(cfg)
    model = ref.get_model(ref.Renderer(net, cfg["rendering"], device=dev), cfg, device=dev)
    pose, dist = ref.LearnPose(ga.N_CAMS, True, True, cfg), ref.Learn_Distortion(ga.N_CAMS, True, True, cfg)
    with torch.no_grad():
        pose.r.copy_(inp["pose_r"]); pose.t.copy_(inp["pose_t"])
        dist.global_scales.copy_(inp["scales"]); dist.global_shifts.copy_(inp["shifts"])
    adam = lambda m, lr: torch.optim.Adam(m.parameters(), lr=lr)
    tr = ref.Trainer(model, adam(model, 1e-3), cfg["training"], device=dev, optimizer_pose=adam(pose, 5e-4), pose_param_net=pose,
                     optimizer_distortion=adam(dist, 5e-4), distortion_net=dist)
    base = np.load(os.path.join(gg.OUT, "weights_d128.npz"))
    assert all(np.array_equal(base[k], v.numpy()) for k, v in net.state_dict().items())    # the seed-42 D=128 network
    drawn_reset = True
    blob = {"imgs": imgs.numpy(), "dpts": dpts.numpy(), "K": inp["K"].numpy(), "steps": np.array(STEPS)}
    for k in ("pose_r", "pose_t", "scales", "shifts"):
        blob["init." + k] = inp[k].numpy()
    real_randperm, real_rand = torch.randperm, torch.rand
    drawn = {}
    del drawn_reset

    def randperm(n, *a, **k):
        drawn["perm"] = real_randperm(n, *a, **k)
        return drawn["perm"]

    def rand(*s, **k):
        out = real_rand(*s, **k)
        if tuple(s) == (1, ga.R, ga.N):
            drawn["jitter"] = out
        return out

    def snapshot(tag):
        for k, v in net.state_dict().items():
            blob[f"{tag}.net.{k}"] = v.detach().numpy().copy()
        for k, v in (("pose_r", pose.r), ("pose_t", pose.t), ("scales", dist.global_scales), ("shifts", dist.global_shifts)):
            blob[f"{tag}.{k}"] = v.detach().numpy().copy()

    torch.randperm, torch.rand = randperm, rand
    try:
        for s, (cam, nb) in enumerate(STEPS):
            torch.manual_seed(100 + s)
            data = {"img": imgs[cam:cam + 1], "img.idx": cam, "img.dpt": dpts[cam:cam + 1], "img.camera_mat": inp["K"],
                    "img.scale_mat": torch.eye(4).unsqueeze(0), "img.ref_imgs": imgs[nb:nb + 1], "img.ref_dpts": dpts[nb:nb + 1],
                    "img.ref_idxs": nb}
            ld = tr.train_step(data, it=s + 1, epoch=0, scheduling_start=10000, render_path=None)   # it = 0 would dump the re-projection PNGs
            blob[f"s{s}.ray_idx"] = drawn["perm"][:ga.R].numpy().copy()
            if "jitter" in drawn:                                    # NDC sampling draws none (rendering.py:168-180)
                blob[f"s{s}.jitter"] = drawn["jitter"].numpy().copy()
            for k in LOGGED:
                blob[f"s{s}.{k}"] = np.float64(float(ld[k]))
            print(f"step {s} cam {cam} nb {nb}: " + "  ".join(f"{k} {float(ld[k]):.6f}" for k in LOGGED))
            if s == 0:
                snapshot("after1")
    finally:
        torch.randperm, torch.rand = real_randperm, real_rand
    snapshot("final")
    return blob


def finish(blob, base):
    # the first-step and whole-run updates of the network, for scale
    d1 = np.concatenate([(blob[f"after1.net.{k}"] - base[k]).ravel() for k in base.files])
    print("after step 1: |update| max %.3e (lr = 1e-3: Adam's first step moves every touched entry by lr)" % np.abs(d1).max())
    keep = {k: v for k, v in blob.items() if not (k.startswith("after1.net.") or k.startswith("final.net."))}
    # network snapshots, compactly: after the first step the SIGN of every entry's move (Adam's first step is lr * g / (|g| + eps):
    # +-lr wherever the gradient is not tiny) as bit planes; after the last step the move itself in float16 (|move| <= 6e-3, so
    # the 11-bit mantissa resolves 3e-6 -- the check is at 1e-4)
    order = sorted(base.files)
    keep["net.order"] = np.array(order)
    d1 = np.concatenate([(blob[f"after1.net.{k}"] - base[k]).ravel() for k in order])
    dK = np.concatenate([(blob[f"final.net.{k}"] - base[k]).ravel() for k in order])
    keep["after1.net.up"], keep["after1.net.down"] = np.packbits(d1 > 0.5e-3), np.packbits(d1 < -0.5e-3)
    keep["final.net.delta_f16"] = dK.astype(np.float16)
    print("entries moved by the first step: %d up, %d down, %d of %d hardly (|g| ~ eps)" %
          ((d1 > 0.5e-3).sum(), (d1 < -0.5e-3).sum(), (np.abs(d1) <= 0.5e-3).sum(), d1.size))
    np.savez_compressed(os.path.join(gg.OUT, "train_steps.npz"), **keep)
    print("wrote tests/golden/train_steps.npz", os.path.getsize(os.path.join(gg.OUT, "train_steps.npz")), "bytes")


if __name__ == "__main__":
    main()
