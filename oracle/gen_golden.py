"""Pin the oracle against the real reference and freeze golden vectors.

TEST INFRASTRUCTURE.  Run in the authoring container only (needs /root/reference):

    python oracle/gen_golden.py            # writes tests/golden/*.npz

For every case it (1) imports the *actual* reference modules on CPU (MagicMock stubs for the
image/IO dependencies the hot path never touches -- SURVEY.md section 8c), (2) runs the reference
render + loss heads + backward on seeded inputs, (3) runs oracle/nerf_oracle.py on the same
inputs and asserts agreement, (4) stores inputs, outputs and gradients in tests/golden/.
One case additionally drives the reference's own Trainer.train_step (aux losses weighted 0)
to prove that the slice restated in `train_step_scope` is what the trainer executes.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
from functools import partial
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("NNR_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)

import nerf_oracle as orc  # noqa: E402


def import_reference():
    for name in ("cv2", "imageio", "timm", "timm.models", "timm.models.layers", "torchvision",
                 "torchvision.transforms", "lpips", "skimage", "skimage.metrics", "matplotlib",
                 "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = MagicMock()
                m.__spec__ = importlib.machinery.ModuleSpec(name, None)
                sys.modules[name] = m
    sys.path.insert(0, REF)
    import model as ref_model  # noqa
    import model.losses as ref_losses  # noqa
    import model.training as ref_training  # noqa
    from model.common import transform_to_world
    torch.Tensor.cuda = lambda self, *a, **k: self
    ref_training.transform_to_world = partial(transform_to_world, device=torch.device("cpu"))
    return ref_model


def base_cfg(hidden):
    import yaml
    with open(os.path.join(REF, "configs", "default.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["model"]["hidden_dim"] = hidden
    cfg["depth"]["type"] = "None"
    return cfg


CASES = {
    # name: dict(hidden, R, N, h, w, hd, wd, rendering overrides, flags)
    "tanks_d128": dict(hidden=128, R=32, N=64, h=60, w=80, hd=30, wd=40, rend={}, jitter=True),
    "tanks_eval_d128": dict(hidden=128, R=32, N=64, h=60, w=80, hd=30, wd=40, rend={}, jitter=False, eval_=True),
    "llff_ndc_d128": dict(hidden=128, R=32, N=64, h=60, w=80, hd=60, wd=80,
                          rend=dict(depth_range=[0.0, 1.0], dist_alpha=True, sample_option="ndc"), jitter=True,
                          far_cam=True),
    "uniform_distalpha_masked_d128": dict(hidden=128, R=48, N=40, h=60, w=80, hd=30, wd=40,
                                          rend=dict(dist_alpha=True), jitter=True, bad_depth=True),
    "masked_inf_eval_d128": dict(hidden=128, R=48, N=40, h=60, w=80, hd=30, wd=40, rend={}, jitter=False,
                                 eval_=True, bad_depth=True, inf_depth=True),
    "white_nonorm_d128": dict(hidden=128, R=32, N=33, h=60, w=80, hd=30, wd=40,
                              rend=dict(white_background=True, normalise_ray=False), jitter=True),
    "zero_pose_d128": dict(hidden=128, R=32, N=64, h=60, w=80, hd=30, wd=40, rend={}, jitter=True, zero_pose=True),
    "tanks_d256_n192": dict(hidden=256, R=16, N=192, h=60, w=80, hd=30, wd=40, rend={}, jitter=True),
    # the two remaining renderer switches: view direction replaced by ones (rendering.py:104-105), relu density (official_nerf.py:77-80)
    "noraydir_relu_d128": dict(hidden=128, R=32, N=48, h=60, w=80, hd=30, wd=40, rend=dict(use_ray_dir=False), jitter=True,
                               model=dict(occ_activation="relu")),
}
N_CAMS = 4
SUBSAMPLE = 2048  # grads of the D=256 case are stored on a fixed stride to keep the fixture small


def make_inputs(c, seed=42):
    g = torch.Generator().manual_seed(seed)
    h, w = c["h"], c["w"]
    f = 0.7 * w
    K = torch.diag(torch.tensor([2 * f / w, -2 * f / h, -1.0, 1.0])).unsqueeze(0)
    pose_r = 0.01 * torch.randn(N_CAMS, 3, generator=g)
    pose_t = 0.01 * torch.randn(N_CAMS, 3, generator=g)
    if c.get("zero_pose"):
        pose_r.zero_(), pose_t.zero_()
    if c.get("far_cam"):
        pose_t[:, 2] += 0.5  # forward-facing NDC scene: keep o_z away from the near plane singularity
    scales = 1 + 0.05 * torch.randn(N_CAMS, 1, generator=g)
    shifts = 0.05 * torch.randn(N_CAMS, 1, generator=g)
    depth_img = 1 + 2 * torch.rand(1, 1, c["hd"], c["wd"], generator=g)
    if c.get("bad_depth"):
        flat = depth_img.view(-1)
        bad = torch.randperm(flat.numel(), generator=g)[: flat.numel() // 10]
        flat[bad[::2]] = 0.0
        if c.get("inf_depth"):   # the reference's own backward is NaN with inf depths: forward-only case
            flat[bad[1::2]] = float("inf")
    img = torch.rand(1, 3, h, w, generator=g)
    ray_idx = torch.randperm(h * w, generator=g)[: c["R"]]
    if c.get("bad_depth"):
        # exactly-zero distorted depth needs scale*0+shift == 0 -> use shift 0 for the camera under test
        shifts[1] = 0.0
    jitter = torch.rand(1, c["R"], c["N"], generator=torch.Generator().manual_seed(seed + 1)) if c["jitter"] else None
    return dict(K=K, pose_r=pose_r, pose_t=pose_t, scales=scales, shifts=shifts, depth_img=depth_img, img=img,
                ray_idx=ray_idx, jitter=jitter, cam=1)


def run_reference(ref, c, inp, cfg):
    torch.manual_seed(42)
    net = ref.OfficialStaticNerf(cfg)
    renderer = ref.Renderer(net, cfg["rendering"], device=torch.device("cpu"))
    model = ref.get_model(renderer, cfg, device=torch.device("cpu"))
    pose = ref.LearnPose(N_CAMS, True, True, cfg)
    dist = ref.Learn_Distortion(N_CAMS, True, True, cfg)
    with torch.no_grad():
        pose.r.copy_(inp["pose_r"]); pose.t.copy_(inp["pose_t"])
        dist.global_scales.copy_(inp["scales"]); dist.global_shifts.copy_(inp["shifts"])
    h, w = c["h"], c["w"]
    cam = inp["cam"]
    eval_ = c.get("eval_", False)
    from model.common import arange_pixels
    from model.losses import Loss

    c2w = pose(cam)
    world_mat = torch.inverse(c2w).unsqueeze(0)
    sc, sh = dist(cam)
    depth_in = inp["depth_img"] * sc + sh
    ray_idx = inp["ray_idx"]
    rgb_gt = inp["img"].view(1, 3, h * w).permute(0, 2, 1)[:, ray_idx]
    p = arange_pixels((h, w), 1)[1][:, ray_idx]
    if inp["jitter"] is not None:
        # the renderer draws torch.rand(1,R,N) as its first RNG use: replay the fixture's generator state
        torch.manual_seed(43)
        chk = torch.rand(1, c["R"], c["N"])
        assert torch.equal(chk, inp["jitter"]), "jitter replay mismatch"
        torch.manual_seed(43)
    out = model(p, ray_idx, inp["K"], world_mat, torch.eye(4).unsqueeze(0), "nope_nerf", it=0, eval_mode=eval_,
                depth_img=depth_in, add_noise=inp["jitter"] is not None, img_size=(h, w))
    res = {"rgb": out["rgb"], "depth_pred": out["depth_pred"], "depth_gt": out["depth_gt"],
           "alpha": out["alpha"], "z_vals": out["z_vals"]}
    grads = {}
    if not eval_:
        crit = Loss(cfg["training"])
        lrgb = crit.get_rgb_full_loss(out["rgb"], rgb_gt, "l1")
        ldep = crit.get_depth_loss(out["depth_pred"], out["depth_gt"])
        loss = 1.0 * lrgb + 0.04 * ldep
        loss.backward()
        res["loss"] = loss.detach()
        for n, p_ in net.named_parameters():
            grads["w." + n] = p_.grad.clone()
        grads["pose_r"] = pose.r.grad.clone()
        grads["pose_t"] = pose.t.grad.clone()
        grads["scales"] = dist.global_scales.grad.clone()
        grads["shifts"] = dist.global_shifts.grad.clone()
    weights = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return res, grads, weights


def run_oracle(c, inp, cfg, weights):
    params = {k: v.clone().requires_grad_(True) for k, v in weights.items()}
    leaves = {k: inp[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
    rcfg = dict(cfg["rendering"])
    rcfg["occ_activation"] = cfg["model"]["occ_activation"]
    eval_ = c.get("eval_", False)
    if eval_:
        h, w = c["h"], c["w"]
        with torch.no_grad():
            c2w = orc.pose_c2w(leaves["pose_r"][inp["cam"]], leaves["pose_t"][inp["cam"]])
            sc, sh = orc.distortion(leaves["scales"], leaves["shifts"], inp["cam"], N_CAMS)
            depth = orc.nearest_gather(inp["depth_img"] * sc + sh, (h, w), inp["ray_idx"])
            out = orc.render(params, orc.pixel_grid(h, w)[:, inp["ray_idx"]], depth, inp["K"],
                             torch.inverse(c2w).unsqueeze(0), torch.eye(4).unsqueeze(0), rcfg, jitter=None, eval_=True)
        return out, {}
    loss, out = orc.train_step_scope(params, leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"],
                                     inp["cam"], inp["K"], inp["depth_img"], inp["img"], (c["h"], c["w"]),
                                     inp["ray_idx"], inp["jitter"], rcfg)
    loss.backward()
    out["loss"] = loss.detach()
    grads = {"w." + k: v.grad for k, v in params.items()}
    grads.update({k: v.grad for k, v in leaves.items()})
    return out, grads


def check(name, a, b, tol):
    a, b = a.detach().double(), b.detach().double()
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, f"{name}: oracle deviates from reference by {err:.3e} (scale {scale:.3e})"
    return err


def trainer_crosscheck(ref, cfg):
    """Drive the reference's own Trainer.train_step with aux losses off and compare the grads
    with train_step_scope on the replayed randperm/rand draws (model/training.py:256-262)."""
    import copy
    c = CASES["tanks_d128"]
    cfg = copy.deepcopy(cfg)
    cfg["training"].update(n_training_points=c["R"], pc_weight=[0.0, 0.0], rgb_s_weight=[0.0, 0.0])
    cfg["rendering"]["num_points"] = c["N"]
    inp = make_inputs(c)
    torch.manual_seed(42)
    net = ref.OfficialStaticNerf(cfg)
    model = ref.get_model(ref.Renderer(net, cfg["rendering"], device=torch.device("cpu")), cfg, device=torch.device("cpu"))
    pose = ref.LearnPose(N_CAMS, True, True, cfg)
    dist = ref.Learn_Distortion(N_CAMS, True, True, cfg)
    with torch.no_grad():
        pose.r.copy_(inp["pose_r"]); pose.t.copy_(inp["pose_t"])
        dist.global_scales.copy_(inp["scales"]); dist.global_shifts.copy_(inp["shifts"])
    sgd = lambda m: torch.optim.SGD(m.parameters(), lr=0.0)
    tr = ref.Trainer(model, sgd(model), cfg["training"], device=torch.device("cpu"), optimizer_pose=sgd(pose),
                     pose_param_net=pose, optimizer_distortion=sgd(dist), distortion_net=dist)
    data = {"img": inp["img"], "img.idx": inp["cam"], "img.dpt": inp["depth_img"][:, 0],
            "img.camera_mat": inp["K"], "img.scale_mat": torch.eye(4).unsqueeze(0)}
    torch.manual_seed(7)
    ld = tr.train_step(data, it=0, epoch=0, scheduling_start=10000, render_path=None)
    torch.manual_seed(7)
    inp["ray_idx"] = torch.randperm(c["h"] * c["w"])[: c["R"]]
    inp["jitter"] = torch.rand(1, c["R"], c["N"])
    weights = {k: v.detach().clone() for k, v in net.state_dict().items()}
    out, grads = run_oracle(c, inp, cfg, weights)
    check("trainer.loss", out["loss"], ld["loss"], 1e-6)
    for n, p_ in net.named_parameters():
        check("trainer.grad." + n, grads["w." + n], p_.grad, 2e-5)
    check("trainer.grad.r", grads["pose_r"], pose.r.grad, 2e-5)
    check("trainer.grad.t", grads["pose_t"], pose.t.grad, 2e-5)
    check("trainer.grad.scale", grads["scales"], dist.global_scales.grad, 2e-5)
    check("trainer.grad.shift", grads["shifts"], dist.global_shifts.grad, 2e-5)
    print("trainer cross-check: oracle train_step_scope == reference Trainer.train_step (aux losses off)")


def main(only=()):
    """`python oracle/gen_golden.py [case ...]`: all cases + the trainer cross-check, or just the named cases (the fixtures of
    the others stay byte-identical in git)."""
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    torch.set_num_threads(8)
    worst = 0.0
    written = {}
    for name, c in CASES.items():
        if only and name not in only:
            continue
        cfg = base_cfg(c["hidden"])
        cfg["rendering"].update(c["rend"])
        cfg["model"].update(c.get("model", {}))
        cfg["rendering"]["num_points"] = c["N"]
        inp = make_inputs(c)
        res, grads, weights = run_reference(ref, c, inp, cfg)
        out, ograds = run_oracle(c, inp, cfg, weights)
        for k in ("rgb", "depth_pred", "depth_gt", "alpha", "z_vals"):
            worst = max(worst, check(f"{name}.{k}", out[k], res[k], 1e-6))
        if grads:
            worst = max(worst, check(f"{name}.loss", out["loss"], res["loss"], 1e-6))
            for k, v in grads.items():
                worst = max(worst, check(f"{name}.grad.{k}", ograds[k], v, 2e-5))
        blob = {"cfg.hidden": c["hidden"], "cfg.R": c["R"], "cfg.N": c["N"], "cfg.h": c["h"], "cfg.w": c["w"],
                "cfg.cam": inp["cam"], "cfg.eval": int(c.get("eval_", False)),
                "cfg.dist_alpha": int(cfg["rendering"]["dist_alpha"]),
                "cfg.ndc": int(cfg["rendering"]["sample_option"] == "ndc"),
                "cfg.white": int(cfg["rendering"]["white_background"]),
                "cfg.normalise_ray": int(cfg["rendering"]["normalise_ray"]),
                "cfg.use_ray_dir": int(cfg["rendering"]["use_ray_dir"]), "cfg.occ_activation": cfg["model"]["occ_activation"],
                "cfg.near": cfg["rendering"]["depth_range"][0], "cfg.far": cfg["rendering"]["depth_range"][1]}
        for k in ("K", "pose_r", "pose_t", "scales", "shifts", "depth_img", "img", "ray_idx"):
            blob["in." + k] = inp[k].numpy()
        if inp["jitter"] is not None:
            blob["in.jitter"] = inp["jitter"].numpy()
        big = c["hidden"] > 128
        for k, v in res.items():
            blob["out." + k] = v.detach().numpy()
        wfile = f"weights_d{c['hidden']}.npz"   # every case of one width shares the seed-42 reference weights
        blob["cfg.weights_file"] = wfile
        wpath = os.path.join(OUT, wfile)
        if wfile not in written:
            if only and os.path.exists(wpath):   # partial regeneration: the shared weights file must already be the same network
                old = np.load(wpath)
                assert all(np.array_equal(old[k], v.numpy()) for k, v in weights.items() if k in old.files and "bias" not in k)
                written[wfile] = {k: torch.from_numpy(old[k]) for k in old.files}
            else:
                np.savez_compressed(wpath, **{k: v.numpy() for k, v in weights.items()})
                written[wfile] = {k: v.clone() for k, v in weights.items()}
        for k, v in weights.items():   # tensors that differ (white_background bias) ride along in the case file
            if not torch.equal(v, written[wfile][k]):
                blob["w." + k] = v.numpy()
        for k, v in grads.items():
            g = v.detach().numpy()
            if big and g.size > SUBSAMPLE:
                stride = g.size // SUBSAMPLE
                blob["gsub." + k] = g.reshape(-1)[::stride].copy()
                blob["gnorm." + k] = np.float64(np.linalg.norm(g.astype(np.float64)))
            else:
                blob["g." + k] = g
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **blob)
        print(f"{name}: oracle == reference (max abs dev so far {worst:.2e}); wrote {name}.npz")
    if not only:
        trainer_crosscheck(ref, base_cfg(128))
    print("done; worst deviation oracle vs reference:", worst)


if __name__ == "__main__":
    main(tuple(sys.argv[1:]))
