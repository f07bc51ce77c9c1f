"""The REFERENCE's own render path timed on this host's cores:  python tools/cpu_reference_baseline.py [--rays R] [--samples N]

Runs in its own process (bench.py starts it): the staged reference package is called `model`, like this repository's drop-in, and the
two cannot share an interpreter.  What is timed is BASELINE.md section 3 (i): `nope_nerf.forward` (model/network.py ->
model/rendering.py:36-167 -> model/official_nerf.py) on rays from a learnable pose and a distorted mono-depth map, the reference's loss
heads (`Loss.get_rgb_full_loss` L1 x 1.0 + `Loss.get_depth_loss` x 0.04) and `loss.backward()` to the MLP, pose and distortion
parameters -- the scope of the fused kernels, nothing of this repository in the process (no oracle, no libnnr).  cv2 / imageio / timm /
lpips / skimage / DPT, which the reference imports at module level and this path never touches, are MagicMock stubs
(oracle/gen_golden.py::import_reference, the recipe the golden fixtures were generated with).  Prints one JSON object."""
import argparse
import importlib.machinery
import json
import os
import sys
import time
from unittest.mock import MagicMock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.environ.get("NNR_REFERENCE_STAGE", os.path.join(ROOT, "gpurun_stage", "reference_cpu"))
IMG_H, IMG_W, N_CAMS = 540, 960, 16


def import_reference():
    for name in ("cv2", "imageio", "timm", "timm.models", "timm.models.layers", "torchvision", "torchvision.transforms", "lpips",
                 "skimage", "skimage.metrics", "matplotlib", "matplotlib.pyplot", "DPT", "DPT.dpt", "DPT.dpt.models"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = MagicMock()
                m.__spec__ = importlib.machinery.ModuleSpec(name, None)
                sys.modules[name] = m
    sys.path.insert(0, STAGE)
    import model as ref
    assert os.path.abspath(os.path.dirname(ref.__file__)) == os.path.abspath(os.path.join(STAGE, "model")), ref.__file__
    return ref


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            return next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), "")
    except OSError:
        return ""


def _merge(dst, src):      # dataloading/configloading.py:37-47 (update_recursive), restated: the staged copy holds model/ and configs/ only
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def train_step_mode(a):
    """BASELINE configs[0] on the REFERENCE ITSELF: configs/default.yaml + configs/Tanks/Ignatius.yaml, 32 rays x 64 samples, hidden_dim
    128, and the reference's own `Trainer.train_step` (model/training.py:67-97: zero_grad, compute_loss with the first-phase per-image
    losses on -- pc_weight = rgb_s_weight = 1 as the YAML has them --, backward, the three Adam steps train.py:85-118 builds) on a
    540 x 960 frame + neighbour frame with mono-depth maps of the frame's size.  SURVEY.md section 6 probed this at ~760 rays/s on 8
    cores.  Threads: swept over {1, 2, 4, 8, 16} unless --threads is given -- 2 k samples per step do not feed 32 threads (round 3 timed
    the oracle port at 32 threads: 52 rays/s) -- and the best is the reported value, the sweep beside it."""
    import yaml
    host_cores = os.cpu_count() or 1
    ref = import_reference()
    # the reference hard-wires device='cuda' defaults on the per-image path (model/common.py:118, model/training.py:322): the same two
    # call-site patches the golden generators use (oracle/gen_golden.py:50-51) -- bindings of the caller, not edits of the staged sources
    from functools import partial
    import model.training as ref_training
    from model.common import transform_to_world
    torch.Tensor.cuda = lambda self, *a_, **k_: self
    ref_training.transform_to_world = partial(transform_to_world, device=torch.device("cpu"))
    with open(os.path.join(STAGE, "configs", "default.yaml")) as f:
        cfg = yaml.safe_load(f)
    with open(os.path.join(STAGE, "configs", "Tanks", "Ignatius.yaml")) as f:
        _merge(cfg, yaml.safe_load(f))
    cfg["model"]["hidden_dim"] = a.hidden
    cfg["rendering"]["num_points"] = a.samples
    cfg["training"]["n_training_points"] = a.rays
    cfg["training"]["vis_reprojection_every"] = 10 ** 9        # no PNG dumps inside the timed steps
    dev = torch.device("cpu")
    torch.manual_seed(42)
    net = ref.OfficialStaticNerf(cfg)
    model = ref.get_model(ref.Renderer(net, cfg["rendering"], device=dev), cfg, device=dev)
    pose = ref.LearnPose(N_CAMS, True, True, cfg)
    dist = ref.Learn_Distortion(N_CAMS, True, True, cfg)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        pose.r.copy_(0.01 * torch.randn(N_CAMS, 3, generator=g)); pose.t.copy_(0.01 * torch.randn(N_CAMS, 3, generator=g))
        dist.global_scales.copy_(1 + 0.05 * torch.randn(N_CAMS, 1, generator=g)); dist.global_shifts.copy_(0.05 * torch.randn(N_CAMS, 1, generator=g))
    tcfg = cfg["training"]
    opt = torch.optim.Adam(model.parameters(), lr=tcfg["learning_rate"])                 # train.py:85-118
    opt_pose = torch.optim.Adam(pose.parameters(), lr=tcfg["pose_lr"])
    opt_dist = torch.optim.Adam(dist.parameters(), lr=tcfg["distortion_lr"])
    tr = ref.Trainer(model, opt, tcfg, device=dev, optimizer_pose=opt_pose, pose_param_net=pose, optimizer_distortion=opt_dist,
                     distortion_net=dist)
    f = 0.7 * IMG_W
    K = torch.diag(torch.tensor([2 * f / IMG_W, -2 * f / IMG_H, -1.0, 1.0])).unsqueeze(0)
    dh, dw = a.depth_hw      # mono-depth maps of 108 x 192 -> a 27 x 48 grid of the per-image terms (pc_ratio 4): the dense point-to-point
                             # match of model/losses.py:125-148 is O(points^2) in MEMORY -- 12.6 GB at the frame's own 540 x 960
    data = {"img": torch.rand(1, 3, IMG_H, IMG_W, generator=g), "img.idx": 3, "img.dpt": 1 + 2 * torch.rand(1, dh, dw, generator=g),
            "img.camera_mat": K, "img.scale_mat": torch.eye(4).unsqueeze(0), "img.ref_imgs": torch.rand(1, 3, IMG_H, IMG_W, generator=g),
            "img.ref_dpts": 1 + 2 * torch.rand(1, dh, dw, generator=g), "img.ref_idxs": 4}
    sweep, it, ld = {}, 1, None      # it = 0 would dump the re-projection PNGs (training.py:344)
    for threads in ([a.threads] if a.threads else [t for t in (1, 2, 4, 8, 16) if t <= host_cores]):
        torch.set_num_threads(threads)
        ts = []
        for i in range(a.warmup + a.steps):
            t0 = time.perf_counter()
            ld = tr.train_step(data, it=it, epoch=0, scheduling_start=10000, render_path=None)
            ts.append(time.perf_counter() - t0)
            it += 1
        sweep[threads] = float(np.median(ts[a.warmup:]))
    best = min(sweep, key=sweep.get)
    print(json.dumps({
        "workload": "BASELINE configs[0]: configs/Tanks/Ignatius.yaml over configs/default.yaml, %d rays x %d samples, hidden_dim %d, the reference's "
                    "Trainer.train_step (per-image losses on a %d x %d grid, three Adam steps) on a %d x %d frame, CPU"
                    % (a.rays, a.samples, a.hidden, dh // 4, dw // 4, IMG_H, IMG_W),
        "ms_per_step": round(sweep[best] * 1e3, 3), "value": round(a.rays / sweep[best], 1), "unit": "rays/s", "kind": "reference",
        "threads": best, "cores": best, "host_cores": host_cores, "cpu": _cpu_model(),
        "thread_sweep_ms_per_step": {str(k): round(v * 1e3, 3) for k, v in sweep.items()}, "final_loss": round(float(ld["loss"]), 6),
        "sample": "median of %d steps after %d warm-ups per thread count; staged, unmodified copy of /root/reference/model; torch %s CPU"
                  % (a.steps, a.warmup, torch.__version__)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-step", action="store_true", help="BASELINE configs[0]: the reference's Trainer.train_step (see train_step_mode)")
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--depth-hw", type=int, nargs=2, default=[108, 192], help="--train-step: size of the mono-depth maps")
    a = ap.parse_args()
    if a.train_step:
        return train_step_mode(a)
    import yaml
    host_cores = os.cpu_count() or 1
    threads = a.threads or min(host_cores, 32)      # (R N x 256) x (256 x 256) GEMMs: more threads only add synchronisation
    torch.set_num_threads(threads)
    ref = import_reference()
    with open(os.path.join(STAGE, "configs", "default.yaml")) as f:
        cfg = yaml.safe_load(f)
    cfg["model"]["hidden_dim"] = a.hidden
    cfg["depth"]["type"] = "None"
    cfg["rendering"]["num_points"] = a.samples
    dev = torch.device("cpu")
    torch.manual_seed(42)
    net = ref.OfficialStaticNerf(cfg)
    model = ref.get_model(ref.Renderer(net, cfg["rendering"], device=dev), cfg, device=dev)
    pose = ref.LearnPose(N_CAMS, True, True, cfg)
    dist = ref.Learn_Distortion(N_CAMS, True, True, cfg)
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        pose.r.copy_(0.01 * torch.randn(N_CAMS, 3, generator=g)); pose.t.copy_(0.01 * torch.randn(N_CAMS, 3, generator=g))
        dist.global_scales.copy_(1 + 0.05 * torch.randn(N_CAMS, 1, generator=g)); dist.global_shifts.copy_(0.05 * torch.randn(N_CAMS, 1, generator=g))
    f = 0.7 * IMG_W
    K = torch.diag(torch.tensor([2 * f / IMG_W, -2 * f / IMG_H, -1.0, 1.0])).unsqueeze(0)
    depth_img = 1 + 2 * torch.rand(1, 1, IMG_H, IMG_W, generator=g)
    img = torch.rand(1, 3, IMG_H, IMG_W, generator=g)
    from model.common import arange_pixels
    from model.losses import Loss
    crit = Loss(cfg["training"])
    params = list(net.parameters()) + list(pose.parameters()) + list(dist.parameters())
    cam, ts, loss = 3, [], None
    for i in range(a.warmup + a.steps):
        for p_ in params:
            p_.grad = None
        t0 = time.perf_counter()
        ray_idx = torch.randperm(IMG_H * IMG_W, generator=g)[:a.rays]                       # model/training.py:256-262
        world_mat = torch.inverse(pose(cam)).unsqueeze(0)                                   # :238
        sc, sh = dist(cam)
        depth_in = depth_img * sc + sh                                                      # :240-245
        rgb_gt = img.view(1, 3, IMG_H * IMG_W).permute(0, 2, 1)[:, ray_idx]
        p = arange_pixels((IMG_H, IMG_W), 1)[1][:, ray_idx]
        out = model(p, ray_idx, K, world_mat, torch.eye(4).unsqueeze(0), "nope_nerf", it=i, eval_mode=False, depth_img=depth_in,
                    add_noise=True, img_size=(IMG_H, IMG_W))
        loss = 1.0 * crit.get_rgb_full_loss(out["rgb"], rgb_gt, "l1") + 0.04 * crit.get_depth_loss(out["depth_pred"], out["depth_gt"])
        loss.backward()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts[a.warmup:]))
    cpu = ""
    try:
        with open("/proc/cpuinfo") as fh:
            cpu = next((l.split(":", 1)[1].strip() for l in fh if l.startswith("model name")), "")
    except OSError:
        pass
    print(json.dumps({
        "value": round(a.rays / med, 2), "unit": "rays/s", "cores": threads, "host_cores": host_cores, "threads": threads, "cpu": cpu,
        "kind": "reference", "s_per_step": round(med, 4), "final_loss": round(float(loss.detach()), 6),
        "sample": f"the reference's own nope_nerf.forward + loss heads + backward (staged copy of /root/reference/model, unmodified), "
                  f"{a.rays} rays x {a.samples} samples, D={a.hidden}, median of {a.steps} steps after {a.warmup} warm-ups, torch "
                  f"{torch.__version__} CPU, {threads} threads on a {host_cores}-core host"}))


if __name__ == "__main__":
    main()
