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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
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
