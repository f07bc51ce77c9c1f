"""Helper process for tests/test_dropin_scripts.py (not a pytest module): runs one of the REFERENCE's own, unmodified entry scripts
(train.py, evaluation/eval_poses.py) with this repository's `model`, `dataloading` and `utils_poses` packages on the import path
in place of the reference's -- the drop-in scenario of SURVEY.md 8(b).  By default the HIP render operator is swapped for the
oracle-backed CPU stand-in (tests/oracle_backend.py); DROPIN_BACKEND=hip leaves the HIP kernels in place (the GPU box: the reference's
train.py, unmodified, on libnnr.so).  tensorboard (absent in this image) is replaced by a recorder.  DROPIN_CPU_DRAWS=1 makes every
random draw of the step (pixel permutation, jitter) come from torch's CPU generator and then move to the device, so that a CPU
run and a GPU run of the same script see the same pixels and jitter and can be compared step by step.

    python tests/dropin_runner.py <script> <config.yaml> [extra args]        (cwd = the reference checkout, for configs/default.yaml)
"""
import importlib.machinery
import json
import os
import runpy
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    script, argv = sys.argv[1], sys.argv[2:]
    # our packages first, the reference checkout last: what a user gets who copies model/, dataloading/ and utils_poses/ over theirs
    sys.path[:0] = [os.path.join(ROOT, "nope-nerf_amd"), HERE, os.path.join(ROOT, "oracle")]
    scalars, epoch_marks = [], []
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, tag, value, step):
            scalars.append((tag, float(value), int(step)))
            if tag == "train/loss_pc_epoch":          # train.py:272 logs this once per epoch, right after the epoch's last step
                epoch_marks.append((int(step), time.perf_counter()))

        def add_image(self, *a, **k):
            pass

    tb.SummaryWriter = SummaryWriter
    tb.__spec__ = importlib.machinery.ModuleSpec(tb.__name__, None)
    sys.modules[tb.__name__] = tb
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    # what the scripts name at import time but this image lacks: open3d frusta (used only under --vis: UI), imageio (the mp4 of
    # eval.py), lpips (a VGG checkpoint; the metric is reported as 0 here)
    stub("utils_poses.vis_cam_traj", draw_camera_frustum_geometry=None)
    stub("imageio", mimwrite=lambda *a, **k: None, imwrite=lambda *a, **k: None)
    import torch

    class LPIPS(torch.nn.Module):
        def __init__(self, net="vgg"):
            super().__init__()

        def forward(self, a, b, normalize=False):
            return torch.zeros(())

    stub("lpips", LPIPS=LPIPS)
    sys.path.append(os.environ.get("NNR_REFERENCE", "/root/reference"))    # everything we do not replace (ATE/, third_party/)

    import model
    import dataloading
    import utils_poses.comp_ate
    for mod in (model, dataloading, utils_poses.comp_ate):
        assert mod.__file__.startswith(ROOT), mod.__file__             # ours, not the reference's
    if os.environ.get("DROPIN_BACKEND", "oracle") != "hip":
        import oracle_backend
        from model import rendering
        rendering.nnr.render_rays = oracle_backend.render_rays
    if os.environ.get("DROPIN_CPU_DRAWS") == "1":
        from nnr import sampling
        real_randperm, real_rand = torch.randperm, torch.rand

        def cpu_randperm(n, *a, device=None, **k):
            return real_randperm(n, *a, **k).to(device if device is not None else "cpu")

        def cpu_rand(*shape, device=None, **k):
            return real_rand(*shape, **k).to(device if device is not None else "cpu")

        torch.randperm, torch.rand = cpu_randperm, cpu_rand
        sampling.randperm_prefix = lambda n, r, device: cpu_randperm(n, device=device)[:r]

    sys.argv = [script] + argv
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        out = os.environ.get("DROPIN_SCALARS")
        if out:
            with open(out, "w") as fh:
                json.dump(scalars, fh)
        out = os.environ.get("DROPIN_TIMES")          # [(iterations done, wall clock)] at every epoch end: the loop's rate (tools/loop_rate.py)
        if out:
            with open(out, "w") as fh:
                json.dump(epoch_marks, fh)


if __name__ == "__main__":
    main()
