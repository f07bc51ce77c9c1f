"""The loss-side switches of the training step against goldens from the reference Trainer (oracle/gen_golden_switches.py ->
tests/golden/loss_switches.npz): weight annealing mid-way, the L2 phase after the switch, depth_loss_type 'invariant', the
trajectory-smoothness terms, detach_gt_depth.  CPU: oracle-backed operator (host logic); gpu: the HIP kernels."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in ("nope-nerf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

from gen_golden_switches import CASES, LOGGED, NET  # noqa: E402  (the case table only; nothing of the reference is imported)

GOLD = np.load(os.path.join(HERE, "golden", "loss_switches.npz"))
R, N = 64, 32


def _run(name, dev, monkeypatch):
    from test_aux_terms import _trainer
    over, epoch, start = CASES[name]
    inp = {k[3:]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith("in.")}
    tr, pose, dist = _trainer(inp, dev, **over)
    ray_idx, jitter = torch.from_numpy(GOLD["ray_idx"]), torch.from_numpy(GOLD["jitter"])
    monkeypatch.setattr(torch, "randperm", lambda n, device=None, **kw: torch.cat([ray_idx, torch.zeros(n - R, dtype=torch.int64)]).to(device))
    real_rand = torch.rand
    monkeypatch.setattr(torch, "rand", lambda *s, device=None, **kw: jitter.to(device) if tuple(s) == (1, R, N) else real_rand(*s, device=device, **kw))
    cam, nb = int(GOLD["cam"]), int(GOLD["nb"])
    data = {"img": inp["img"].to(dev), "img.idx": cam, "img.dpt": inp["dpt"].to(dev), "img.camera_mat": inp["K"].to(dev),
            "img.scale_mat": torch.eye(4).unsqueeze(0).to(dev), "img.ref_imgs": inp["ref_img"].to(dev),
            "img.ref_dpts": inp["ref_dpt"].to(dev), "img.ref_idxs": nb}
    ld = tr.train_step(data, it=1, epoch=epoch, scheduling_start=start, render_path=None)
    for k in LOGGED:
        assert abs(float(ld[k].detach()) - float(GOLD[f"{name}.out.{k}"])) <= 1e-5, (name, k, float(ld[k]), float(GOLD[f"{name}.out.{k}"]))
    got = {"pose_r": pose.r.grad, "pose_t": pose.t.grad, "scales": dist.global_scales.grad, "shifts": dist.global_shifts.grad}
    sd = dict(tr.model.renderer.model.named_parameters())
    got.update({"net." + k: sd[k].grad for k in NET})
    for k, g in got.items():
        want = GOLD[f"{name}.g.{k}"]
        g = g.detach().cpu().numpy() if g is not None else np.zeros_like(want)
        assert float(np.abs(g - want).max()) / max(1.0, float(np.abs(want).max())) <= 1e-4, (name, k)


@pytest.mark.parametrize("name", sorted(CASES))
def test_switch_matches_reference_on_the_cpu_stand_in(name, monkeypatch):
    import oracle_backend
    from model import rendering
    monkeypatch.setattr(rendering.nnr, "render_rays", oracle_backend.render_rays)
    _run(name, torch.device("cpu"), monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_switch_matches_reference_on_the_hip_kernels(name, monkeypatch):
    _run(name, torch.device("cuda"), monkeypatch)
