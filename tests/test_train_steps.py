"""A SEQUENCE of training steps against the reference (oracle/gen_golden_steps.py -> tests/golden/train_steps.npz): six
Trainer.train_step calls with the three Adam optimisers over changing frames, first-phase losses on, every step's pixel permutation
and jitter replayed from the reference run.  Checked: each step's loss dictionary, the direction every network entry moved in the
first step, and all parameters after the last step.  Runs on the CPU with the oracle-backed operator (host logic: loss scaling,
optimiser wiring, pack-cache invalidation) and, marked gpu, on the HIP kernels."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in ("nope-nerf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

GOLD = np.load(os.path.join(HERE, "golden", "train_steps.npz"))
BASE = np.load(os.path.join(HERE, "golden", "weights_d128.npz"))
N_CAMS, R, N = 6, 64, 32
LOGGED = ("loss", "loss_rgb", "loss_depth", "loss_pc", "loss_rgb_s", "l2_mean")


NDC_RENDERING = dict(sample_option="ndc", dist_alpha=True, depth_range=[0.0, 1.0])


def _replay(dev, monkeypatch, prefix="", rendering=None):
    from test_aux_terms import _trainer
    inp = {k: torch.from_numpy(GOLD["init." + k]) for k in ("pose_r", "pose_t", "scales", "shifts")}
    tr, pose, dist = _trainer(inp, dev, adam=True, rendering_overrides=rendering)
    imgs, dpts, K = (torch.from_numpy(GOLD[k]).to(dev) for k in ("imgs", "dpts", "K"))
    real_rand = torch.rand
    losses, after1 = [], None
    for s, (cam, nb) in enumerate(GOLD[prefix + "steps"]):
        cam, nb = int(cam), int(nb)
        ray_idx = torch.from_numpy(GOLD[f"{prefix}s{s}.ray_idx"])
        jitter = torch.from_numpy(GOLD[f"{prefix}s{s}.jitter"]) if f"{prefix}s{s}.jitter" in GOLD.files else None
        monkeypatch.setattr(torch, "randperm", lambda n, device=None, **kw: torch.cat([ray_idx, torch.zeros(n - R, dtype=torch.int64)]).to(device))
        monkeypatch.setattr(torch, "rand", lambda *sh, device=None, **kw: jitter.to(device) if (jitter is not None and tuple(sh) == (1, R, N))
                            else real_rand(*sh, device=device, **kw))
        data = {"img": imgs[cam:cam + 1], "img.idx": cam, "img.dpt": dpts[cam:cam + 1], "img.camera_mat": K,
                "img.scale_mat": torch.eye(4).unsqueeze(0).to(dev), "img.ref_imgs": imgs[nb:nb + 1], "img.ref_dpts": dpts[nb:nb + 1],
                "img.ref_idxs": nb}
        ld = tr.train_step(data, it=s + 1, epoch=0, scheduling_start=10000, render_path=None)
        losses.append({k: float(ld[k].detach()) for k in LOGGED})
        if s == 0:
            after1 = _flat_net(tr.model)
    return losses, after1, _flat_net(tr.model), pose, dist, tr


def _flat_net(model):
    sd = model.renderer.model.state_dict()
    return np.concatenate([sd[k].detach().cpu().numpy().ravel() for k in GOLD["net.order"]])


def _check(losses, after1, final, pose, dist, tr, loss_tol):
    for s, got in enumerate(losses):
        for k in LOGGED:
            assert abs(got[k] - float(GOLD[f"s{s}.{k}"])) <= loss_tol * max(1.0, abs(float(GOLD[f"s{s}.{k}"]))), (s, k, got[k], float(GOLD[f"s{s}.{k}"]))
    init = np.concatenate([BASE[k].ravel() for k in GOLD["net.order"]])
    n = init.size
    up, down = np.unpackbits(GOLD["after1.net.up"])[:n].astype(bool), np.unpackbits(GOLD["after1.net.down"])[:n].astype(bool)
    d1 = after1 - init
    # Adam's first step moves an entry by lr * g / (|g| + 1e-8): the direction must agree wherever the reference moved it;
    # entries whose gradient is at the 1e-8 scale (sums that cancel) may differ between two fp32 evaluation orders
    wrong = (up & (d1 < 0)) | (down & (d1 > 0))
    assert wrong.sum() <= 5e-4 * (up | down).sum(), (int(wrong.sum()), int((up | down).sum()))
    still = ~(up | down)
    assert (np.abs(d1[still]) > 0.5e-3).sum() <= 5e-4 * n
    dK, ref = final - init, GOLD["final.net.delta_f16"].astype(np.float32)
    rel = np.linalg.norm(dK - ref) / np.linalg.norm(ref)
    assert rel <= 5e-3, rel                              # the whole six-step move of the network, relative L2 (the float16 storage
    assert (np.abs(dK - ref) > 2e-4).sum() <= 1e-3 * n   # of the golden alone is 2e-4); entry-wise but for sign-flipped stragglers
    for k, t in (("pose_r", pose.r), ("pose_t", pose.t), ("scales", dist.global_scales), ("shifts", dist.global_shifts)):
        got, want, start = t.detach().cpu().numpy(), GOLD["final." + k], GOLD["init." + k]
        moved = np.abs(want - start).max()
        assert np.abs(got - want).max() <= 0.01 * moved + 1e-7, (k, float(np.abs(got - want).max()), float(moved))


def test_six_steps_match_the_reference_on_the_cpu_stand_in(monkeypatch):
    import oracle_backend
    from model import rendering
    monkeypatch.setattr(rendering.nnr, "render_rays", oracle_backend.render_rays)
    _check(*_replay(torch.device("cpu"), monkeypatch), loss_tol=2e-5)


@pytest.mark.gpu
def test_six_steps_match_the_reference_on_the_hip_kernels(monkeypatch):
    # measured r01: losses to 5e-7, 0 of 119 256 first-step directions differ, six-step move 2.8e-4 relative L2, poses to 1e-8
    _check(*_replay(torch.device("cuda"), monkeypatch), loss_tol=2e-5)


def _check_ndc(losses, after1, final, pose, dist, tr, loss_tol):
    """The LLFF-style sequence (NDC sampling, dist_alpha; three Adam steps): losses, poses / distortions, and the moves of six
    network tensors (float16 deltas in the golden)."""
    for s, got in enumerate(losses):
        for k in LOGGED:
            want = float(GOLD[f"ndc.s{s}.{k}"])
            assert abs(got[k] - want) <= loss_tol * max(1.0, abs(want)), (s, k, got[k], want)
    sd = tr.model.renderer.model.state_dict()
    for key in [k[len("ndc.final.delta_f16."):] for k in GOLD.files if k.startswith("ndc.final.delta_f16.")]:
        ref = GOLD["ndc.final.delta_f16." + key].astype(np.float32)
        got = sd[key].detach().cpu().numpy() - BASE[key]
        # the two layers that read the encodings hold entries whose gradient is at Adam's eps (1e-8) scale, where a 1e-9 change of
        # the gradient moves the update by 1e-5: 0.6 % / 0.2 % relative L2 even for the CPU stand-in; the others sit at the
        # float16 resolution of the golden (2e-4)
        assert np.linalg.norm(got - ref) <= 0.02 * max(np.linalg.norm(ref), 1e-12), key
    for k, t in (("pose_r", pose.r), ("pose_t", pose.t), ("scales", dist.global_scales), ("shifts", dist.global_shifts)):
        got, want, start = t.detach().cpu().numpy(), GOLD["ndc.final." + k], GOLD["init." + k]
        assert np.abs(got - want).max() <= 0.01 * np.abs(want - start).max() + 1e-7, k


def test_ndc_steps_match_the_reference_on_the_cpu_stand_in(monkeypatch):
    import oracle_backend
    from model import rendering
    monkeypatch.setattr(rendering.nnr, "render_rays", oracle_backend.render_rays)
    _check_ndc(*_replay(torch.device("cpu"), monkeypatch, "ndc.", NDC_RENDERING), loss_tol=2e-5)


@pytest.mark.gpu
def test_ndc_steps_match_the_reference_on_the_hip_kernels(monkeypatch):
    _check_ndc(*_replay(torch.device("cuda"), monkeypatch, "ndc.", NDC_RENDERING), loss_tol=2e-5)
