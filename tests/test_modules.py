"""CPU: LearnPose / Learn_Distortion / LearnFocal against goldens from the reference classes (oracle/gen_golden_modules.py ->
tests/golden/modules.npz) in the configurations the render goldens do not visit."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in ("nope-nerf_amd", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))

from gen_golden_modules import FOCAL_CASES, SCALES, SHIFTS  # noqa: E402  (tables only; nothing of the reference is imported)

GOLD = np.load(os.path.join(HERE, "golden", "modules.npz"))


@pytest.mark.parametrize("tag,kw", [("composed", dict(learn_R=True, learn_t=True, init=True)), ("frozen_R", dict(learn_R=False, learn_t=True)),
                                    ("frozen_t", dict(learn_R=True, learn_t=False))])
def test_learn_pose(tag, kw):
    import model as mdl
    n = 5
    init = torch.from_numpy(GOLD["pose.init"]) if kw.get("init") else None
    m = mdl.LearnPose(n, kw["learn_R"], kw["learn_t"], {}, init_c2w=init)
    with torch.no_grad():
        m.r.copy_(torch.from_numpy(GOLD["pose.r"])); m.t.copy_(torch.from_numpy(GOLD["pose.t"]))
    out = torch.stack([m(i) for i in range(n)])
    (out * torch.from_numpy(GOLD["pose.G"])).sum().backward()
    np.testing.assert_allclose(out.detach().numpy(), GOLD[f"pose.{tag}.out"], rtol=0, atol=1e-6)
    for k, p in (("r", m.r), ("t", m.t)):
        assert int(p.requires_grad) == int(GOLD[f"pose.{tag}.requires_grad.{k}"])
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        np.testing.assert_allclose(g.numpy(), GOLD[f"pose.{tag}.g.{k}"], rtol=0, atol=2e-6)
    assert sorted(m.state_dict().keys()) == list(GOLD[f"pose.{tag}.keys"])
    assert m.get_t() is m.t


@pytest.mark.parametrize("fix", [True, False])
def test_learn_distortion_floor_and_last_camera(fix):
    import model as mdl
    n = 5
    m = mdl.Learn_Distortion(n, True, True, {"distortion": {"fix_scaleN": fix}})
    with torch.no_grad():
        m.global_scales.copy_(torch.tensor(SCALES).view(n, 1)); m.global_shifts.copy_(torch.tensor(SHIFTS).view(n, 1))
    vals, total = [], 0.0
    for i in range(n):
        s, h = m(i)
        vals.append([float(s), float(h)])
        total = total + (3.0 * s + 2.0 * h).sum()
    total.backward()
    np.testing.assert_allclose(np.array(vals), GOLD[f"dist.fix{int(fix)}.out"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(m.global_scales.grad.numpy(), GOLD[f"dist.fix{int(fix)}.g.scales"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(m.global_shifts.grad.numpy(), GOLD[f"dist.fix{int(fix)}.g.shifts"], rtol=0, atol=1e-7)


@pytest.mark.parametrize("i", range(len(FOCAL_CASES)))
def test_learn_focal_variants(i):
    import model as mdl
    fx_only, order, init = FOCAL_CASES[i]
    if fx_only and isinstance(init, list):
        # the reference turns the LIST into a 2-vector parameter and returns a (2,2) "focal", which its own callers cannot use
        # (training.py:248-252 concatenates fxfy[0:1] with scalars); here the first entry initialises the single focal
        assert GOLD[f"focal.{i}.out"].shape == (2, 2)
        m = mdl.LearnFocal(True, fx_only, order=order, init_focal=init)
        assert m(0).shape == (2,) and float(m(0)[0]) == pytest.approx(init[0], rel=1e-6) and float(m(0)[1]) == pytest.approx(init[0], rel=1e-6)
        return
    m = mdl.LearnFocal(True, fx_only, order=order, init_focal=init)
    out = m(0)
    (out * torch.tensor([2.0, -3.0])).sum().backward()
    np.testing.assert_allclose(out.detach().numpy(), GOLD[f"focal.{i}.out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(m.fx.grad.numpy(), GOLD[f"focal.{i}.g.fx"], rtol=0, atol=1e-6)
    if not fx_only:
        np.testing.assert_allclose(m.fy.grad.numpy(), GOLD[f"focal.{i}.g.fy"], rtol=0, atol=1e-6)
    assert sorted(m.state_dict().keys()) == list(GOLD[f"focal.{i}.keys"])
