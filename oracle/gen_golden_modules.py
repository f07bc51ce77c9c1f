"""Golden vectors for the three small parameter modules around the render path (reference model/poses.py:6-33,
model/distortions.py:4-26, model/intrinsics.py:5-70) in the configurations the render goldens do not visit: poses composed with
initial camera-to-world matrices, rotation / translation frozen, the scale floor and the free last scale of the distortion, all
(fx_only, order, init) variants of the focal.  REFERENCE classes on CPU; forward values and gradients of a fixed linear functional
in tests/golden/modules.npz.  Authoring container only:  python oracle/gen_golden_modules.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from gen_golden_poses import trajectory  # noqa: E402

FOCAL_CASES = [(fx_only, order, init) for fx_only in (True, False) for order in (1, 2) for init in (None, 1.7, [1.7, 2.3])]
SCALES = [1.2, 0.005, -0.3, 0.7, 0.9]
SHIFTS = [0.1, -0.2, 0.05, 0.0, 0.3]


def main():
    ref = gg.import_reference()
    blob = {}
    g = torch.Generator().manual_seed(9)
    n = 5
    init, _ = trajectory(n, 5, 0.0)
    r0, t0 = 0.1 * torch.randn(n, 3, generator=g), 0.2 * torch.randn(n, 3, generator=g)
    G = torch.randn(n, 4, 4, generator=g)
    blob.update({"pose.init": init.numpy(), "pose.r": r0.numpy(), "pose.t": t0.numpy(), "pose.G": G.numpy()})
    for tag, kw in (("composed", dict(learn_R=True, learn_t=True, init_c2w=init.clone())), ("frozen_R", dict(learn_R=False, learn_t=True)),
                    ("frozen_t", dict(learn_R=True, learn_t=False))):
        m = ref.LearnPose(n, kw["learn_R"], kw["learn_t"], {}, init_c2w=kw.get("init_c2w"))
        with torch.no_grad():
            m.r.copy_(r0); m.t.copy_(t0)
        out = torch.stack([m(i) for i in range(n)])
        (out * G).sum().backward()
        blob[f"pose.{tag}.out"] = out.detach().numpy()
        for k, p in (("r", m.r), ("t", m.t)):
            blob[f"pose.{tag}.requires_grad.{k}"] = int(p.requires_grad)
            blob[f"pose.{tag}.g.{k}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
        blob[f"pose.{tag}.keys"] = np.array(sorted(m.state_dict().keys()))
    for fix in (True, False):
        m = ref.Learn_Distortion(n, True, True, {"distortion": {"fix_scaleN": fix}})
        with torch.no_grad():
            m.global_scales.copy_(torch.tensor(SCALES).view(n, 1)); m.global_shifts.copy_(torch.tensor(SHIFTS).view(n, 1))
        vals, total = [], 0.0
        for i in range(n):
            s, h = m(i)
            vals.append([float(s), float(h)])
            total = total + (3.0 * s + 2.0 * h).sum()
        total.backward()
        blob[f"dist.fix{int(fix)}.out"] = np.array(vals)
        blob[f"dist.fix{int(fix)}.g.scales"] = m.global_scales.grad.numpy()
        blob[f"dist.fix{int(fix)}.g.shifts"] = m.global_shifts.grad.numpy()
    for i, (fx_only, order, init_f) in enumerate(FOCAL_CASES):
        m = ref.LearnFocal(True, fx_only, order=order, init_focal=init_f)
        out = m(0)
        (out * torch.tensor([2.0, -3.0])).sum().backward()
        blob[f"focal.{i}.out"] = out.detach().numpy()
        blob[f"focal.{i}.g.fx"] = m.fx.grad.numpy()
        if not fx_only:
            blob[f"focal.{i}.g.fy"] = m.fy.grad.numpy()
        blob[f"focal.{i}.keys"] = np.array(sorted(m.state_dict().keys()))
    out = os.path.join(gg.OUT, "modules.npz")
    np.savez_compressed(out, **blob)
    print("wrote", out, os.path.getsize(out), "bytes;", len(FOCAL_CASES), "focal variants")


if __name__ == "__main__":
    main()
