"""Which STAGE owns the tail of the fp64 yardstick?     python tools/fp64_bisect_seeds.py [D R N n_seeds]

tests/test_gpu_split3.py compares whole gradient tensors over 12 seeds: some (pose_r, layers0.6.weight) sit at a median of 2.3x the CPU oracle's
distance to fp64.  This runs tools/fp64_bisect.py's stage-by-stage comparison over the same number of seeds and prints, per stage, the MEDIAN over
the seeds of (HIP relative L2 against fp64) / (CPU fp32 relative L2 against fp64): the first stage where the column leaves 1 is where the excess is made.
Writes nothing; keep the output under profiles/."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import fp64_bisect as fb
from nnr import lib as L
import nerf_oracle as orc


def main():
    D, R, N, n_seeds = (int(x) for x in (sys.argv[1:5] + ["256", "256", "64", "12"][len(sys.argv) - 1:]))
    kinds = os.environ.get("NNR_BISECT_KINDS", L.fp32_products()).split(",")
    ratios = {k: {} for k in kinds}
    absolute = {}
    for i in range(n_seeds):
        seed = 333 + 100 * i
        g = torch.Generator().manual_seed(seed)
        params = orc.init_params(D, seed + 1)
        pts_o = torch.randn(R, 3, generator=g) * 0.1
        d = torch.randn(R, 3, generator=g)
        pts_d = d / d.norm(dim=-1, keepdim=True)
        view = -pts_d
        z = torch.linspace(0, 1, N)
        z = 0.01 * (1 - z) + 10.0 * z
        mid = 0.5 * (z[1:] + z[:-1])
        z_lo, z_hi = torch.cat([z[:1], mid]), torch.cat([mid, z[-1:]])
        jitter = torch.rand(R, N, generator=g)
        d_rgb = torch.randn(R, 3, generator=g) / R
        d_dist = torch.randn(R, generator=g) / R * 0.04
        args = (params, pts_o, pts_d, view, z_lo, z_hi, jitter, d_rgb, d_dist)
        ref = fb.trace(*args, torch.float64)
        cpu = fb.trace(*args, torch.float32)
        rl2 = lambda got, r: float((got.reshape(r.shape) - r).norm()) / max(float(r.norm()), 1e-300)
        # what a rigid motion of the camera sees of the three per-ray gradients (the chain rule of a rotation w and a translation t at the
        # identity: dL/dw = sum over rays of v x g for every vector v that turns with the camera, dL/dt = sum of g_o), formed in double from
        # each evaluation's own per-ray gradients: is an excess in the POSE gradients inherited (errors that do not cancel over the rays)?
        def rigid(tr):
            o, dd, vv = pts_o.double(), pts_d.double(), view.double()
            tr["rigid rotation"] = (torch.cross(o, tr["d pts_o"].reshape(R, 3), dim=1) + torch.cross(dd, tr["d pts_d"].reshape(R, 3), dim=1)
                                    + torch.cross(vv, tr["d view"].reshape(R, 3), dim=1)).sum(0)
            tr["rigid translation"] = tr["d pts_o"].reshape(R, 3).sum(0)
            tr["sum over rays d pts_d"] = tr["d pts_d"].reshape(R, 3).sum(0)
        rigid(ref)
        rigid(cpu)
        for kind in kinds:
            hip = fb.hip(*args, D, kind)
            rigid(hip)
            for k, r in ref.items():
                c, h = rl2(cpu[k], r), rl2(hip[k], r)
                ratios[kind].setdefault(k, []).append(h / max(c, 1e-12))
                if kind == kinds[0]:
                    absolute.setdefault(k, []).append((c, h))
        print("seed %d done" % seed, flush=True)
    print("\nD=%d, %d rays x %d samples, %d seeds: per stage, median over the seeds of HIP / CPU-fp32 (relative L2 against the fp64 trace); [min .. max]; "
          "median absolute rel-L2 CPU / HIP (%s)" % (D, R, N, n_seeds, kinds[0]))
    for k in ratios[kinds[0]]:
        line = "%-22s" % k
        for kind in kinds:
            v = np.asarray(ratios[kind][k])
            line += " | %-7s %5.2f [%5.2f .. %6.2f]" % (kind, np.median(v), v.min(), v.max())
        a = np.asarray(absolute[k])
        line += " | %.2e / %.2e" % (np.median(a[:, 0]), np.median(a[:, 1]))
        print(line)


if __name__ == "__main__":
    main()
