"""Golden vectors for the point-cloud loss: runs the REFERENCE Loss.get_pc_loss (model/losses.py:114-148)
on seeded clouds, asserts the oracle restatement reproduces it, and freezes inputs / matches / loss / gradients in
tests/golden/pc_loss.npz.  Run in the authoring container only (needs /root/reference):  python oracle/gen_golden_pc.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402  (import recipe for the reference + the oracle module)
import nerf_oracle as orc  # noqa: E402


def clouds(seed, S, D):
    """Two views of a noisy surface (like two back-projected depth maps), plus exact duplicates so ties are exercised."""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(max(S, D), 3, generator=g) * torch.tensor([4.0, 3.0, 1.0]) + torch.tensor([-2.0, -1.5, 2.0])
    x = base[:S] + 0.02 * torch.randn(S, 3, generator=g)
    y = base[torch.randperm(max(S, D), generator=g)[:D]] + 0.02 * torch.randn(D, 3, generator=g)
    y[5] = y[3]          # duplicated destination points: the FIRST index must win
    y[D - 1] = y[3]
    x[7] = y[11]         # a zero distance (gradient must be 0 there, like torch's norm backward)
    return x.unsqueeze(0), y.unsqueeze(0)


def main():
    ref = gg.import_reference()
    import model.losses as ref_losses
    cfg = gg.base_cfg(128)["training"]
    loss_mod = ref_losses.Loss(cfg)
    blob = {}
    for name, (S, D) in {"a": (700, 650), "b": (1300, 2100)}.items():
        x, y = clouds(11 if name == "a" else 12, S, D)
        xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        l_ref = loss_mod.get_pc_loss(xr, yr)
        l_ref.backward()
        xo, yo = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        l_orc = orc.pc_loss(xo, yo)
        l_orc.backward()
        assert torch.equal(l_ref, l_orc), (l_ref, l_orc)
        assert torch.equal(xr.grad, xo.grad) and torch.equal(yr.grad, yo.grad)
        xt, yt = x[0].permute(1, 0), y[0].permute(1, 0)
        i_xy = loss_mod.comp_closest_pts_idx_with_split(xt, yt)
        i_yx = loss_mod.comp_closest_pts_idx_with_split(yt, xt)
        assert torch.equal(i_xy, orc.closest_idx(xt, yt)) and torch.equal(i_yx, orc.closest_idx(yt, xt))
        for k, v in (("x", x[0]), ("y", y[0]), ("idx_xy", i_xy), ("idx_yx", i_yx), ("loss", l_ref.detach()),
                     ("gx", xr.grad[0]), ("gy", yr.grad[0])):
            blob[f"{name}.{k}"] = v.numpy()
        print(f"pc case {name}: S={S} D={D} loss={float(l_ref):.6f}; oracle == reference (bit-exact)")
    np.savez_compressed(os.path.join(gg.OUT, "pc_loss.npz"), **blob)
    print("wrote tests/golden/pc_loss.npz")


if __name__ == "__main__":
    main()
