"""Point-cloud loss.  CPU: the oracle restatement against the golden vectors minted from the reference
(oracle/gen_golden_pc.py) and the product's CPU branch.  GPU (-m gpu): the HIP nearest-neighbour / error kernels through
the C ABI against the golden vectors (indices bit-exact, loss and gradients 1e-6) and against the oracle at the size the
trainer uses (96 x 168 points per cloud)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "pc_loss.npz"))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))


def _case(name):
    t = lambda k: torch.from_numpy(GOLD[f"{name}.{k}"])
    return t("x"), t("y"), t("idx_xy"), t("idx_yx"), t("loss"), t("gx"), t("gy")


@pytest.mark.parametrize("name", ["a", "b"])
def test_oracle_matches_reference_golden(name):
    import nerf_oracle as orc
    x, y, i_xy, i_yx, loss, gx, gy = _case(name)
    assert torch.equal(orc.closest_idx(x.t(), y.t()), i_xy) and torch.equal(orc.closest_idx(y.t(), x.t()), i_yx)
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    l = orc.pc_loss(xr[None], yr[None])
    l.backward()
    np.testing.assert_allclose(l.detach().numpy(), loss.numpy(), rtol=0, atol=1e-7)
    np.testing.assert_allclose(xr.grad.numpy(), gx.numpy(), rtol=0, atol=1e-7)
    np.testing.assert_allclose(yr.grad.numpy(), gy.numpy(), rtol=0, atol=1e-7)


def test_product_cpu_branch_matches_golden():
    from model.losses import Loss
    x, y, i_xy, _, loss, _, _ = _case("a")
    lm = Loss({'depth_loss_type': 'l1', 'match_method': 'dense', 'with_ssim': False})
    assert torch.equal(lm.comp_closest_pts_idx_with_split(x.t(), y.t()), i_xy)
    np.testing.assert_allclose(lm.get_pc_loss(x[None], y[None]).numpy(), loss.numpy(), rtol=0, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["a", "b"])
def test_hip_matches_reference_golden(name):
    from nnr import pointcloud
    x, y, i_xy, i_yx, loss, gx, gy = _case(name)
    xd, yd = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    idx, dist = pointcloud.nearest(xd, yd)
    assert torch.equal(idx.cpu(), i_xy)                      # first index on the duplicated points too
    assert torch.equal(pointcloud.nearest(yd, xd)[0].cpu(), i_yx)
    ref_dist = torch.linalg.norm(x - y[i_xy], dim=1)
    np.testing.assert_allclose(dist.cpu().numpy(), ref_dist.numpy(), rtol=0, atol=1e-7)
    l = pointcloud.point_point_error(xd, yd) + pointcloud.point_point_error(yd, xd)
    (3.0 * l).backward()                                     # non-unit upstream gradient
    np.testing.assert_allclose(l.item(), float(loss), rtol=0, atol=1e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy() / 3.0, gx.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(yd.grad.cpu().numpy() / 3.0, gy.numpy(), rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_hip_matches_oracle_at_trainer_size_and_through_the_loss_module():
    import nerf_oracle as orc
    from model.losses import Loss
    g = torch.Generator().manual_seed(5)
    S = 96 * 168                                             # (384/4) x (672/4): SURVEY 8 f1
    surf = torch.rand(S, 3, generator=g) * torch.tensor([6.0, 4.0, 2.0])
    x = surf + 0.01 * torch.randn(S, 3, generator=g)
    y = surf[torch.randperm(S, generator=g)] + 0.01 * torch.randn(S, 3, generator=g)
    xo, yo = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    lo = orc.pc_loss(xo[None], yo[None])
    lo.backward()
    lm = Loss({'depth_loss_type': 'l1', 'match_method': 'dense', 'with_ssim': False})
    xd, yd = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    ld = lm.get_pc_loss(xd[None], yd[None])                  # the product path: losses.py -> nnr.pointcloud -> libnnr.so
    ld.backward()
    from nnr import pointcloud
    assert torch.equal(pointcloud.nearest(xd, yd)[0].cpu(), orc.closest_idx(x.t(), y.t()))
    np.testing.assert_allclose(ld.item(), lo.item(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(xd.grad.cpu().numpy(), xo.grad.numpy(), rtol=0, atol=1e-8 + 1e-6 / S * 100)
    np.testing.assert_allclose(yd.grad.cpu().numpy(), yo.grad.numpy(), rtol=0, atol=1e-8 + 1e-6 / S * 100)


@pytest.mark.gpu
def test_rejects_cpu_tensors_and_bad_shapes():
    from nnr import pointcloud
    with pytest.raises(RuntimeError):
        pointcloud.nearest(torch.zeros(4, 3), torch.zeros(4, 3))
    with pytest.raises(ValueError):
        pointcloud.nearest(torch.zeros(4, 2, device="cuda"), torch.zeros(4, 3, device="cuda"))
