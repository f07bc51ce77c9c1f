"""CPU: the novel-view camera paths (model/trajectories.py, re-exported by model.common as vis/render.py imports them) against
golden vectors produced by the reference's functions (oracle/gen_golden_traj.py -> tests/golden/trajectories.npz)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))

from model import common  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "trajectories.npz"))


@pytest.mark.parametrize("name", ["short", "long"])
def test_paths_match_the_reference(name):
    g = lambda k: GOLD[f"{name}.{k}"]
    c2ws, i_train = torch.from_numpy(g("c2ws")), g("i_train")
    n = len(c2ws)
    tol = dict(rtol=0, atol=1e-6)
    np.testing.assert_allclose(common.interp_poses(c2ws, 17).numpy(), g("interp"), **tol)
    for deg in (2, 100):
        np.testing.assert_allclose(common.interp_poses_bspline(c2ws, 31, i_train, deg).numpy(), g(f"bspline{deg}"), **tol)
    hwf = np.tile(np.array([[48.0], [64.0], [57.6]], np.float32), (n, 1, 1))
    np.testing.assert_allclose(common.generate_spiral_nerf(c2ws, np.array([2., 4.]), 12, hwf).numpy(), g("spiral"), **tol)
    got = common.get_poses_at_times(c2ws, i_train.astype(np.float64), g("at_times_q")).numpy()
    np.testing.assert_allclose(got, g("at_times"), **tol)
    np.testing.assert_allclose(common.scipy_bspline(c2ws[:, :3, 3].numpy(), n=20, degree=3, periodic=True), g("periodic"), **tol)


def test_circle_and_shapes():
    np.testing.assert_allclose(common.create_spheric_poses(3.0, 0.4, n_poses=10), GOLD["circle"], rtol=0, atol=1e-12)
    c2ws = torch.from_numpy(GOLD["long.c2ws"])
    out = common.interp_poses_bspline(c2ws, 40, GOLD["long.i_train"], 100)
    assert out.shape == (40, 4, 4) and out.dtype == torch.float32
    R = out[:, :3, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3).expand(40, 3, 3), atol=1e-5)      # slerp keeps rotations proper
    assert torch.equal(out[:, 3], torch.tensor([0.0, 0, 0, 1]).expand(40, 4))
    # a clamped B-spline starts and ends on the first / last control point
    assert torch.allclose(out[0, :3, 3], c2ws[0, :3, 3], atol=1e-6) and torch.allclose(out[-1, :3, 3], c2ws[-1, :3, 3], atol=1e-6)
