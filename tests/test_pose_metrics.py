"""CPU: trajectory metrics (utils_poses/) against golden vectors produced by the reference's own functions
(oracle/gen_golden_poses.py -> tests/golden/pose_metrics.npz), plus the properties the metrics must have."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))

from utils_poses.align_traj import align_ate_c2b_use_a2b, align_scale_c2b_use_a2b, pts_dist_max, umeyama_sim3  # noqa: E402
from utils_poses.comp_ate import compute_ATE, compute_rpe, rotation_error, translation_error  # noqa: E402
from utils_poses.lie_group_helper import SO3_to_quat, quat_to_SO3  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "pose_metrics.npz"))
CASES = sorted({k.split(".")[0] for k in GOLD.files})


@pytest.mark.parametrize("name", CASES)
def test_alignment_and_errors_match_the_reference(name):
    g = lambda k: GOLD[f"{name}.{k}"]
    gt, est = torch.from_numpy(g("gt")), torch.from_numpy(g("est"))
    aligned = align_ate_c2b_use_a2b(est, gt)
    np.testing.assert_allclose(aligned.numpy(), g("aligned"), rtol=0, atol=2e-5)
    half = align_ate_c2b_use_a2b(est, gt, est[: len(est) // 2 + 1].clone())
    np.testing.assert_allclose(half.numpy(), g("aligned_half"), rtol=0, atol=2e-5)
    # the metrics on the reference's aligned trajectory: same numbers
    assert abs(compute_ATE(gt.numpy(), g("aligned")) - float(g("ate"))) <= 1e-6
    rpe_t, rpe_r = compute_rpe(gt.numpy(), g("aligned"))
    assert abs(rpe_t - float(g("rpe_t"))) <= 1e-6 and abs(rpe_r - float(g("rpe_r"))) <= 1e-6
    scaled, sc = align_scale_c2b_use_a2b(est.clone(), gt.clone())
    assert abs(float(sc) - float(g("scale"))) <= 1e-6
    np.testing.assert_allclose(scaled.numpy(), g("scaled"), rtol=0, atol=1e-6)
    assert abs(float(pts_dist_max(gt[:, :3, 3])) - float(g("extent"))) <= 1e-6


def test_sim3_moved_trajectory_aligns_back_exactly():
    rng = np.random.default_rng(0)
    n = 20
    gt = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    gt[:, :3, :3] = quat_to_SO3(rng.standard_normal((n, 4))).astype(np.float32)
    gt[:, :3, 3] = rng.standard_normal((n, 3)).astype(np.float32)
    A = quat_to_SO3(rng.standard_normal(4)).astype(np.float32)
    est = gt.copy()
    est[:, :3, :3] = A @ gt[:, :3, :3]
    est[:, :3, 3] = 2.5 * gt[:, :3, 3] @ A.T + np.array([1.0, -2.0, 0.5], np.float32)
    aligned = align_ate_c2b_use_a2b(torch.from_numpy(est), torch.from_numpy(gt)).numpy()
    assert compute_ATE(gt, aligned) <= 1e-5
    rpe_t, rpe_r = compute_rpe(gt, aligned)
    assert rpe_t <= 1e-5 and rpe_r <= 2e-3        # arccos near 1 amplifies fp32 rounding
    s, R, t = umeyama_sim3(gt[:, :3, 3], est[:, :3, 3])
    assert abs(s - 1 / 2.5) <= 1e-5 and np.allclose(R, A.T, atol=1e-5)


def test_error_primitives():
    e = np.eye(4)
    assert rotation_error(e) == 0.0 and translation_error(e) == 0.0
    th = 0.3
    e[:3, :3] = [[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]]
    e[:3, 3] = [3, 4, 12]
    assert abs(rotation_error(e) - th) <= 1e-12 and abs(translation_error(e) - 13.0) <= 1e-12
    q = SO3_to_quat(e[:3, :3])
    assert np.allclose(quat_to_SO3(q), e[:3, :3])


def test_reflection_guard_keeps_a_proper_rotation():
    # planar, mirrored point sets: the unconstrained optimum is a reflection
    a = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 2, 0]], np.float64)
    b = a * np.array([1, -1, 1])
    _, R, _ = umeyama_sim3(b, a)
    assert abs(np.linalg.det(R) - 1) <= 1e-9
