"""Absolute trajectory error and relative pose error (reference utils_poses/comp_ate.py:5-73).  Inputs are sequences of 4x4
camera-to-world matrices, `pred` already aligned to `gt` (align_traj.align_ate_c2b_use_a2b).  Vectorised over the frames;
float64 accumulation like the reference's numpy loops when given float64, float32 otherwise."""
import numpy as np


def rotation_error(pose_error):
    """Angle of the rotation part of one 4x4 relative pose: arccos((trace - 1) / 2), clipped into [-1, 1]."""
    cos = 0.5 * (pose_error[0, 0] + pose_error[1, 1] + pose_error[2, 2] - 1.0)
    return np.arccos(max(min(cos, 1.0), -1.0))


def translation_error(pose_error):
    """Length of the translation part of one 4x4 relative pose."""
    t = pose_error[:3, 3]
    return np.sqrt(t[0] ** 2 + t[1] ** 2 + t[2] ** 2)


def _relative(poses):
    poses = np.asarray(poses)
    return np.linalg.inv(poses[:-1]) @ poses[1:]          # frame i -> frame i+1, (n-1, 4, 4)


def compute_rpe(gt, pred):
    """-> (mean translation error, mean rotation error [rad]) of the frame-to-frame motions (comp_ate.py:33-51)."""
    err = np.linalg.inv(_relative(gt)) @ _relative(pred)
    trans = np.sqrt((err[:, :3, 3] ** 2).sum(-1))
    cos = np.clip(0.5 * (err[:, 0, 0] + err[:, 1, 1] + err[:, 2, 2] - 1.0), -1.0, 1.0)
    return np.mean(trans), np.mean(np.arccos(cos))


def compute_ATE(gt, pred):
    """RMSE of the camera-centre distances (comp_ate.py:53-73)."""
    gt, pred = np.asarray(gt), np.asarray(pred)
    d = np.sqrt(((gt[:len(pred), :3, 3] - pred[:, :3, 3]) ** 2).sum(-1))
    return np.sqrt(np.mean(d ** 2))
