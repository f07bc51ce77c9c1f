"""sim(3) alignment of an estimated trajectory to a reference one (reference utils_poses/align_traj.py:9-97; the solver it
calls is ATE/align_utils.py:99-108 -> ATE/align_trajectory.py:30-80, Umeyama 1991 on the camera centres, all frames)."""
import numpy as np
import torch

from utils_poses.lie_group_helper import convert3x4_4x4


def umeyama_sim3(model, data):
    """Least-squares s, R, t with  model ~ s * R @ data + t  for two (n,3) point sets (Umeyama, PAMI 13(4), 1991).
    The reflection guard follows the reference's: the sign of det(U)det(V) (align_trajectory.py:59-61)."""
    model, data = np.asarray(model), np.asarray(data)
    mu_m, mu_d = model.mean(0), data.mean(0)
    m0, d0 = model - mu_m, data - mu_d
    n = model.shape[0]
    cov = m0.T @ d0 / n
    var_d = (d0 * d0).sum() / n
    U, sv, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt.T) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    with np.errstate(divide='ignore', invalid='ignore'):      # coincident centres (identity initialisation): s is nan, as in the reference
        s = np.trace(np.diag(sv) @ S) / var_d
    return s, R, mu_m - s * R @ mu_d


def pts_dist_max(pts):
    """Largest distance of any point to the FIRST point (align_traj.py:9-25: the reference indexes row 0 of the pairwise
    table, so this is not the diameter of the set; kept as is because the scale alignment is defined through it)."""
    d = pts - pts[0:1]
    return d.norm(dim=1).max() if torch.is_tensor(pts) else np.linalg.norm(d, axis=1).max()


def align_ate_c2b_use_a2b(traj_a, traj_b, traj_c=None):
    """Apply the sim(3) that maps trajectory a onto b to trajectory c (default: a itself).  (N,3|4,4) tensors in, (N,4,4) out
    on a's device.  Rotations are aligned by R only, centres by s*R*t + t0 (align_traj.py:28-73)."""
    device = traj_a.device
    a = traj_a.float().cpu().numpy()
    b = traj_b.float().cpu().numpy()
    c = a.copy() if traj_c is None else traj_c.float().cpu().numpy()
    s, R, t = umeyama_sim3(b[:, :3, 3], a[:, :3, 3])         # b ~ s R a + t
    R = R.astype(np.float32)
    rot = R[None] @ c[:, :3, :3]
    pos = float(s) * (R[None] @ c[:, :3, 3:4]) + t.astype(np.float32)[None, :, None]
    out = convert3x4_4x4(np.concatenate([rot, pos], axis=2))
    return torch.from_numpy(out).to(device)


def align_scale_c2b_use_a2b(traj_a, traj_b, traj_c=None):
    """Scale c's camera centres (in place, like the reference) by the a->b extent ratio; -> ((N,4,4), scale)
    (align_traj.py:77-97)."""
    if traj_c is None:
        traj_c = traj_a.clone()
    scale = pts_dist_max(traj_b[:, :3, 3]) / pts_dist_max(traj_a[:, :3, 3])
    traj_c[:, :3, 3] *= scale
    if traj_c.shape[1] == 3:
        traj_c = convert3x4_4x4(traj_c)
    return traj_c, scale
