"""Novel-view camera paths for vis/render.py (reference model/common.py:333-404,511-615): interpolation of the learned poses
(slerp + linear or B-spline centres), the NeRF-style spiral around the average camera, and the circular path.  Host-side scipy /
numpy on a few hundred poses; the frames along the path are rendered by the forward-only fused kernel (model/imaging.py)."""
import numpy as np
import scipy.interpolate as si
import torch
from scipy.spatial.transform import Rotation, Slerp


def _pad44(p34):
    out = torch.cat([p34, torch.zeros_like(p34[:, 0:1])], dim=1)
    out[:, 3, 3] = 1.0
    return out


def normalize(v):
    return v / np.linalg.norm(v)


def viewmatrix(z, up, pos):
    """3x4 [x|y|z|pos]: z along `z`, x = up x z, y = z x x."""
    z = normalize(z)
    x = normalize(np.cross(up, z))
    return np.stack([x, normalize(np.cross(z, x)), z, pos], 1)


def poses_avg(poses):
    """Average camera of (n,3,5) poses [R|t|hwf]: mean centre, summed view / up axes, the first pose's hwf column."""
    frame = viewmatrix(poses[:, :3, 2].sum(0), poses[:, :3, 1].sum(0), poses[:, :3, 3].mean(0))
    return np.concatenate([frame, poses[0, :3, -1:]], 1)


def _slerp_matrices(rotations, key_times, query_times):
    """float32 (m,3,3) tensor of the key rotations spherically interpolated at the query times."""
    rotations = rotations.numpy() if torch.is_tensor(rotations) else np.asarray(rotations)
    return torch.from_numpy(Slerp(key_times, Rotation.from_matrix(rotations))(query_times).as_matrix().astype(np.float32))


def interp_poses(c2ws, N_views):
    """(n,4,4) -> (N_views,4,4): rotations slerped over uniform key times, centres resampled linearly with torch's
    half-pixel-centre 'linear' interpolate (so the end points are held, not hit exactly)  (common.py:511-522)."""
    n = c2ws.shape[0]
    rots = _slerp_matrices(c2ws[:, :3, :3], np.linspace(0, 1, n), np.linspace(0, 1, N_views))
    centres = torch.nn.functional.interpolate(c2ws[:, :3, 3:].permute(2, 1, 0), size=N_views, mode='linear').permute(2, 1, 0)
    return _pad44(torch.cat([rots, centres], dim=2))


def scipy_bspline(cv, n=100, degree=3, periodic=False):
    """n samples of the clamped (or periodic) B-spline through control vertices cv (common.py:563-589)."""
    cv = np.asarray(cv)
    count = cv.shape[0]
    if periodic:
        kv = np.arange(-degree, count + degree + 1)
        factor, fraction = divmod(count + degree + 1, count)
        cv = np.roll(np.concatenate((cv,) * factor + (cv[:fraction],)), -1, axis=0)
        degree = np.clip(degree, 1, degree)
    else:
        degree = np.clip(degree, 1, count - 1)
        kv = np.clip(np.arange(count + degree + 1) - degree, 0, count - degree)
    top = count - (degree * (1 - periodic))
    return si.BSpline(kv, cv, degree)(np.linspace(0, top, n))


def interp_poses_bspline(c2ws, N_novel_imgs, input_times, degree):
    """Centres along a B-spline with the learned centres as control points, rotations slerped over `input_times`
    (common.py:523-531)."""
    centres = torch.tensor(scipy_bspline(c2ws[:, :3, 3], n=N_novel_imgs, degree=degree, periodic=False).astype(np.float32))
    times = np.linspace(input_times[0], input_times[-1], N_novel_imgs)
    return _pad44(torch.cat([_slerp_matrices(c2ws[:, :3, :3], input_times, times), centres.unsqueeze(2)], dim=2))


def interp_t(trans, input_times, target_times):
    """Centres at the target times from the bracketing key frames, with the reference's weights (common.py:544-559): the key
    BELOW the target is weighted by the distance to itself, so the result runs from the upper key towards the lower one; a target
    that coincides with a key time divides 0 by 0.  Unused by the scripts; kept call-compatible."""
    out = []
    for t in target_times:
        diff = t - input_times
        lo = np.argmin(np.where(diff < 0, 1000, diff))
        hi = np.argmin(-np.where(diff > 0, -1000, diff))
        span = input_times[hi] - input_times[lo]
        out.append((t - input_times[lo]) / span * trans[lo] + (input_times[hi] - t) / span * trans[hi])
    return torch.stack(out, axis=0)


def get_poses_at_times(c2ws, input_times, target_times):
    rots = _slerp_matrices(c2ws[:, :3, :3], input_times, target_times)
    return _pad44(torch.cat([rots, interp_t(c2ws[:, :3, 3:], input_times, target_times)], dim=2))


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    """N 3x5 poses on a spiral of radii `rads` around the 3x5 camera c2w, all looking at the point `focal` ahead of it
    (common.py:381-392; the 0.2 / 0.2 / 0.1 shape factors are the reference's)."""
    rads = np.array(list(rads) + [1.])
    hwf = c2w[:, 4:5]
    look = np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.]))
    path = []
    for th in np.linspace(0., 2. * np.pi * rots, N + 1)[:-1]:
        eye = np.dot(c2w[:3, :4], np.array([0.2 * np.cos(th), -0.2 * np.sin(th), -np.sin(th * zrate) * 0.1, 1.]) * rads)
        path.append(np.concatenate([viewmatrix(eye - look, up, eye), hwf], 1))
    return path


def generate_spiral_nerf(learned_poses, bds, N_novel_views, hwf):
    """Two-turn spiral sized by the 90th percentile of the learned centres, focus depth from the bounds (common.py:591-615).
    -> (N,3,4) float32 tensor."""
    poses = np.concatenate((learned_poses[:, :3, :4].detach().cpu().numpy(), hwf[:len(learned_poses)]), axis=-1)
    c2w = poses_avg(poses)
    print('recentered', c2w.shape)
    up = normalize(poses[:, :3, 1].sum(0))
    close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
    dt = .75
    focal = 1. / ((1. - dt) / close_depth + dt / inf_depth)
    rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
    path = render_path_spiral(c2w, up, rads, focal, close_depth * .2, zrate=.5, rots=2, N=N_novel_views)
    return torch.tensor(np.stack(path).astype(np.float32))[:, :3, :4]


def create_spheric_poses(radius, mean_h, n_poses=120):
    """(n_poses,3,4) cameras on a circle around the z axis, pitched by -15 degrees (common.py:333-369)."""
    phi = -np.pi / 12
    tilt = np.array([[1, 0, 0], [0, np.cos(phi), -np.sin(phi)], [0, np.sin(phi), np.cos(phi)]])
    lift = np.array([[1, 0, 0, 0], [0, 1, 0, 2 * mean_h], [0, 0, 1, -radius]])
    axes = np.array([[-1, 0, 0], [0, 0, 1], [0, 1, 0]])
    out = []
    for th in np.linspace(0, 2 * np.pi, n_poses + 1)[:-1]:
        pan = np.array([[np.cos(th), 0, -np.sin(th)], [0, 1, 0], [np.sin(th), 0, np.cos(th)]])
        out.append(axes @ (pan @ tilt @ lift))
    return np.stack(out, 0)
