"""Golden vectors for the trajectory metrics ("PSNR + ATE/RPE via the reference's own
utils_poses/comp_ate.py:33-73 / align_traj.py:26-69").  Runs the REFERENCE functions (and the ATE/ solver they call) on seeded
trajectories and freezes inputs + outputs in tests/golden/pose_metrics.npz.  Authoring container only:
    python oracle/gen_golden_poses.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("NNR_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "pose_metrics.npz")


def trajectory(n, seed, noise):
    """A smooth camera path + a sim(3)-moved, perturbed copy of it: (gt, est) as (n,4,4) float32 tensors."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    s = np.linspace(0, 1, n)
    pos = np.stack([np.sin(2.2 * s), 0.3 * np.cos(3 * s), 0.5 * s], -1) + 0.01 * rng.standard_normal((n, 3))
    rot = Rotation.from_rotvec(np.stack([0.2 * s, 0.4 * np.sin(s), 0.1 * np.cos(2 * s)], -1)).as_matrix()
    gt = np.tile(np.eye(4), (n, 1, 1))
    gt[:, :3, :3], gt[:, :3, 3] = rot, pos
    A = Rotation.from_rotvec(rng.standard_normal(3)).as_matrix()
    scale, shift = 0.4 + rng.random(), rng.standard_normal(3)
    est = np.tile(np.eye(4), (n, 1, 1))
    wobble = Rotation.from_rotvec(noise * rng.standard_normal((n, 3))).as_matrix()
    est[:, :3, :3] = A @ rot @ wobble
    est[:, :3, 3] = scale * pos @ A.T + shift + noise * rng.standard_normal((n, 3))
    return torch.from_numpy(gt).float(), torch.from_numpy(est).float()


def main():
    sys.path.insert(0, REF)
    from utils_poses.align_traj import align_ate_c2b_use_a2b, align_scale_c2b_use_a2b, pts_dist_max
    from utils_poses.comp_ate import compute_ATE, compute_rpe
    blob = {}
    for name, n, seed, noise in (("clean", 12, 1, 0.0), ("noisy", 30, 2, 0.02), ("short", 3, 3, 0.05), ("rough", 17, 4, 0.3)):
        gt, est = trajectory(n, seed, noise)
        aligned = align_ate_c2b_use_a2b(est, gt)
        other = align_ate_c2b_use_a2b(est, gt, est[: n // 2 + 1].clone())
        ate = compute_ATE(gt.numpy(), aligned.numpy())
        rpe_t, rpe_r = compute_rpe(gt.numpy(), aligned.numpy())
        scaled, sc = align_scale_c2b_use_a2b(est.clone(), gt.clone())
        blob.update({f"{name}.gt": gt.numpy(), f"{name}.est": est.numpy(), f"{name}.aligned": aligned.numpy(),
                     f"{name}.aligned_half": other.numpy(), f"{name}.ate": np.float64(ate), f"{name}.rpe_t": np.float64(rpe_t),
                     f"{name}.rpe_r": np.float64(rpe_r), f"{name}.scaled": scaled.numpy(), f"{name}.scale": np.float64(float(sc)),
                     f"{name}.extent": np.float64(float(pts_dist_max(gt[:, :3, 3])))})
        print(f"{name:6s} n={n:3d}  ATE {ate:.6f}  RPE_t {rpe_t:.6f}  RPE_r {rpe_r:.6f}  scale {float(sc):.6f}")
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
