"""Golden vectors for the novel-view camera paths vis/render.py builds from the learned poses (reference model/common.py:333-404,
511-615: interp_poses, interp_poses_bspline, scipy_bspline, get_poses_at_times, generate_spiral_nerf, create_spheric_poses).
Runs the REFERENCE functions on seeded poses and freezes inputs + outputs in tests/golden/trajectories.npz.
Authoring container only:  python oracle/gen_golden_traj.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from gen_golden_poses import trajectory  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "trajectories.npz")


def main():
    gg.import_reference()
    from model import common as rc
    blob = {}
    for name, n, seed in (("short", 5, 11), ("long", 23, 12)):
        c2ws, _ = trajectory(n, seed, 0.0)
        i_train = np.array([i for i in range(n + n // 8 + 1) if i % 8 != 4][:n])       # training indices with held-out gaps
        hwf = np.tile(np.array([[48.0], [64.0], [57.6]], np.float32), (n, 1, 1))
        blob[f"{name}.c2ws"] = c2ws.numpy()
        blob[f"{name}.i_train"] = i_train
        blob[f"{name}.interp"] = rc.interp_poses(c2ws, 17).numpy()
        for deg in (2, 100):
            blob[f"{name}.bspline{deg}"] = rc.interp_poses_bspline(c2ws, 31, i_train, deg).numpy()
        blob[f"{name}.spiral"] = rc.generate_spiral_nerf(c2ws, np.array([2., 4.]), 12, hwf).numpy()
        times = i_train.astype(np.float64)
        q = np.linspace(times[0] + 0.25, times[-1] - 0.25, 9)
        blob[f"{name}.at_times_q"] = q
        blob[f"{name}.at_times"] = rc.get_poses_at_times(c2ws, times, q).numpy()
        blob[f"{name}.periodic"] = rc.scipy_bspline(c2ws[:, :3, 3].numpy(), n=20, degree=3, periodic=True)
    blob["circle"] = rc.create_spheric_poses(3.0, 0.4, n_poses=10)
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
