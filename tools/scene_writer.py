#!/usr/bin/env python
"""Synthetic scene writer: a forward-facing analytic scene, ray-cast in numpy and written in the reference's on-disk layout, so
that training / evaluation can run offline (SURVEY.md section 8 row f4, configs 4-5 of 8(d)).

    python tools/scene_writer.py OUT_DIR [--scene NAME] [--frames 16] [--size 120 160] [--factor 2] [--seed 0]

Layout written (what reference dataloading/common.py:59-141,289-314 and dataloading/dataset.py:53-150 read):

    OUT_DIR/NAME/images/000.png ...          8-bit RGB frames, sorted by name
    OUT_DIR/NAME/images_F/000.png ...        the same views at 1/F resolution (only with --factor F; the reference would shell out
                                             to ImageMagick for these when `resize_factor: F` is set and the folder is missing)
    OUT_DIR/NAME/poses_bounds.npy            (n, 17) float64, LLFF: 3x5 [down|right|back|centre|(H, W, focal)] row-major + near, far
    OUT_DIR/NAME/dpt/depth_000.npz ...       key 'pred', (1, h, w) float32: monocular depth = true z-depth under a per-frame
                                             affine distortion  (pred = (z - shift_i) / scale_i), as DPT's output is
    OUT_DIR/NAME/depth/000.png               16-bit z-depth in millimetres (`with_depth: True`)
    OUT_DIR/NAME/gt_poses.npz                'poses' (n,4,4) camera-to-world, OpenCV axes (`customized_poses: True`)
    OUT_DIR/NAME/intrinsics.npz              'K' (3,3) pixel intrinsics of images/ (`customized_focal: True`)
    OUT_DIR/NAME/scene.json                  what was drawn (seed, cameras, distortion) -- for tests, not read by any loader

Cameras follow the OpenGL convention of the render path (x right, y up, looking down -z; reference dataset.py:101-104 builds
K = diag(2f/w, -2f/h, -1, 1)).  Pixel (x, y) looks along ((2x/(w-1) - 1) w/2f, -(2y/(h-1) - 1) h/2f, -1): the very mapping the
render path applies (model/common.py:13-40,186-237), so the frames are exactly consistent with the rays that will be traced.
"""
import argparse
import json
import os

import numpy as np
from PIL import Image


def look_at(eye, target, up=(0.0, 1.0, 0.0), roll=0.0):
    """Camera-to-world rotation with columns [right, up, back]."""
    back = eye - target
    back = back / np.linalg.norm(back)
    right = np.cross(np.asarray(up, float), back)
    right = right / np.linalg.norm(right)
    true_up = np.cross(back, right)
    c, s = np.cos(roll), np.sin(roll)
    return np.stack([c * right + s * true_up, -s * right + c * true_up, back], axis=1)


def camera_path(n, seed):
    """A smooth hand-held sweep in front of the scene: (n,4,4) camera-to-world."""
    rng = np.random.default_rng(seed)
    s = np.linspace(0.0, 1.0, n)
    ph = rng.uniform(0, 2 * np.pi, 3)
    eye = np.stack([0.55 * np.sin(1.6 * np.pi * s + ph[0]) * (0.6 + 0.4 * s), 0.25 * np.sin(2.3 * np.pi * s + ph[1]),
                    0.15 * np.cos(1.1 * np.pi * s + ph[2])], axis=-1)
    target = np.stack([0.15 * np.sin(np.pi * s), 0.1 * np.cos(1.7 * np.pi * s), np.full(n, -3.2)], axis=-1)
    roll = 0.04 * np.sin(2 * np.pi * s + ph[0])
    c2w = np.tile(np.eye(4), (n, 1, 1))
    for i in range(n):
        c2w[i, :3, :3] = look_at(eye[i], target[i], roll=roll[i])
        c2w[i, :3, 3] = eye[i]
    return c2w


class Scene:
    """A textured back wall, a tilted floor and a few spheres; colours are smooth functions of the hit point."""

    def __init__(self, seed):
        rng = np.random.default_rng(seed + 1000)
        self.spheres = [(np.array([-0.9, -0.1, -3.0]), 0.55), (np.array([0.7, 0.35, -2.6]), 0.4),
                        (np.array([0.1, -0.45, -3.6]), 0.6), (np.array([1.3, -0.5, -3.9]), 0.5)]
        self.planes = [(np.array([0.0, 0.0, 1.0]), -4.6), (np.array([0.0, 1.0, 0.12]) / np.hypot(1.0, 0.12), -1.1)]  # n.p = d
        k = len(self.spheres) + len(self.planes)
        self.freq = rng.uniform(1.2, 3.2, (k, 3, 3)) * rng.choice([-1.0, 1.0], (k, 3, 3))
        self.phase = rng.uniform(0, 2 * np.pi, (k, 3))
        self.base = rng.uniform(0.25, 0.75, (k, 3))

    def colour(self, obj, p):
        return np.clip(self.base[obj] + 0.35 * np.sin(p @ self.freq[obj].T + self.phase[obj]), 0.0, 1.0)

    def trace(self, o, d):
        """o (3,), d (m,3) with d_z = -1  ->  rgb (m,3), t (m,): the hit is o + t d, so t is the z-depth."""
        t_best = np.full(d.shape[0], np.inf)
        obj = np.full(d.shape[0], -1)
        for j, (n, dist) in enumerate(self.planes):
            den = d @ n
            t = np.where(np.abs(den) > 1e-9, (dist - o @ n) / np.where(np.abs(den) > 1e-9, den, 1.0), np.inf)
            hit = (t > 1e-3) & (t < t_best)
            t_best, obj = np.where(hit, t, t_best), np.where(hit, j, obj)
        for j, (c, r) in enumerate(self.spheres):
            oc = o - c
            a, b, cc = (d * d).sum(-1), 2 * d @ oc, oc @ oc - r * r
            disc = b * b - 4 * a * cc
            t = np.where(disc > 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), np.inf)
            hit = (t > 1e-3) & (t < t_best)
            t_best, obj = np.where(hit, t, t_best), np.where(hit, len(self.planes) + j, obj)
        assert (obj >= 0).all(), "a ray left the scene: keep the cameras in front of the back wall"
        p = o + t_best[:, None] * d
        rgb = np.zeros_like(d)
        for j in range(len(self.planes) + len(self.spheres)):
            m = obj == j
            if m.any():
                rgb[m] = self.colour(j, p[m])
        return rgb, t_best


def pixel_dirs(h, w, f):
    """Camera-space directions of all pixels, row-major, for images of (h, w) with focal f in pixels."""
    xs = (2.0 * np.arange(w) / (w - 1) - 1.0) * (w / (2.0 * f))
    ys = -(2.0 * np.arange(h) / (h - 1) - 1.0) * (h / (2.0 * f))
    gx, gy = np.meshgrid(xs, ys)
    return np.stack([gx.ravel(), gy.ravel(), -np.ones(h * w)], axis=-1)


def render_view(scene, c2w, h, w, f):
    d = pixel_dirs(h, w, f) @ c2w[:3, :3].T           # world directions, not normalised: t stays the z-depth
    rgb, z = scene.trace(c2w[:3, 3], d)
    return rgb.reshape(h, w, 3), z.reshape(h, w)


def llff_rows(c2w, h, w, f, near, far):
    """(n,17): the LLFF pose block stores the rotation columns as [down, right, back] (dataset.py:56 undoes exactly this)."""
    n = c2w.shape[0]
    blk = np.zeros((n, 3, 5))
    blk[:, :, 0] = -c2w[:, :3, 1]
    blk[:, :, 1] = c2w[:, :3, 0]
    blk[:, :, 2] = c2w[:, :3, 2]
    blk[:, :, 3] = c2w[:, :3, 3]
    blk[:, :, 4] = np.array([h, w, f])
    return np.concatenate([blk.reshape(n, 15), near[:, None], far[:, None]], axis=1)


def write_scene(out_dir, scene='synthetic', frames=16, size=(120, 160), focal_ratio=0.9, factor=None, seed=0, depth_size=None,
                distort=True):
    """Write the scene; -> dict of what was drawn (also saved as scene.json).  `size` = (h, w) of images/; `depth_size` = (h, w)
    of the monocular depth maps (default: the resolution training will see, i.e. size / factor)."""
    h, w = size
    f = focal_ratio * w
    root = os.path.join(out_dir, scene)
    for sub in ('images', 'dpt', 'depth') + ((f'images_{factor}',) if factor else ()):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    world = Scene(seed)
    c2w = camera_path(frames, seed)
    rng = np.random.default_rng(seed + 2000)
    scales = rng.uniform(0.8, 1.25, frames) if distort else np.ones(frames)
    shifts = rng.uniform(-0.15, 0.15, frames) if distort else np.zeros(frames)
    hs, ws = (h // factor, w // factor) if factor else (h, w)
    hd, wd = depth_size or (hs, ws)
    near, far = np.zeros(frames), np.zeros(frames)
    for i in range(frames):
        name = f'{i:03d}'
        rgb, z = render_view(world, c2w[i], h, w, f)
        Image.fromarray(np.round(rgb * 255).astype(np.uint8)).save(os.path.join(root, 'images', name + '.png'))
        Image.fromarray(np.round(z * 1000).astype(np.uint16)).save(os.path.join(root, 'depth', name + '.png'))
        near[i], far[i] = 0.9 * z.min(), 1.1 * z.max()
        if factor:   # the same view at 1/F: same field of view, focal f/F (common.py:131 divides the stored focal by F)
            small, _ = render_view(world, c2w[i], hs, ws, f * ws / w)
            Image.fromarray(np.round(small * 255).astype(np.uint8)).save(os.path.join(root, f'images_{factor}', name + '.png'))
        _, zd = render_view(world, c2w[i], hd, wd, f * wd / w)
        np.savez(os.path.join(root, 'dpt', f'depth_{name}.npz'), pred=((zd - shifts[i]) / scales[i]).astype(np.float32)[None])
    np.save(os.path.join(root, 'poses_bounds.npy'), llff_rows(c2w, h, w, f, near, far))
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    np.savez(os.path.join(root, 'gt_poses.npz'), poses=c2w @ flip)
    np.savez(os.path.join(root, 'intrinsics.npz'), K=np.array([[f, 0, w / 2.0], [0, f, h / 2.0], [0, 0, 1.0]]))
    meta = {'scene': scene, 'frames': frames, 'size': [h, w], 'focal': f, 'factor': factor, 'seed': seed, 'depth_size': [hd, wd],
            'c2w': c2w.tolist(), 'scales': scales.tolist(), 'shifts': shifts.tolist(), 'near': near.tolist(), 'far': far.tolist()}
    with open(os.path.join(root, 'scene.json'), 'w') as fh:
        json.dump(meta, fh)
    return meta


def main():
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('out_dir')
    ap.add_argument('--scene', default='synthetic')
    ap.add_argument('--frames', type=int, default=16)
    ap.add_argument('--size', type=int, nargs=2, default=(120, 160), metavar=('H', 'W'))
    ap.add_argument('--factor', type=int, default=None)
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    m = write_scene(a.out_dir, a.scene, a.frames, tuple(a.size), factor=a.factor, seed=a.seed)
    print(f"wrote {m['frames']} frames {m['size']} focal {m['focal']:.1f} to {os.path.join(a.out_dir, a.scene)}")


if __name__ == '__main__':
    main()
