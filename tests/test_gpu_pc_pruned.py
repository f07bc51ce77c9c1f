"""GPU (-m gpu): the pruned nearest-neighbour search (csrc/nnr_pointcloud.hip: pc_nearest_pruned_kernel) returns EXACTLY what the
brute-force kernel returns -- indices and distances bit for bit -- on the clouds the trainer produces (two back-projected depth maps in
raster order), on unstructured random clouds (where the boxes prune nothing), with duplicated points, with points whose distances tie
in the last bit of the sqrt, with NaN / inf coordinates, and on ragged sizes.  The brute-force kernel is selected in a child process
(NNR_PC_PRUNED=1 is read once per process; the pruned search is opt-in: slower on white-noise depth)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(%r, "nope-nerf_amd"))
from nnr import pointcloud
z = np.load(sys.argv[1])
out = {}
for k in sorted(set(n.split(".")[0] for n in z.files)):
    idx, dist = pointcloud.nearest(torch.from_numpy(z[k + ".src"]).cuda(), torch.from_numpy(z[k + ".dst"]).cuda())
    out[k + ".idx"], out[k + ".dist"] = idx.cpu().numpy(), dist.cpu().numpy()
np.savez(sys.argv[2], **out)
""" % ROOT


def _grid_cloud(h, w, g, jitter=0.0):
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    d = 1 + 2 * torch.rand(h, w, generator=g)
    p = torch.stack([xs * d * 0.7, ys * d * 0.4, -d], -1).view(h * w, 3)
    return (p + jitter * torch.randn(p.shape, generator=g)).float()


def test_pruned_search_equals_the_brute_force_search_bit_for_bit(tmp_path):
    g = torch.Generator().manual_seed(3)
    cases = {
        "grid135x240": (_grid_cloud(135, 240, g), _grid_cloud(135, 240, g)),                  # the trainer's size at 540 x 960
        "grid_ragged": (_grid_cloud(37, 53, g), _grid_cloud(41, 47, g)),                       # S != D, neither a multiple of 64 / 512
        "random": (torch.rand(5000, 3, generator=g) * 4, torch.rand(7001, 3, generator=g) * 4),
        "tiny": (torch.rand(3, 3, generator=g), torch.rand(5, 3, generator=g)),
    }
    # duplicates: every destination point twice (the FIRST copy must win), and a source set that contains destination points exactly
    dst = _grid_cloud(30, 40, g)
    cases["duplicates"] = (torch.cat([dst[::3], _grid_cloud(30, 40, g)]), torch.cat([dst, dst]))
    # ties in the last bits: destinations on a coarse lattice, sources at lattice midpoints (many exactly equal distances) plus points
    # whose d2 differ by one ulp (same sqrt after rounding): the smaller index must win
    lat = torch.stack(torch.meshgrid(torch.arange(12.), torch.arange(12.), torch.arange(6.), indexing="ij"), -1).view(-1, 3)
    lat2 = lat.clone()
    lat2[:, 0] = torch.nextafter(lat2[:, 0], torch.full_like(lat2[:, 0], 100.0))
    cases["ties"] = ((lat[:400] + 0.5).contiguous(), torch.cat([lat2, lat, lat2]))
    # NaN / inf coordinates on either side
    s_bad, d_bad = _grid_cloud(20, 30, g), _grid_cloud(20, 30, g)
    s_bad[5, 1] = float("nan"); s_bad[77] = float("inf")
    d_bad[3, 0] = float("nan"); d_bad[100] = float("inf"); d_bad[200, 2] = float("-inf")
    cases["non_finite"] = (s_bad, d_bad)
    blob = {}
    for k, (s, d) in cases.items():
        blob[k + ".src"], blob[k + ".dst"] = s.numpy(), d.numpy()
    inp = str(tmp_path / "in.npz")
    np.savez(inp, **blob)
    res = {}
    for tag, extra in (("pruned", {"NNR_PC_PRUNED": "1"}), ("brute", {})):
        out = str(tmp_path / (tag + ".npz"))
        env = {k: v for k, v in os.environ.items() if k not in ("NNR_PC_PRUNED", "NNR_PC_BRUTE")}
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", CHILD, inp, out], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        res[tag] = np.load(out)
    for k in cases:
        a, b = res["pruned"][k + ".idx"], res["brute"][k + ".idx"]
        assert np.array_equal(a, b), (k, int((a != b).sum()), np.nonzero(a != b)[0][:5], a[a != b][:5], b[a != b][:5])
        assert np.array_equal(res["pruned"][k + ".dist"].view(np.uint32), res["brute"][k + ".dist"].view(np.uint32)), k
    # and against numpy on the small exact case: first index of the minimum of the float32 norms
    s, d = cases["duplicates"]
    dd = torch.linalg.norm(s[:, None, :] - d[None, :, :], dim=-1)
    assert np.array_equal(res["pruned"]["duplicates.idx"], dd.argmin(dim=1).numpy())
