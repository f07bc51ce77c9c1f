"""The nearest-neighbour search of the per-image point-cloud loss (nnr_aux.hip: aux_pc_search_kernel, which walks the destination depth map's
pixel grid around the source's projection and prunes by the distance to the destination RAYS) against the exhaustive search of
nnr_pointcloud.hip through the C ABI (nnr_pc_nearest) on the SAME clouds -- the X / Y the forward kernel wrote into its workspace.
Indices and distances must be identical, bit for bit: noise depths (the bench's clouds), smooth depths (mono-depth maps), large relative
poses (clouds that barely overlap, sources behind the other camera), data-parallel shards, a scaled second cloud."""
import ctypes as C
import math
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "nope-nerf_amd"))
pytestmark = pytest.mark.gpu


def _rot(axis, ang):
    a = torch.tensor(axis, dtype=torch.float64)
    a = a / a.norm()
    K = torch.tensor([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]], dtype=torch.float64)
    return torch.eye(3, dtype=torch.float64) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)


def _depths(kind, h, w, g):
    if kind == "noise":
        return 1 + 2 * torch.rand(h, w, generator=g)
    ys, xs = torch.meshgrid(torch.linspace(0, 1, h), torch.linspace(0, 1, w), indexing="ij")
    d = 2 + 0.6 * torch.sin(5 * xs + 0.3) * torch.cos(4 * ys) + 0.02 * torch.rand(h, w, generator=g)
    if kind == "steps":      # depth discontinuities and a near-limit clamp
        d = torch.where(xs > 0.55, d * 0.35, d)
        d[: h // 8] = 0.001
    return d


def _search(hd, wd, hr, wr, kind, rel, scale2, shard=(0, 0), K=None, seed=0):
    from nnr import lib as L
    lib = L.load()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(seed)
    d1, d2 = _depths(kind, hd, wd, g).to(dev), _depths(kind, hd, wd, g).to(dev)
    if K is None:
        f = 0.7 * wd
        K = torch.diag(torch.tensor([2 * f / wd, -2 * f / hd, -1.0, 1.0]))
    Kinv = torch.linalg.inv(K.double()).float()
    K_c, Kinv_c, rel_c = (t.reshape(16).contiguous().float().to(dev) for t in (K, Kinv, rel))
    s2 = torch.tensor([scale2], dtype=torch.float32, device=dev)
    flags = L.AUX_PC | L.AUX_SCALE_PCS
    cfg = L.AuxCfg(hd, wd, hr, wr, 0.05, flags, int(shard[0]), int(shard[1]))
    n_ws = lib.nnr_aux_workspace_floats(C.byref(cfg))
    ws = torch.zeros(n_ws + 2, dtype=torch.float32, device=dev)
    ws = ws[(ws.data_ptr() % 8) // 4:]
    out = torch.empty(4, dtype=torch.float32, device=dev)
    L.check(lib.nnr_aux_terms_fwd(C.byref(cfg), L.ptr(d1), L.ptr(d2), None, None, L.ptr(K_c), L.ptr(Kinv_c), L.ptr(rel_c), L.ptr(s2), None,
                                  L.ptr(out), L.ptr(ws), L.stream()), "nnr_aux_terms_fwd")
    torch.cuda.synchronize()
    S = hr * wr
    as_i64 = lambda lo: ws[lo:lo + 2 * S].view(torch.int64)      # workspace layout: csrc/nnr_api.cpp aux_fill
    idx_xy, idx_yx = as_i64(4 * S).clone(), as_i64(6 * S).clone()
    X, Y = ws[20 * S:23 * S].view(S, 3).clone(), ws[23 * S:26 * S].view(S, 3).clone()
    dist_xy, dist_yx = ws[28 * S:29 * S].clone(), ws[29 * S:30 * S].clone()
    lo, hi = (0, S) if shard == (0, 0) else shard
    return X, Y, idx_xy[lo:hi], dist_xy[lo:hi], idx_yx[lo:hi], dist_yx[lo:hi], out, (lo, hi)


def _check(hd, wd, hr, wr, kind, rel, scale2=1.0, shard=(0, 0), K=None, seed=0):
    from nnr import pointcloud
    X, Y, ixy, dxy, iyx, dyx, out, (lo, hi) = _search(hd, wd, hr, wr, kind, rel, scale2, shard, K, seed)
    ri, rd = pointcloud.nearest(X[lo:hi].contiguous(), Y)
    assert torch.equal(ixy, ri), "X -> Y: %d of %d indices differ" % (int((ixy != ri).sum()), hi - lo)
    assert torch.equal(dxy.view(torch.int32), rd.view(torch.int32))
    ri, rd = pointcloud.nearest(Y[lo:hi].contiguous(), X)
    assert torch.equal(iyx, ri), "Y -> X: %d of %d indices differ" % (int((iyx != ri).sum()), hi - lo)
    assert torch.equal(dyx.view(torch.int32), rd.view(torch.int32))
    if shard == (0, 0):
        S = hr * wr
        want = (dxy.double().sum() + dyx.double().sum()) / S
        assert abs(float(out[0]) - float(want)) <= 1e-5 * float(want)


def _rel(axis=(0.0, 1.0, 0.0), ang=0.0, t=(0.0, 0.0, 0.0)):
    m = torch.eye(4, dtype=torch.float64)
    m[:3, :3] = _rot(axis, ang)
    m[:3, 3] = torch.tensor(t, dtype=torch.float64)
    return m.float()


CASES = {
    "noise_identity": ("noise", _rel(), 1.0),
    "noise_small_pose": ("noise", _rel((0.2, 1.0, 0.1), 0.05, (0.05, -0.02, 0.03)), 1.0),
    "smooth_small_pose": ("smooth", _rel((0.2, 1.0, 0.1), 0.03, (0.04, 0.01, -0.02)), 1.0),
    "smooth_scaled": ("smooth", _rel((1.0, 0.3, 0.0), 0.02, (0.0, 0.02, 0.0)), 1.7),
    "steps_medium_pose": ("steps", _rel((0.0, 1.0, 0.2), 0.4, (0.8, 0.1, -0.3)), 1.0),
    "smooth_large_pose": ("smooth", _rel((0.1, 1.0, 0.0), 1.2, (2.5, 0.0, -1.0)), 1.0),      # barely overlapping clouds
    "noise_behind": ("noise", _rel((0.0, 1.0, 0.0), 3.0, (0.2, 0.0, -4.0)), 0.6),            # the other camera looks the other way
    "smooth_sideways": ("smooth", _rel((0.0, 1.0, 0.0), math.pi / 2, (2.0, 0.0, -2.0)), 1.0),  # sources beside the camera plane (p_z ~ 0)
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_ray_aware_search_equals_the_exhaustive_search(name):
    kind, rel, s2 = CASES[name]
    _check(96, 128, 48, 64, kind, rel, s2)


def test_first_phase_grid_and_a_data_parallel_shard():
    """540 x 960 maps on the 135 x 240 grid of pc_ratio 4 (the bench's first-phase step), noise and smooth depths; then a shard of the sources."""
    rel = _rel((0.2, 1.0, 0.1), 0.04, (0.05, -0.02, 0.03))
    _check(540, 960, 135, 240, "noise", rel)
    _check(540, 960, 135, 240, "smooth", rel, seed=3)
    _check(540, 960, 135, 240, "smooth", rel, shard=(8100, 16211), seed=4)


def test_general_intrinsics_with_skew_and_principal_point():
    K = torch.tensor([[1.3, 0.07, 0.11, 0.0], [0.0, -2.2, -0.05, 0.0], [0.0, 0.0, -1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    _check(96, 128, 48, 64, "noise", _rel((0.3, 1.0, 0.2), 0.1, (0.1, 0.05, 0.02)), 1.0, K=K)
    _check(96, 128, 48, 64, "smooth", _rel((0.3, 1.0, 0.2), 0.1, (0.1, 0.05, 0.02)), 1.3, K=K, seed=5)


def test_weighted_sum_and_its_backward_equal_the_torch_expression():
    """NNR_AUX_WEIGHTED: out[3] = w_pc loss_pc + w_rgbs loss_rgb_s from the finishing kernel, with its own backward (one upstream gradient,
    the weights applied in the kernels) -- against the same losses combined by torch, including the mixed case where the separate losses
    carry gradients of their own."""
    from nnr import aux as nnr_aux
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(11)
    hd, wd, hr, wr = 96, 128, 48, 64
    d1, d2 = _depths("smooth", hd, wd, g).to(dev)[None, None], _depths("smooth", hd, wd, g).to(dev)[None, None]
    i1, i2 = torch.rand(1, 3, hr, wr, generator=g).to(dev), torch.rand(1, 3, hr, wr, generator=g).to(dev)
    f = 0.7 * wd
    K = torch.diag(torch.tensor([2 * f / wd, -2 * f / hd, -1.0, 1.0]))[None].to(dev)
    Kinv = torch.linalg.inv(K)

    def leaves():
        rel = _rel((0.2, 1.0, 0.1), 0.03, (0.04, 0.01, -0.02))[None].to(dev).requires_grad_(True)
        s2 = torch.tensor(1.3, device=dev, requires_grad=True)
        aff = torch.tensor([1.1, 0.05, 0.9, -0.02], device=dev, requires_grad=True)
        return rel, s2, aff

    w = (0.7, 1.3)
    for extra_pc, extra_rgbs in ((0.0, 0.0), (0.5, 0.25)):
        rel, s2, aff = leaves()
        out = nnr_aux.aux_terms(d1, d2, rel, s2, i1, i2, K, Kinv, (hr, wr), 0.05, aff=aff, weights=w)
        assert len(out) == 4
        loss = out[3] + extra_pc * out[0] + extra_rgbs * out[1] if (extra_pc or extra_rgbs) else out[3]
        loss.backward()
        rel_b, s2_b, aff_b = leaves()
        l_pc, l_rgbs, _ = nnr_aux.aux_terms(d1, d2, rel_b, s2_b, i1, i2, K, Kinv, (hr, wr), 0.05, aff=aff_b)
        want = w[0] * l_pc + w[1] * l_rgbs
        assert float(out[3].detach()) == float(want.detach())                                      # the same two products and sum, rounded the same way
        ((want + extra_pc * l_pc + extra_rgbs * l_rgbs) if (extra_pc or extra_rgbs) else want).backward()
        for a, b, name in ((rel, rel_b, "rel"), (s2, s2_b, "scale2"), (aff, aff_b, "aff")):
            scale = max(1e-6, float(b.grad.abs().max()))
            assert float((a.grad - b.grad).abs().max()) / scale <= 2e-6, (name, extra_pc)


@pytest.mark.parametrize("which", ["rows", "tiles"])
def test_each_search_kernel_alone_equals_the_exhaustive_search(which):
    """The product search runs the light sources in the eight-lanes-per-source kernel and the heavy 8 x 8 tiles in the wave-per-tile kernel;
    NNR_PC_SEARCH=rows / tiles (read once per process) runs ONE of them on everything: the cases above again, in a child process."""
    import subprocess
    env = dict(os.environ, NNR_PC_SEARCH=which)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "exhaustive_search and not alone or first_phase or intrinsics",
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
