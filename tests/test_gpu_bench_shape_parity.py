"""GPU (-m gpu): END-TO-END parity at the BENCHMARK shapes, against the oracle evaluated on this host.

The golden cases pin the path against the reference at sizes the reference generated in seconds (<= 96 rays); at the benchmark
shapes the suite so far had a forward check on 16 rays and layer-local checks against operands the kernels stashed themselves.
Here the whole Trainer-scope step -- pose -> c2w -> inverse, depth distortion, mono-depth gather, ray generation, sampling, encodings,
the 12 layers, compositing, both loss heads, full backward -- runs

* at BASELINE configs[1] (1024 rays x 192 samples, D = 256, fp32) on the HIP kernels and through `oracle.train_step_scope` on the host
  (a few seconds), and every output and all 28 gradient tensors (12 weights, 12 biases, pose r / t, scale, shift) are compared at the
  1e-4 bar of `north_star`;
* at BASELINE configs[2] (4096 rays x 128 samples, D = 256, bf16 products): rays are independent, so a 256-ray subset of the 4096-ray
  step -- its loss normalised by the same global counts -- is the subset's own step; the HIP kernels run all 4096 rays, the
  bf16-arithmetic oracle the 256, and the subset's gradients are isolated on the HIP side by loss heads whose gradient is zero outside
  the subset (the backward is linear in it; launch shapes, plans and kernels are the benchmark's).
"""
import os
import sys
import time

import numpy as np
import pytest
import torch

import golden_util as gu
import nerf_oracle as orc

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # bench.py (the stated bf16 tolerances)
H, W = 270, 480
FP32_SHAPE = (1024, 192, 256)      # BASELINE configs[1]
BF16_SHAPE = (4096, 128, 256)      # BASELINE configs[2]


def _case(R, N, D, seed=1234):
    """An in-memory case with the keys of a tests/golden/*.npz file (see oracle/gen_golden.py::make_inputs)."""
    g = torch.Generator().manual_seed(seed)
    f = 0.7 * W
    K = torch.diag(torch.tensor([2 * f / W, -2 * f / H, -1.0, 1.0])).unsqueeze(0)
    case = {
        "cfg.hidden": D, "cfg.N": N, "cfg.dist_alpha": 0, "cfg.ndc": 0, "cfg.near": 0.01, "cfg.far": 10.0, "cfg.normalise_ray": 1,
        "cfg.white": 0, "cfg.h": H, "cfg.w": W, "cfg.cam": 1, "cfg.eval": 0,
        "in.K": K.numpy(), "in.pose_r": (0.01 * torch.randn(gu.N_CAMS, 3, generator=g)).numpy(),
        "in.pose_t": (0.01 * torch.randn(gu.N_CAMS, 3, generator=g)).numpy(),
        "in.scales": (1 + 0.05 * torch.randn(gu.N_CAMS, 1, generator=g)).numpy(),
        "in.shifts": (0.05 * torch.randn(gu.N_CAMS, 1, generator=g)).numpy(),
        "in.depth_img": (1 + 2 * torch.rand(1, 1, H, W, generator=g)).numpy(),
        "in.img": torch.rand(1, 3, H, W, generator=g).numpy(),
        "in.ray_idx": torch.randperm(H * W, generator=g)[:R].numpy(),
        "in.jitter": torch.rand(1, R, N, generator=g).numpy(),
    }
    case["weights"] = orc.init_params(D, seed + 1)
    return case


def _oracle(case, cfg_extra=None, subset=None, n_total=None):
    t = gu.tensors(case)
    cfg = gu.render_cfg(case)
    cfg.update(cfg_extra or {})
    cam = int(case["cfg.cam"])
    ray_idx, jitter = t["ray_idx"], t["jitter"]
    if subset is not None:
        ray_idx, jitter = ray_idx[subset], jitter[:, subset]
    params = {k: v.clone().requires_grad_(True) for k, v in case["weights"].items()}
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
    loss, out = orc.train_step_scope(params, leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], cam, t["K"],
                                     t["depth_img"], t["img"], (H, W), ray_idx, jitter, cfg)
    if subset is not None:      # both heads are means over the rays of the step: the subset's share of the 4096-ray loss
        loss = loss * (len(subset) / n_total)
    loss.backward()
    grads = {"w." + k: v.grad for k, v in params.items()}
    grads.update({k: v.grad for k, v in leaves.items()})
    out["loss"] = loss.detach()
    return out, grads


def test_fp32_step_at_1024x192_matches_the_oracle_end_to_end(capsys):
    from test_gpu_parity import run_hip
    R, N, D = FP32_SHAPE
    case = _case(R, N, D)
    t0 = time.perf_counter()
    out, grads = run_hip(case)
    torch.cuda.synchronize()
    t_hip = time.perf_counter() - t0
    t0 = time.perf_counter()
    ref, rgrads = _oracle(case)
    t_orc = time.perf_counter() - t0
    np.testing.assert_allclose(out["z_vals"].cpu().numpy(), ref["z_vals"].detach().numpy(), rtol=0, atol=1e-6)
    worst_out = 0.0
    for k in ("rgb", "depth_pred", "depth_gt", "alpha"):
        got, want = out[k].detach().cpu(), ref[k].detach()
        assert got.shape == want.shape, k
        err = float((got - want).abs().max())
        worst_out = max(worst_out, err)
        assert err <= 1e-4, (k, err)
    assert abs(float(out["loss"]) - float(ref["loss"])) <= 1e-4
    assert len(rgrads) == 28
    worst, worst_l2 = ("", 0.0), ("", 0.0)
    for k, r in rgrads.items():
        g = grads[k].detach().cpu().double()
        r = r.double()
        scale = max(1.0, float(r.abs().max()))
        err = float((g - r).abs().max()) / scale
        worst = max(worst, (k, err), key=lambda x: x[1])
        assert err <= 1e-4, (k, err)
        # the bar above is ABSOLUTE for every gradient tensor (all stay below 1): the relative one is what would see a wrong small tensor
        rl2 = gu.rel_l2(g.numpy(), r.numpy())
        gu.parity_log("fp32 1024x192 vs oracle %s max-abs %.3e rel-L2 %.3e ref-max %.3e" % (k, err, rl2, float(r.abs().max())))
        worst_l2 = max(worst_l2, (k, rl2), key=lambda x: x[1])
        assert rl2 <= gu.REL_L2_TOL, (k, rl2)
    with capsys.disabled():
        print("\nfp32 1024x192 D=256 end to end vs oracle: outputs %.2e, worst of 28 gradient tensors max-abs %.2e (%s), relative L2 %.2e (%s); "
              "HIP %.2f s (first call), oracle %.2f s" % (worst_out, worst[1], worst[0], worst_l2[1], worst_l2[0], t_hip, t_orc))


def test_bf16_step_at_4096x128_matches_the_bf16_oracle_on_a_256_ray_subset(capsys):
    """configs[2] at full size.  HIP: all 4096 rays; oracle (bf16 arithmetic, nerf_oracle.mlp_bf16): rays 0, 16, 32, ...  The subset's
    gradients on the HIP side = the step's gradients with d(loss)/d(output) zeroed outside the subset: same launch shapes, same
    kernels, same plans as the benchmark."""
    import nnr                                     # noqa: F401  (the package must be importable before model)
    from test_gpu_parity import build
    from model.common import arange_pixels
    R, N, D = BF16_SHAPE
    case = _case(R, N, D, seed=4321)
    sub = torch.arange(0, R, 16)
    dev = torch.device("cuda")
    net, model, pose, dist, t = build(case, dev)
    model.renderer.cfg['mfma_dtype'] = 'bf16'
    cam = int(case["cfg.cam"])
    ray_idx = t["ray_idx"].to(dev)
    world_mat = torch.inverse(pose(cam)).unsqueeze(0)
    sc, sh = dist(cam)
    depth_in = t["depth_img"].to(dev) * sc + sh
    p = arange_pixels((H, W), device=dev)[1][:, ray_idx]
    jit = t["jitter"].to(dev)
    orig = torch.rand
    torch.rand = lambda *a, **k: jit
    t0 = time.perf_counter()
    try:
        out = model(p, ray_idx, t["K"].to(dev), world_mat, torch.eye(4, device=dev)[None], 'nope_nerf', it=0, eval_mode=False,
                    depth_img=depth_in, add_noise=True, img_size=(H, W))
    finally:
        torch.rand = orig
    # loss heads restricted to the subset, normalised by the counts of the WHOLE step (model/losses.py:27-32,59-64: means over rays)
    keep = torch.zeros(R, device=dev)
    keep[sub.to(dev)] = 1.0
    rgb_gt = t["img"].to(dev).view(1, 3, H * W).permute(0, 2, 1)[:, ray_idx]
    assert out["depth_pred"].shape[-1] == R          # no masked depths in this case: per-ray tensors line up with `keep`
    l_rgb = ((out["rgb"] - rgb_gt).abs().sum(-1) * keep).sum() / R
    l_dep = ((out["depth_pred"] - out["depth_gt"]).abs().reshape(-1) * keep).sum() / R
    (l_rgb + 0.04 * l_dep).backward()
    torch.cuda.synchronize()
    t_hip = time.perf_counter() - t0
    grads = {"w." + k: v.grad for k, v in net.named_parameters()}
    grads.update(pose_r=pose.r.grad, pose_t=pose.t.grad, scales=dist.global_scales.grad, shifts=dist.global_shifts.grad)
    t0 = time.perf_counter()
    ref, rgrads = _oracle(case, {"mfma_dtype": "bf16"}, subset=sub, n_total=R)
    t_orc = time.perf_counter() - t0
    for k in ("rgb", "depth_pred"):
        got = out[k].detach().cpu().reshape(R, -1)[sub]
        err = float((got - ref[k].detach().reshape(len(sub), -1)).abs().max())
        assert err <= 1e-4, (k, err)
    assert abs(float(l_rgb + 0.04 * l_dep) - float(ref["loss"])) <= 1e-4
    worst = ("", 0.0)
    for k, r in rgrads.items():
        g = grads[k].detach().cpu().double()
        r = r.double()
        if float(r.abs().max()) == 0:
            assert float(g.abs().max()) == 0, k
            continue
        l2 = float((g - r).norm() / r.norm())
        worst = max(worst, (k, l2), key=lambda x: x[1])
        assert l2 <= 2.5e-2, (k, l2)
    with capsys.disabled():
        print("\nbf16 4096x128 D=256 (256-ray subset) vs bf16-arithmetic oracle: worst gradient relative L2 %.2e (%s); HIP %.2f s, oracle %.2f s"
              % (worst[1], worst[0], t_hip, t_orc))
    # ... and against the FP32 oracle on the same subset: what BASELINE configs[2] ("tolerance vs fp32 ref") asks to be STATED.  bf16 products in a ReLU
    # network flip the gates of near-zero pre-activations, so this is a statement about the arithmetic, not about the kernels (the comparison
    # above is); the figures are printed into the suite's log and carried by bench.py's bf16_4096x128 block (BF16_TOLERANCE_VS_FP32).
    ref32, rgrads32 = _oracle(case, None, subset=sub, n_total=R)
    out_err = {}
    for k in ("rgb", "depth_pred"):
        got = out[k].detach().cpu().reshape(R, -1)[sub]
        out_err[k] = float((got - ref32[k].detach().reshape(len(sub), -1)).abs().max())
    gl2, gmax = {}, {}
    for k, r in rgrads32.items():
        g = grads[k].detach().cpu().double()
        r = r.double()
        if float(r.abs().max()) == 0:
            continue
        gl2[k] = float((g - r).norm() / r.norm())
        gmax[k] = float((g - r).abs().max()) / float(r.abs().max())
        gu.parity_log("bf16 4096x128 vs FP32 oracle (256-ray subset) %s rel-L2 %.3e max-abs/|ref|max %.3e" % (k, gl2[k], gmax[k]))
    wl2, wmx = max(gl2.items(), key=lambda kv: kv[1]), max(gmax.items(), key=lambda kv: kv[1])
    with capsys.disabled():
        print("bf16 4096x128 D=256 (256-ray subset) vs FP32 oracle: rgb max-abs %.2e, depth_pred max-abs %.2e; 28 gradient tensors: relative L2 median %.3f, "
              "worst %.3f (%s); max-abs / max|ref| worst %.3f (%s)"
              % (out_err["rgb"], out_err["depth_pred"], float(np.median(list(gl2.values()))), wl2[1], wl2[0], wmx[1], wmx[0]))
    # the STATED tolerances of the mode (bench.py: BF16_TOLERANCE_VS_FP32): outputs 5e-4 absolute, gradients 0.25 in relative L2
    import bench
    tol = bench.BF16_TOLERANCE_VS_FP32
    assert out_err["rgb"] <= tol["rgb_max_abs"] and out_err["depth_pred"] <= tol["depth_max_abs"], out_err
    assert wl2[1] <= tol["grad_rel_l2"], wl2
