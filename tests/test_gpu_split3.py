"""GPU (-m gpu): the fp32 mode with every product taken as six bf16 MFMA terms (NNR_F_SPLIT3, csrc/nnr_split.h) or -- round 6, forward and
input-gradient chain -- as three fp16 MFMA terms of two-term operands (NNR_F_SPLIT2, csrc/nnr_split2.h).

The claim to check is not "close to the fp32 kernels" but "AS CLOSE TO THE EXACT RESULT as the fp32 kernels": the whole Trainer-scope
step (12 layers, compositing, both loss heads, full backward) is evaluated in fp64 by the oracle on this host, and on the GPU twice --
fp32 MFMAs, and three-term products -- and the two errors are compared tensor by tensor.  Then the 1e-4 parity bar of `north_star`
at the benchmark shape, with the three-term products.
"""
import numpy as np
import pytest
import torch

import nerf_oracle as orc
import test_gpu_bench_shape_parity as sp

pytestmark = pytest.mark.gpu
SPLIT_KINDS = ("split3", "split2")      # held to the SAME bars (gate ii)


def _oracle64(case):
    """test_gpu_bench_shape_parity._oracle with every tensor in float64."""
    import golden_util as gu
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        t = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in gu.tensors(case).items()}
        cfg = gu.render_cfg(case)
        params = {k: v.double().clone().requires_grad_(True) for k, v in case["weights"].items()}
        leaves = {k: t[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
        loss, out = orc.train_step_scope(params, leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], int(case["cfg.cam"]),
                                         t["K"], t["depth_img"], t["img"], (sp.H, sp.W), t["ray_idx"], t["jitter"], cfg)
        loss.backward()
    finally:
        torch.set_default_dtype(prev)
    grads = {"w." + k: v.grad for k, v in params.items()}
    grads.update({k: v.grad for k, v in leaves.items()})
    out["loss"] = loss.detach()
    return out, grads


def _errors(out, grads, ref, rgrads):
    e = {}
    for k in ("rgb", "depth_pred", "alpha"):
        e["out." + k] = float((out[k].detach().cpu().double() - ref[k].detach()).abs().max())
    for k, r in rgrads.items():
        e[k] = float((grads[k].detach().cpu().double() - r).abs().max()) / max(1e-30, float(r.abs().max()))
    return e


def _rel_l2(out, grads, ref, rgrads):
    """Relative L2 per tensor against fp64: unlike max-abs / max|ref| (above), not dominated by a single flipped ReLU gate."""
    import golden_util as gu
    e = {"out." + k: gu.rel_l2(out[k].detach().cpu().double().numpy(), ref[k].detach().numpy()) for k in ("rgb", "depth_pred", "alpha")}
    e.update({k: gu.rel_l2(grads[k].detach().cpu().double().numpy(), r.numpy()) for k, r in rgrads.items()})
    return e


@pytest.mark.parametrize("D", [256, 128])
def test_three_term_products_are_as_close_to_fp64_as_fp32_mfmas(D, capsys):
    from nnr import lib as L
    from test_gpu_parity import run_hip
    case = sp._case(256, 64, D, seed=77 + D)
    ref, rgrads = _oracle64(case)
    errs, l2 = {}, {}
    # the yardstick: the CPU fp32 oracle (== the reference, bit for bit) against the same fp64 evaluation
    cout, cgrads = sp._oracle(case)
    errs["cpu"], l2["cpu"] = _errors(cout, cgrads, ref, rgrads), _rel_l2(cout, cgrads, ref, rgrads)
    prev = L.fp32_products()
    try:
        for kind in ("mfma",) + SPLIT_KINDS:
            L.set_fp32_products(kind)
            out, grads = run_hip(case)
            errs[kind] = _errors(out, grads, ref, rgrads)
            l2[kind] = _rel_l2(out, grads, ref, rgrads)
    finally:
        L.set_fp32_products(prev)
    worst = {kind: max(errs[kind].values()) for kind in errs}
    mean = {kind: float(np.mean(list(errs[kind].values()))) for kind in errs}
    with capsys.disabled():
        print("\nD=%d, 256 x 64 step vs fp64: worst relative error of the 3 outputs + 28 gradient tensors -- fp32 MFMAs %.2e, six bf16 terms "
              "%.2e, three fp16 terms %.2e; mean over tensors %.2e / %.2e / %.2e"
              % (D, worst["mfma"], worst["split3"], worst["split2"], mean["mfma"], mean["split3"], mean["split2"]))
    for kind in SPLIT_KINDS:
        for k in errs["mfma"]:
            # tensor by tensor: no worse than three times the fp32-MFMA error -- both are rounding noise (1e-7 .. 1e-6 of the tensor's scale for
            # most tensors), which of the two is smaller varies from tensor to tensor and with the split-K partition of the weight gradient;
            # what must not happen is a tensor at another ORDER (a missing term shows as 4e-4, tools/split3_debug.py).  Floor 1e-6.
            assert errs[kind][k] <= max(3.0 * errs["mfma"][k], 1e-6), (kind, k, errs[kind][k], errs["mfma"][k])
        assert mean[kind] <= 1.25 * mean["mfma"] + 1e-8, kind
    # Against the yardstick, in relative L2.  What the measurements say (profiles/r04/a_parity_rel_l2.txt,
    # a_fp64_bisect_d256.txt): stage by stage every forward plane of the HIP kernels is as close to fp64 as the CPU oracle's or closer
    # (position encoding 1.6e-5 against 1.9e-5, hidden layers 1.1-1.6e-5 against 1.3-1.9e-5), and the backward planes agree to 1e-6 ..
    # 1e-5 -- UNTIL a ReLU gate that sits within rounding of zero falls on the other side than fp64's; from that layer down the whole
    # gradient of either fp32 evaluation is 1e-4 .. 1e-3 off in relative L2.  Every case has 10-25 such gates per layer in 16 k samples,
    # on the CPU and on the GPU alike; WHICH evaluation catches the heavier ones differs from case to case (seed 333 of the bisect: CPU
    # 7e-4, HIP 1.3e-5 on the first layers' gradients; this test's first seed: CPU 1e-3, HIP 5.5e-3).  So a per-tensor bound on ONE case
    # bounds nothing; over several cases the two are the same: the geometric mean over seeds of HIP / CPU per tensor, and over all
    # tensors, is what is asserted.
    import golden_util as gu
    import os
    ratios = {kind: {k: [l2[kind][k] / max(l2["cpu"][k], 1e-6)] for k in l2["cpu"]} for kind in ("mfma",) + SPLIT_KINDS}
    for k in l2["cpu"]:
        gu.parity_log("fp64 yardstick D=%d seed %d %s: rel-L2 cpu-fp32 %.3e hip-mfma %.3e hip-split3 %.3e hip-split2 %.3e | max-abs/|ref|max cpu %.3e mfma %.3e split3 %.3e split2 %.3e"
                      % (D, 77 + D, k, l2["cpu"][k], l2["mfma"][k], l2["split3"][k], l2["split2"][k], errs["cpu"][k], errs["mfma"][k], errs["split3"][k],
                         errs["split2"][k]))
    for seed in [177 + D + 100 * i for i in range(11)]:      # 12 cases with the first one (four were thin for a heavy-tailed statistic)
        case_s = sp._case(256, 64, D, seed=seed)
        ref_s, rgrads_s = _oracle64(case_s)
        cout_s, cgrads_s = sp._oracle(case_s)
        l2c = _rel_l2(cout_s, cgrads_s, ref_s, rgrads_s)
        prev = L.fp32_products()
        try:
            for kind in ("mfma",) + SPLIT_KINDS:
                L.set_fp32_products(kind)
                out_s, grads_s = run_hip(case_s)
                l2k = _rel_l2(out_s, grads_s, ref_s, rgrads_s)
                for k in l2c:
                    ratios[kind][k].append(l2k[k] / max(l2c[k], 1e-6))
                    gu.parity_log("fp64 yardstick D=%d seed %d %s: rel-L2 cpu-fp32 %.3e hip-%s %.3e" % (D, seed, k, l2c[k], kind, l2k[k]))
        finally:
            L.set_fp32_products(prev)
    gm = lambda v: float(np.exp(np.mean(np.log(np.maximum(np.asarray(v, dtype=np.float64), 1e-12)))))
    per_tensor = {kind: {k: gm(v) for k, v in ratios[kind].items()} for kind in ratios}
    overall = {kind: gm([x for v in ratios[kind].values() for x in v]) for kind in ratios}
    worst = {kind: max(per_tensor[kind].items(), key=lambda kv: kv[1]) for kind in ratios}
    median = {kind: {k: float(np.median(v)) for k, v in ratios[kind].items()} for kind in ratios}
    worst_med = {kind: max(median[kind].items(), key=lambda kv: kv[1]) for kind in ratios}
    with capsys.disabled():
        print("D=%d: worst per-tensor MEDIAN over the 12 seeds of HIP / CPU-fp32 -- " % D
              + ", ".join("%s %.2f (%s)" % (kind, worst_med[kind][1], worst_med[kind][0]) for kind in ratios))
        print("D=%d vs fp64 in relative L2 over 12 seeds: geometric mean of HIP / CPU-fp32 over all tensors -- " % D
              + ", ".join("%s %.2f" % (kind, overall[kind]) for kind in ratios) + "; worst tensor (geometric mean over seeds) "
              + ", ".join("%s %.2f (%s)" % (kind, worst[kind][1], worst[kind][0]) for kind in ratios)
              + "; first seed alone, mean rel-L2: CPU fp32 %.2e, " % np.mean(list(l2["cpu"].values()))
              + ", ".join("%s %.2e" % (kind, np.mean(list(l2[kind].values()))) for kind in ratios))
    gu.parity_log("fp64 yardstick D=%d summary: overall geometric-mean ratio " % D + " ".join("%s %.3f" % (kind, overall[kind]) for kind in ratios)
                  + "; worst tensor " + " ".join("%s %.3f (%s)" % (kind, worst[kind][1], worst[kind][0]) for kind in ratios)
                  + "; worst median " + " ".join("%s %.3f (%s)" % (kind, worst_med[kind][1], worst_med[kind][0]) for kind in ratios))
    if os.environ.get("NNR_FP64_YARDSTICK_REPORT_ONLY") != "1":
        for kind in ratios:
            # measured (profiles/r04/b_gpu_tests.txt): overall 0.58 / 0.60 at D = 256, 0.50 / 0.49 at D = 128 -- the HIP kernels are on average CLOSER
            # to fp64 than the CPU oracle --; worst single tensor over the four seeds 4.2 - 4.8 (heavy-tailed: one gate decides a tensor)
            assert overall[kind] <= 1.5, (D, kind, overall[kind])
            # (12 seeds, profiles/r05/d_gpu_tests.txt: overall 0.49 / 0.44 at D = 256, 0.43 / 0.44 at D = 128; worst tensor 2.0 / 1.6 and 2.6 / 2.65)
            # (round 6, profiles/r06/n_yardstick_double_inv4.txt: overall 0.47-0.49 / 0.42-0.47; worst tensor 2.36 / 1.8-2.0)
            assert worst[kind][1] <= 4.0, (D, kind, worst[kind])
            # the median over the seeds is what one gate cannot move.  Rounds 4-5: 1.95-2.35, the pose rotation and layers0.6.weight, in EVERY
            # product mode.  Round 6 bisected it (profiles/r06/h_*, k_, l_, m_, n_): the render operator is at or below 1.0 stage by stage down to
            # the per-ray gradients and their rigid-motion sums; double compositing / front-end backward moved nothing (2.30 -> 2.35); the 4 x 4
            # inverses of the FORWARD ray generation did -- fp32 cofactors put 1e-7 into the ray origins, times the 2^9 encoding frequency, into
            # every gradient.  With the cofactors in double: 1.37 at D = 256, 1.55 at D = 128 (n_yardstick_double_inv4.txt) -- and a golden
            # vector of the REFERENCE out of its 1e-4 bar (pose_r 1.02e-4), the first-20-step deviation of the training run doubled: the
            # reference inverts in float, and parity with it is the gate (nnr_camera.hip inv4).  The tail is understood and stays; bar 2.5.
            assert worst_med[kind][1] <= 2.5, (D, kind, worst_med[kind])


@pytest.mark.parametrize("kind", SPLIT_KINDS)
def test_three_term_step_at_1024x192_matches_the_oracle_end_to_end(kind, capsys):
    from nnr import lib as L
    from test_gpu_parity import run_hip
    R, N, D = sp.FP32_SHAPE
    case = sp._case(R, N, D)
    prev = L.set_fp32_products(kind)
    try:
        out, grads = run_hip(case)
    finally:
        L.set_fp32_products(prev)
    ref, rgrads = sp._oracle(case)
    worst_out = 0.0
    for k in ("rgb", "depth_pred", "depth_gt", "alpha"):
        err = float((out[k].detach().cpu() - ref[k].detach()).abs().max())
        worst_out = max(worst_out, err)
        assert err <= 1e-4, (k, err)
    import golden_util as gu
    worst, worst_l2 = ("", 0.0), ("", 0.0)
    for k, r in rgrads.items():
        err = float((grads[k].detach().cpu().double() - r.double()).abs().max()) / max(1.0, float(r.abs().max()))
        worst = max(worst, (k, err), key=lambda x: x[1])
        assert err <= 1e-4, (k, err)
        rl2 = gu.rel_l2(grads[k].detach().cpu().double().numpy(), r.double().numpy())
        gu.parity_log("%s 1024x192 vs oracle %s max-abs %.3e rel-L2 %.3e ref-max %.3e" % (kind, k, err, rl2, float(r.abs().max())))
        worst_l2 = max(worst_l2, (k, rl2), key=lambda x: x[1])
        assert rl2 <= gu.REL_L2_TOL, (k, rl2)
    with capsys.disabled():
        print("\n%s products, 1024x192 D=256 end to end vs oracle: outputs %.2e, worst of 28 gradient tensors max-abs %.2e (%s), "
              "relative L2 %.2e (%s)" % (kind, worst_out, worst[1], worst[0], worst_l2[1], worst_l2[0]))


@pytest.mark.parametrize("D", [256, 128])
def test_two_term_input_gradient_survives_amplifying_layers(D, capsys):
    """The input-gradient chain of the two-term mode runs every sample in a scaled fp16 domain (nnr_mlp_dgrad_f16.hip).  Its first build set
    the scale once, at the top of the chain, with eight binades of head room: a TRAINED network's layers amplify the gradient (|W^T| of a
    hidden layer has column sums of 2 and more once the weights have grown), 2^8 over the chain is enough to overflow a conversion, and the
    step then returns NaN gradients behind a finite loss -- the reference's train.py died of it after ~300 iterations on a synthetic scene
    (tests/test_gpu_loop_rate.py, per-image losses off).  Hidden weights x 6 make every hidden layer amplify by ~2.5 (init: 0.41): the
    gradient grows by three orders of magnitude down the chain.  Statement: every gradient finite, and as close to fp64 as the fp32-MFMA
    kernels' (same order: a factor 5, floor 1e-3 -- ReLU gates decide the rest, as in the yardstick above)."""
    from nnr import lib as L
    from test_gpu_parity import run_hip
    case = sp._case(64, 64, D, seed=4242 + D)
    for k in list(case["weights"]):
        if k.startswith("layers") and k.endswith(".weight"):
            case["weights"][k] = case["weights"][k] * 6.0
    ref, rgrads = _oracle64(case)
    growth = float(rgrads["w.layers0.0.bias"].abs().max() / max(float(rgrads["w.layers1.6.bias"].abs().max()), 1e-300))
    prev = L.fp32_products()
    l2 = {}
    try:
        for kind in ("mfma", "split2"):
            L.set_fp32_products(kind)
            out, grads = run_hip(case)
            for k, g in grads.items():
                assert torch.isfinite(g).all(), (kind, k)
            assert torch.isfinite(out["rgb"]).all() and torch.isfinite(out["loss"])
            l2[kind] = _rel_l2(out, grads, ref, rgrads)
    finally:
        L.set_fp32_products(prev)
    with capsys.disabled():
        print("\nD=%d, hidden weights x 6: d(bias) of the first layer / of the last hidden layer = %.1f (fp64); worst relative L2 vs fp64: fp32 MFMAs %.2e, "
              "two fp16 terms %.2e" % (D, growth, max(l2["mfma"].values()), max(l2["split2"].values())))
    for k in l2["mfma"]:
        assert l2["split2"][k] <= max(5.0 * l2["mfma"][k], 1e-3), (k, l2["split2"][k], l2["mfma"][k])


def test_two_term_forward_activation_beyond_fp16_range_is_loud():
    """The one bound NNR_F_SPLIT2 has and the other fp32 modes do not (include/nnr.h): a hidden ACTIVATION of 65504 or more does not fit the
    first fp16 term.  It must not pass silently: the outputs are non-finite (the trainer's NaN check then stops the run, as it does for the
    reference), while the six-term mode renders the same network.  First-layer weights x 2e5 put hidden 1 at ~1e5."""
    from nnr import lib as L
    from test_gpu_parity import run_hip
    case = sp._case(32, 64, 128, seed=77)
    case["weights"]["layers0.0.weight"] = case["weights"]["layers0.0.weight"] * 2.0e5
    prev = L.fp32_products()
    try:
        L.set_fp32_products("split3")
        out3, _ = run_hip(case, eval_=True)
        L.set_fp32_products("split2")
        out2, _ = run_hip(case, eval_=True)
    finally:
        L.set_fp32_products(prev)
    assert torch.isfinite(out3["rgb"]).all()
    assert not torch.isfinite(out2["rgb"]).all()
