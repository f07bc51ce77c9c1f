"""How far is bf16-product arithmetic from the reference's fp32, layer by layer -- and would keeping the two encoding-fed products
(hidden 1, the posenc columns of the skip layer: inputs sin / cos(2^9 x)) in fp32 change that?  CPU only: oracle/nerf_oracle.py's
bf16 emulation reproduces the HIP kernels' gradients to 1e-3 (tests/test_gpu_parity.py), so the question is answered without a GPU.

    python tools/bf16_error_study.py > profiles/r02/bf16_error_study.txt

Prints, for two golden cases, the relative L2 deviation of every weight gradient and the pose gradients from the fp32 reference
golden: all products in bf16 (what the kernels do) against the variant with the encoding-fed products in fp32."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("oracle", "tests", "nope-nerf_amd"):
    sys.path.insert(0, os.path.join(ROOT, p))
import golden_util as gu      # noqa: E402
import nerf_oracle as orc     # noqa: E402


def mlp_fp32_encoding_layers(params, pts, viewdir, *, dist_alpha, occ_activation="softplus", pos_levels=10, dir_levels=4):
    W = lambda n: params[n + ".weight"]
    B = lambda n: params[n + ".bias"]
    lin16 = lambda n, v: orc._RoundGrad.apply(orc._Bf16Matmul.apply(v, W(n))) + B(n)
    e = orc.posenc(pts, pos_levels)
    h = F.relu(F.linear(e, W("layers0.0"), B("layers0.0")))                               # fp32
    for n in ("layers0.2", "layers0.4", "layers0.6"):
        h = F.relu(lin16(n, h))
    D = h.shape[1]
    w5 = W("layers1.0")
    h = F.relu(orc._RoundGrad.apply(orc._Bf16Matmul.apply(h, w5[:, :D])) + F.linear(e, w5[:, D:]) + B("layers1.0"))   # e-part fp32
    for n in ("layers1.2", "layers1.4", "layers1.6"):
        h = F.relu(lin16(n, h))
    raw = F.linear(h, W("fc_density"), B("fc_density"))
    occ = F.softplus(raw) if occ_activation == "softplus" else raw.relu()
    if not dist_alpha:
        occ = 1 - torch.exp(-occ)
    wg = W("rgb_layers.0")
    wm, bm = wg[:, :D] @ W("fc_feature"), wg[:, :D] @ B("fc_feature") + B("rgb_layers.0")
    pre = orc._RoundGrad.apply(orc._Bf16Matmul.apply(h, wm) + orc._Bf16Matmul.apply(orc.posenc(viewdir, dir_levels), wg[:, D:])) + bm
    return torch.sigmoid(F.linear(F.relu(pre), W("fc_rgb"), B("fc_rgb"))), occ


def deviations(case, net):
    keep, orc.mlp_bf16 = orc.mlp_bf16, net
    try:
        t, cfg = gu.tensors(case), gu.render_cfg(case)
        cfg["mfma_dtype"] = "bf16"
        h, w, cam = int(case["cfg.h"]), int(case["cfg.w"]), int(case["cfg.cam"])
        params = {k: v.clone().requires_grad_(True) for k, v in case["weights"].items()}
        leaves = {k: t[k].clone().requires_grad_(True) for k in ("pose_r", "pose_t", "scales", "shifts")}
        loss, out = orc.train_step_scope(params, leaves["pose_r"], leaves["pose_t"], leaves["scales"], leaves["shifts"], cam, t["K"],
                                         t["depth_img"], t["img"], (h, w), t["ray_idx"], t["jitter"], cfg)
        loss.backward()
    finally:
        orc.mlp_bf16 = keep
    gold = gu.golden_grads(case)

    def rel(key, a):
        kind, ref, _ = gold[key]
        a = a.detach().numpy().reshape(-1)
        if kind != "full":
            a = a[::a.size // gu.SUBSAMPLE]
        return float(np.linalg.norm(a - ref.reshape(-1)) / np.linalg.norm(ref))
    res = {k: rel("w." + k, v.grad) for k, v in params.items() if k.endswith("weight")}
    res.update(pose_r=rel("pose_r", leaves["pose_r"].grad), pose_t=rel("pose_t", leaves["pose_t"].grad))
    res["rgb (max abs)"] = float(np.abs(out["rgb"].detach().numpy() - case["out.rgb"]).max())
    return res


if __name__ == "__main__":
    torch.set_num_threads(8)
    for name in ("tanks_d128", "tanks_d256_n192"):
        case = gu.load_case(name)
        a, b = deviations(case, orc.mlp_bf16), deviations(case, mlp_fp32_encoding_layers)
        print("%s: deviation from the fp32 reference golden (relative L2 per gradient tensor)" % name)
        print("  %-22s %12s %28s" % ("tensor", "all bf16", "encoding-fed products fp32"))
        for k in a:
            print("  %-22s %12.3e %28.3e" % (k, a[k], b[k]))
