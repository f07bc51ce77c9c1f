"""Is the HIP path's convergence a sample of the reference's distribution?  

tests/golden/conv_llff_seeds.npz (oracle/gen_golden_conv.py --seeds) holds N independent runs of the REFERENCE's own training loop --
same scene, network and initial poses; frame order and every pixel pick drawn from another seed -- with their draws recorded.  This tool
replays every one of them through this repository's Trainer on the HIP kernels (tests/test_conv_reference.py::_replay), under each arm
(Adam arithmetic "single" / "fused"  x  fp32 products "split3" / "mfma"), and reports the PAIRED differences to the reference run that
made the same draws (the batch noise of a run cancels in the pair): mean, standard deviation, and a one-sample t-test of the differences
against 0, per arm; and a paired test between the two Adam arithmetics.

    python tools/conv_seeds_hip.py [--arms single:split3,fused:split3,...] [--seeds 8]  ->  JSON lines + a summary"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("nope-nerf_amd", "tools", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import torch


class _Patch:
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def main():
    import test_conv_reference as T
    import train_scene
    from nnr import lib as L
    from pathlib import Path
    from scipy import stats
    blob = np.load(os.path.join(ROOT, "tests", "golden", "conv_llff_seeds.npz"))
    arms = [("single", "split3"), ("fused", "split3"), ("single", "mfma"), ("fused", "mfma")]
    if "--arms" in sys.argv:
        arms = [tuple(a.split(":")) for a in sys.argv[sys.argv.index("--arms") + 1].split(",")]
    n_seeds = len(blob["seeds"])
    if "--seeds" in sys.argv:
        n_seeds = min(n_seeds, int(sys.argv[sys.argv.index("--seeds") + 1]))
    cols = [str(c) for c in blob["columns"]]
    ref = {c: blob["final"][:n_seeds, i] for i, c in enumerate(cols)}
    real_build, real_randperm = train_scene.build, torch.randperm
    n_steps = blob["order"].shape[1]
    rows = []
    for adam, products in arms:
        L.set_fp32_products(products)

        def build(cfg, dev, frames, _a=adam):
            cfg['training'].update(adam_arithmetic=_a)
            return real_build(cfg, dev, frames)

        train_scene.build = build
        for s in range(n_seeds):
            gold = {"order": blob["order"][s], "ray_idx": blob["ray_idx"][s], "init.pose_r": blob["init.pose_r"], "init.pose_t": blob["init.pose_t"]}
            with tempfile.TemporaryDirectory() as tmp:
                losses, psnr, errs, _, _ = T._replay(Path(tmp), torch.device("cuda"), _Patch(), n_steps, gold=gold)
            torch.randperm = real_randperm
            l20 = blob["losses20"][s]
            dev20 = float((np.abs(losses[:20] - l20) / np.maximum(1.0, np.abs(l20))).max())
            rows.append(dict(adam=adam, products=products, seed=int(blob["seeds"][s]), psnr=round(psnr, 4), ate=round(errs["ate"], 5),
                             rpe_r=round(errs["rpe_rot_deg"], 4), ref_psnr=round(float(ref["psnr"][s]), 4), ref_ate=round(float(ref["ate"][s]), 5),
                             ref_rpe_r=round(float(ref["rpe_r"][s]), 4), dev20=dev20))
            print(json.dumps(rows[-1]), flush=True)
    train_scene.build = real_build
    print("reference, %d seeds: PSNR mean %.3f std %.3f (%.2f .. %.2f); ATE mean %.4f std %.4f; RPE_r mean %.3f std %.3f"
          % (n_seeds, ref["psnr"].mean(), ref["psnr"].std(ddof=1), ref["psnr"].min(), ref["psnr"].max(), ref["ate"].mean(), ref["ate"].std(ddof=1),
             ref["rpe_r"].mean(), ref["rpe_r"].std(ddof=1)))
    by_arm = {}
    for adam, products in arms:
        r = [x for x in rows if x["adam"] == adam and x["products"] == products]
        d = {k: np.array([x[k] - x["ref_" + k] for x in r]) for k in ("psnr", "ate", "rpe_r")}
        by_arm[(adam, products)] = d
        line = "HIP adam=%s products=%s: PSNR mean %.3f std %.3f" % (adam, products, np.mean([x["psnr"] for x in r]), np.std([x["psnr"] for x in r], ddof=1))
        for k in ("psnr", "ate", "rpe_r"):
            t = stats.ttest_1samp(d[k], 0.0)
            line += "; paired d%s mean %+.4f std %.4f (t %.2f, p %.3f)" % (k, d[k].mean(), d[k].std(ddof=1), t.statistic, t.pvalue)
        line += "; first-20-steps deviation max %.2e" % max(x["dev20"] for x in r)
        print(line)
    for products in sorted({p for _, p in arms}):
        if ("single", products) in by_arm and ("fused", products) in by_arm:
            a, b = by_arm[("single", products)]["psnr"], by_arm[("fused", products)]["psnr"]
            t = stats.ttest_rel(a, b)
            print("single vs fused Adam arithmetic, products=%s: PSNR difference mean %+.4f dB (paired t %.2f, p %.3f)" % (products, (a - b).mean(), t.statistic, t.pvalue))


if __name__ == "__main__":
    main()
