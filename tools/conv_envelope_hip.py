"""The HIP side of the chaos envelope (tests/test_conv_reference.py): the 800-step replay of the reference's convergence run on the
HIP kernels under implementation switches that change only last bits or nothing (fused front end on / off, one-launch Adam on / off,
torch's fused vs multi-kernel Adam) -- is the HIP path's final PSNR / ATE a sample of the same distribution as the reference's own
run-to-run spread (tests/golden/conv_llff_envelope.npz)?      python tools/conv_envelope_hip.py"""
import itertools
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
    real_build, real_randperm = train_scene.build, torch.randperm
    n = len(T.GOLD["order"])
    from nnr import lib as L
    rows = []
    # (the three-term products became the fp32 default on the strength of two replays; here every implementation switch that
    # changes last bits, under BOTH product modes -- 2 x 6 replays of the same 800 steps)
    for products, (front, one_adam, fused_adam) in itertools.product(("mfma", "split3"), itertools.product((True, False), (True, False), (True, False))):
        if one_adam and not fused_adam:
            continue
        L.set_fp32_products(products)

        def build(cfg, dev, frames, _f=front, _o=one_adam, _a=fused_adam):
            cfg['training'].update(fuse_front_end=_f, one_launch_adam=_o, fuse_optimizers=_a)
            return real_build(cfg, dev, frames)

        train_scene.build = build
        with tempfile.TemporaryDirectory() as tmp:
            from pathlib import Path
            losses, psnr, errs, _, _ = T._replay(Path(tmp), torch.device("cuda"), _Patch(), n)
        torch.randperm = real_randperm
        ref = T.GOLD["losses"]
        dev = np.abs(losses - ref) / np.maximum(1.0, np.abs(ref))
        rows.append(dict(products=products, fuse_front_end=front, one_launch_adam=one_adam, fused_adam=fused_adam, psnr=round(psnr, 3), ate=round(errs["ate"], 4),
                         rpe_r=round(errs["rpe_rot_deg"], 3), dev20=float(dev[:20].max()), dev50=float(dev[:50].max())))
        print(json.dumps(rows[-1]), flush=True)
    env = np.load(os.path.join(ROOT, "tests", "golden", "conv_llff_envelope.npz"))
    for products in ("mfma", "split3"):
        ps = [r["psnr"] for r in rows if r["products"] == products]
        at = [r["ate"] for r in rows if r["products"] == products]
        print("HIP %s: PSNR %.2f .. %.2f (mean %.3f, std %.3f), ATE %.4f .. %.4f; reference envelope PSNR %.2f .. %.2f (mean %.3f, std %.3f)"
              % (products, min(ps), max(ps), float(np.mean(ps)), float(np.std(ps)), min(at), max(at), env["runs"][:, 1].min(), env["runs"][:, 1].max(),
                 float(env["runs"][:, 1].mean()), float(env["runs"][:, 1].std())))


if __name__ == "__main__":
    main()
