"""Golden values of the SSIM metric evaluation/eval.py reports (reference model/eval_images.py:91 ->
third_party/pytorch_ssim.ssim): runs the REFERENCE function on seeded image pairs, tests/golden/ssim.npz.
Authoring container only:  python oracle/gen_golden_ssim.py"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("NNR_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ssim.npz")

if __name__ == "__main__":
    sys.path.insert(0, REF)
    from third_party import pytorch_ssim
    g = torch.Generator().manual_seed(0)
    blob = {}
    for i, (h, w, noise) in enumerate(((24, 32, 0.1), (13, 17, 0.3), (11, 40, 0.02))):
        a = torch.rand(1, 3, h, w, generator=g)
        b = (a + noise * torch.randn(1, 3, h, w, generator=g)).clamp(0, 1)
        blob[f"a{i}"], blob[f"b{i}"], blob[f"v{i}"] = a.numpy(), b.numpy(), np.float64(float(pytorch_ssim.ssim(a, b)))
        print(h, w, float(blob[f"v{i}"]))
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
