"""VERDICT r04 next #3, the cheap experiment: would a ray-chunked training step whose stash never leaves the 256 MiB memory-side cache
(Infinity Cache) be faster?  The existing forward -> compositing -> input gradient -> weight gradient sequence is run at ray counts whose
stash fits that cache (bf16: 9 KB / sample -> ~28 k samples; three-term fp32: 17.8 KB / sample -> ~14 k) and at multiples of it, each
kernel timed between the others (bench.kernel_roofline(sequence_reps=...): the weight gradient reads what the two kernels before it just
wrote).  Reported: ns per sample of every kernel per shape.  The weight gradient's plan spreads any sample count over the whole chip, so
its per-sample time isolates the memory effect; the forward / input gradient of a small chunk also lose workgroups (one per 128 / 256
samples).      python tools/chunk_experiment.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch

import bench

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    import model as mdl
    from nnr import lib as L
    cfg = bench.full_cfg(bench.R_PER_GPU)
    torch.manual_seed(42)
    net = mdl.OfficialStaticNerf(cfg).to(dev)
    for bf16, n, rays in ((True, 128, (112, 224, 448, 896, 1792, 4096)), (False, 192, (36, 76, 152, 304, 608, 1024))):
        for r in rays:
            out = bench.kernel_roofline(net, dev, reps=3, bf16=bf16, rays=r, n_samples=n, sequence_reps=12)
            k = out["kernels"]
            s = r * n
            row = {"mode": "bf16" if bf16 else "fp32-three-term", "rays": r, "samples": s,
                   "stash_MiB": round(s * (9.0 if bf16 else 17.8) / 1024, 1)}
            for name in ("mlp_fwd", "mlp_dgrad", "mlp_wgrad"):
                row[name + "_ms"] = k[name]["sequence_ms"]
                row[name + "_ns_per_sample"] = round(k[name]["sequence_ms"] * 1e6 / s, 3)
            print(json.dumps(row), flush=True)
