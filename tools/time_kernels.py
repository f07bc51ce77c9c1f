"""Time the fused-MLP kernels at one shape (HIP events on the launch stream):  python tools/time_kernels.py R N [bf16] [reps]
NNR_LIB=<path> selects a profiling variant of the library (csrc/build.py --variant)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch

import bench

if __name__ == "__main__":
    import model as mdl
    R, N = int(sys.argv[1]), int(sys.argv[2])
    bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    dev = torch.device("cuda", 0)
    torch.manual_seed(42)
    net = mdl.OfficialStaticNerf(bench.full_cfg(R, n_samples=N)).to(dev)
    r = bench.kernel_roofline(net, dev, reps=reps, bf16=bf16, rays=R, n_samples=N, sequence_reps=20)
    print(json.dumps({"lib": os.environ.get("NNR_LIB", "product"), "shape": [R, N], "bf16": bf16,
                      "ms": {k: v["ms"] for k, v in r["kernels"].items()},
                      "in_sequence_ms": {k: v.get("sequence_ms") for k, v in r["kernels"].items() if v.get("sequence_ms") is not None}}))
