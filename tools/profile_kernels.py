"""Run only the fused-MLP kernels of one step a few times -- the command rocprofv3 wraps for --kernel-trace/--stats and for the
--pmc counter passes.   python tools/profile_kernels.py [reps [R N [bf16]]]   (default: BASELINE configs[1], 1024 x 192, fp32)"""
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
    cfg = bench.full_cfg(bench.R_PER_GPU)
    torch.manual_seed(42)
    net = mdl.OfficialStaticNerf(cfg).to(dev)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    R, N = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (bench.R_PER_GPU, bench.N_SAMPLES)
    bf16 = len(sys.argv) > 4 and sys.argv[4] == "bf16"
    print(json.dumps(bench.kernel_roofline(net, dev, reps=reps, bf16=bf16, rays=R, n_samples=N)))
