"""Run only the fused-MLP kernels of one BASELINE configs[1] step (1024 rays x 192 samples, D=256) a few times --
the command rocprofv3 wraps for --kernel-trace/--stats and for the --pmc counter passes."""
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
    print(json.dumps(bench.kernel_roofline(net, dev, reps=reps)))
