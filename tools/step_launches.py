"""Which Python line launches each small kernel of a training step: torch.profiler with stacks over a few steps of bench.py's
trainer.   python tools/step_launches.py [--bf16] [--aux] [R N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch
from torch.profiler import ProfilerActivity, profile

import bench

if __name__ == "__main__":
    bf16 = "--bf16" in sys.argv
    aux = "--aux" in sys.argv      # the first training phase: per-image losses on
    nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
    R, N = (nums + [bench.R_PER_GPU, bench.N_SAMPLES])[:2] if len(nums) >= 2 else (bench.R_PER_GPU, bench.N_SAMPLES)
    dev = torch.device("cuda", 0)
    trainer, net = bench.build_trainer(dev, 1, aux, bf16, R, N)
    data = bench.synthetic_batch(dev)
    run = lambda it: trainer.train_step(data, it=it, epoch=0, scheduling_start=10000, render_path=None)
    for it in range(1, 6):
        run(it)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for it in range(6, 9):
            run(it)
        torch.cuda.synchronize()
    rows = {}
    # group device kernels by (kernel name, innermost repo frame of the launching op)
    for e in prof.key_averages(group_by_stack_n=12):
        if e.device_time_total <= 0 or e.self_device_time_total <= 0:
            continue
        stack = [s for s in e.stack if "/nope-nerf_amd/" in s or "bench.py" in s]
        where = stack[0].split("/nope-nerf_amd/")[-1] if stack else (e.stack[0] if e.stack else "?")
        rows[(e.key, where)] = rows.get((e.key, where), 0) + e.self_device_time_total
    for (k, w), t in sorted(rows.items(), key=lambda kv: -kv[1])[:60]:
        print("%8.1f us/step  %-46s %s" % (t / 3.0, k[:46], w[:110]))
    print("\nfill / zero / copy ops with shapes (3 steps):")
    import collections
    cnt = collections.Counter()
    for e in prof.events():
        if e.name in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::ones", "aten::ones_like", "aten::full", "aten::where", "aten::index", "aten::mul", "aten::ne", "aten::lt", "aten::clone", "aten::contiguous", "aten::cat", "aten::to") and e.device_type.name == "CPU":
            cnt[(e.name, str(e.input_shapes)[:80])] += 1
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:50]:
        print("  x%.1f/step %s %s" % (v / 3.0, k[0], k[1]))
