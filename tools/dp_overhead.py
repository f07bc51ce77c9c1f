"""What a data-parallel step costs beyond the single-GPU step, measured on ONE GPU: rank 0 of a virtual W-rank job renders its
1024-ray shard of the (1024 W)-ray step -- the whole step's pixel pick and jitter draw, the global valid-depth count, the
flat gradient bucket, an all-reduce issued on a real one-rank RCCL group (launch cost without link time) and the copy-back.
The driver measures the real N-GPU runs; this isolates the host/launch overhead we control.

    python tools/dp_overhead.py [--world 8] [--steps 40] [--aux]

--aux: with the first-phase per-image losses on (point cloud + surface reprojection).  Their block is sharded by source points
(nnr_aux_cfg.shard_lo / shard_hi): the extra time of a rank over its aux-free step should fall as 1 / W."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch
import torch.distributed as dist

import bench


def timed(trainer, data, steps, warmup=8):
    for i in range(warmup):
        trainer.train_step(data, it=i, epoch=0, scheduling_start=10000, render_path=None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        trainer.train_step(data, it=warmup + i, epoch=0, scheduling_start=10000, render_path=None)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rank", type=int, default=0, help="which rank of the virtual job this process plays")
    ap.add_argument("--skip-single", action="store_true", help="only the virtual-rank run (for a kernel trace of it)")
    ap.add_argument("--aux", action="store_true", help="first-phase per-image losses on")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    data = bench.synthetic_batch(dev)
    trainer, _ = bench.build_trainer(dev, 1, a.aux)
    single = float('nan') if a.skip_single else timed(trainer, data, a.steps)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from nnr import parallel
    parallel.world_size = lambda: a.world
    parallel.rank = lambda: a.rank
    trainer, _ = bench.build_trainer(dev, a.world, a.aux)
    virtual = timed(trainer, data, a.steps)
    print(json.dumps({"aux": a.aux, "world": a.world, "single_ms": round(single, 4), "virtual_rank": a.rank, "virtual_rank_ms": round(virtual, 4),
                      "overhead_ms": round(virtual - single, 4), "overhead_frac": round(virtual / single - 1, 4)}))
    dist.destroy_process_group()
