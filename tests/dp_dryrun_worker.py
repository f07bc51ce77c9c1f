"""One rank of the 8-rank CPU dry run of the data-parallel step (tests/test_parallel_gloo.py::test_eight_ranks_through_the_bench_launcher...).

Started by bench.self_launch -- the very launcher `bench.py --gpus N` uses: torch.distributed.run, --master-addr 127.0.0.1, one process per rank --
with RANK / WORLD_SIZE / MASTER_* in the environment.  Joins a gloo group, builds the Trainer of tests/test_parallel_gloo.py (the HIP render
operator swapped for the CPU oracle backend: TEST infrastructure, no GPU here), runs two sharded training steps and -- rank 0 -- writes losses and
the all-reduced gradients of both steps to the .npz named on the command line, plus one JSON line with the `collective` block bench.py prints."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "nope-nerf_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("NNR_NO_AUTO_DIST", "1")      # this script joins the group itself (as bench.py does)

import numpy as np
import torch
import torch.distributed as dist


def run_steps(name, n_rays, steps=2):
    """`steps` training steps of the step scope on one golden case's inputs, n_rays per step in total; returns per step (losses, gradients)."""
    import golden_util as gu
    from test_parallel_gloo import _build
    tr, net, pose, distn, data = _build(gu.load_case(name), n_rays)
    torch.manual_seed(123)                      # same permutation and jitter stream on every rank
    out = []
    for it in range(steps):
        ld = tr.train_step(data, it=it, epoch=0, scheduling_start=10000, render_path=None)
        grads = {k: v.grad.clone().numpy() for k, v in net.named_parameters()}
        grads.update(r=pose.r.grad.clone().numpy(), t=pose.t.grad.clone().numpy(), scale=distn.global_scales.grad.clone().numpy(),
                     shift=distn.global_shifts.grad.clone().numpy())
        out.append(({k: float(ld[k]) for k in ("loss", "loss_rgb", "loss_depth")}, grads))
    return out


def main():
    name, n_rays, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = run_steps(name, n_rays)
        import bench
        n_grad = sum(int(np.prod(v.shape)) for v in res[0][1].values()) + 9
        ranks_seen, ar_us = bench.allreduce_probe(torch.device("cpu"), n_grad, reps=3)
        if rank == 0:
            flat = {}
            for i, (losses, grads) in enumerate(res):
                for k, v in losses.items():
                    flat["s%d.loss.%s" % (i, k)] = np.float64(v)
                for k, v in grads.items():
                    flat["s%d.grad.%s" % (i, k)] = v
            np.savez(out_path, **flat)
            print(json.dumps({"n_gpus": world, "collective": {"backend": "gloo (CPU dry run)", "rccl_ranks_seen": ranks_seen,
                                                              "allreduce_us": round(ar_us, 1), "bucket_floats": n_grad}}), flush=True)
    finally:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
