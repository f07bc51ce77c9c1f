"""GPU (-m gpu): data parallelism with two REAL processes sharing the one GPU of the test box (gloo carries the all-reduce of the
device tensors; RCCL refuses two ranks on one device).  Both ranks run model.Trainer.train_step on the HIP kernels; the
all-reduced gradients and loss must equal the single-process step on all rays.  Complements the virtual-rank test
(test_gpu_dp.py) and the CPU gloo test (test_parallel_gloo.py)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
N_RAYS = 96


def _worker(rank, world, port, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(os.path.dirname(here), "nope-nerf_amd"), os.path.join(os.path.dirname(here), "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import golden_util as gu
    from test_gpu_dp import _trainer
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    case = gu.load_case("tanks_d128")
    tr, mods, data = _trainer(case, N_RAYS)
    torch.manual_seed(321)
    torch.cuda.manual_seed(321)
    ld = tr.train_step(data, it=0, epoch=0, scheduling_start=10000, render_path=None)
    torch.cuda.synchronize()
    grads = [p.grad.detach().cpu().numpy().reshape(-1) for m in mods for p in m.parameters()]
    np.savez(os.path.join(out_dir, f"w{world}_r{rank}.npz"), g=np.concatenate(grads),
             loss=np.array([float(ld[k]) for k in ("loss", "loss_rgb", "loss_depth")]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_two_processes_reproduce_the_single_process_step():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(1, port, d), nprocs=1, join=True)
        try:
            mp.spawn(_worker, args=(2, port, d), nprocs=2, join=True)
        except Exception as e:   # a torch build whose gloo cannot move device tensors
            if "gloo" in str(e).lower() or "not supported" in str(e).lower():
                pytest.skip("gloo cannot all-reduce device tensors in this build: %s" % str(e)[:200])
            raise
        ref = np.load(os.path.join(d, "w1_r0.npz"))
        r0, r1 = np.load(os.path.join(d, "w2_r0.npz")), np.load(os.path.join(d, "w2_r1.npz"))
    np.testing.assert_array_equal(r0["g"], r1["g"])                         # one all-reduce: identical on both ranks
    scale = max(1.0, float(np.abs(ref["g"]).max()))
    assert float(np.abs(r0["g"] - ref["g"]).max()) / scale <= 1e-5
    np.testing.assert_allclose(r0["loss"], ref["loss"], rtol=0, atol=1e-5)
