"""Time of the pixel pick (nnr.sampling.randperm_prefix: the torch key draw + select + finish) and of the jitter rows of a shard
(nnr.sampling.rand_rows) against what they replace.   python tools/time_pick.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch

from nnr import sampling

dev = torch.device("cuda", 0)
torch.cuda.init()
torch.zeros(1, device=dev)


def t(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


n = 540 * 960
for r in (1024, 4096, 8192, 32768):
    sampling._state["checked"] = sampling._CHECKS
    print("pick of %5d from %d: %7.1f us   (torch.randperm[:r]: %7.1f us)" % (r, n, t(lambda: sampling.randperm_prefix(n, r, dev)), t(lambda: torch.randperm(n, device=dev)[:r], 50)))
for world in (2, 8):
    total, rows = world * 1024 * 192, 1024 * 192
    sampling._rows_state["checked"] = sampling._CHECKS
    print("jitter rows of one of %d ranks (1024 x 192 of %d x 192): %6.1f us   (whole tensor + slice: %6.1f us)"
          % (world, world * 1024, t(lambda: sampling.rand_rows(total, rows, rows, dev)), t(lambda: torch.rand(1, world * 1024, 192, device=dev)[:, 1024:2048].contiguous())))
