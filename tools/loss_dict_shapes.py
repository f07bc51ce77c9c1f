"""What Trainer.train_step returns, entry by entry (type, shape, dtype, device), with and without the per-image block:  python tools/loss_dict_shapes.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch
import bench

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    for aux in (False, True):
        trainer, net = bench.build_trainer(dev, 1, aux, False, 256, 64)
        data = bench.synthetic_batch(dev)
        ld = trainer.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
        print("aux", aux, {k: (type(v).__name__, tuple(v.shape), str(v.dtype), str(v.device)) if torch.is_tensor(v) else type(v).__name__ for k, v in ld.items()})
        print("   item():", {k: round(v.item(), 6) for k, v in ld.items() if torch.is_tensor(v) and v.numel() == 1})
        for k, v in ld.items():      # the early host copy answers exactly what the device tensor holds
            if torch.is_tensor(v) and v.numel() == 1:
                assert v.item() == torch.Tensor.item(v), (k, v.item(), torch.Tensor.item(v))
        for i in range(3):           # and keeps doing so over further steps (slot rotation)
            ld2 = trainer.train_step(data, it=2 + i, epoch=0, scheduling_start=10000, render_path=None)
            assert all(v.item() == torch.Tensor.item(v) for v in ld2.values() if torch.is_tensor(v) and v.numel() == 1)
        trainer.flush_nan_check()
        print("   early host copies == device values over 4 steps; NaN flag clean")
