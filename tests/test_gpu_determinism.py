"""GPU (-m gpu): bit-reproducibility of training.  Every reduction on the gradient path has a fixed order -- the weight-gradient
partials are summed slot by slot, the pose / distortion sums by one workgroup in lane-tree + wave order (nnr_camera.hip), the
per-image losses through per-block partials added in block order and 64-bit fixed-point scatters (nnr_aux.hip) -- so two runs
from the same seed must end in bit-identical parameters, with or without the first-phase per-image losses.  Round 3: no float atomic
is left on any gradient path (depth-gather backward: one owner ray per depth pixel; per-image block: one owner grid point per depth
pixel; stand-alone point-cloud backward: a gather per destination), -munsafe-fp-atomics is gone from the build, and the `coarse` cases
run with mono-depth maps 4x coarser than the image, where those scatters really collide."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(aux, bf16, steps=25, coarse=False):
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda")
    old = (bench.IMG_H, bench.IMG_W)
    bench.IMG_H, bench.IMG_W = 120, 160
    try:
        trainer, net = bench.build_trainer(dev, 1, aux=aux, bf16=bf16, rays_per_gpu=512, n_samples=64)
        # coarse: mono-depth maps 4x coarser than the image (what DPT delivers) -- several rays then share a depth pixel in the gather's
        # backward, several grid points one depth pixel in the per-image block -- and the per-image block forced onto its torch expression
        # (with_ssim: the fused block does not cover it), whose point-cloud term goes through nnr_pc_nearest / nnr_pc_error_bwd
        data = bench.synthetic_batch(dev, depth_hw=(bench.IMG_H // 4, bench.IMG_W // 4) if coarse else None)
        if coarse and aux:
            trainer.loss.cfg = dict(trainer.loss.cfg, with_ssim=True)
        torch.manual_seed(1234)
        torch.cuda.manual_seed(1234)
        losses = []
        for i in range(steps):
            ld = trainer.train_step(data, it=i, epoch=0, scheduling_start=10000, render_path=None)
            losses.append(ld['loss'].detach().clone())
        trainer.flush_nan_check()
        torch.cuda.synchronize()
        params = [p.detach().clone() for m in (net, trainer.pose_param_net, trainer.distortion_net) for p in m.parameters()]
        return torch.stack(losses), params
    finally:
        bench.IMG_H, bench.IMG_W = old


@pytest.mark.parametrize("aux,bf16,coarse", [(False, False, False), (True, False, False), (False, True, False), (False, False, True),
                                             (True, False, True)])
def test_25_training_steps_twice_are_bit_identical(aux, bf16, coarse):
    l1, p1 = _run(aux, bf16, coarse=coarse)
    l2, p2 = _run(aux, bf16, coarse=coarse)
    assert torch.equal(l1, l2), (l1 - l2).abs().max()
    assert bool(torch.isfinite(l1).all()) and float(l1[-1]) < float(l1[0])     # it trains
    for a, b in zip(p1, p2):
        assert torch.equal(a, b), float((a - b).abs().max())
