"""GPU (-m gpu): the logged loss scalars of a training step are copied to the host right behind the forward (model/training.py:
_EarlyScalar) so that the `.item()` calls of the reference's loop (train.py:211-214) wait for that copy only, not for backward and
optimizer.  They must answer exactly what the device tensors hold, keep doing so when the pinned buffers rotate, fall back to the
device value once their buffer has been reused, and the deferred NaN check must still fire (one step late, or at flush)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _trainer(aux):
    import bench
    dev = torch.device("cuda", 0)
    trainer, net = bench.build_trainer(dev, 1, aux, False, 256, 64)
    return trainer, net, bench.synthetic_batch(dev)


@pytest.mark.parametrize("aux", [False, True])
def test_early_host_copies_equal_the_device_values(aux):
    from model.training import _EarlyScalar
    trainer, _, data = _trainer(aux)
    kept = []
    for i in range(7):
        ld = trainer.train_step(data, it=1 + i, epoch=0, scheduling_start=10000, render_path=None)
        scal = {k: v for k, v in ld.items() if torch.is_tensor(v) and v.numel() == 1 and k not in ("scale", "shift")}
        assert scal and all(isinstance(v, _EarlyScalar) for v in scal.values()), {k: type(v).__name__ for k, v in scal.items()}
        for k, v in scal.items():
            assert v.item() == torch.Tensor.item(v), (i, k)
        assert float(ld["loss"]) == ld["loss"].item() and (ld["loss"] + 0).item() == ld["loss"].item()      # every other use is the device tensor
        kept.append(ld)
    # the first step's scalars: their pinned buffer has been reused since (four buffers in turn) -- the device value answers
    for k, v in kept[0].items():
        if isinstance(v, _EarlyScalar):
            assert v.item() == torch.Tensor.item(v)
    trainer.flush_nan_check()


def test_nan_loss_is_still_caught_one_step_late():
    trainer, net, data = _trainer(False)
    trainer.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
    trainer.flush_nan_check()
    with torch.no_grad():
        net.fc_rgb.weight.fill_(float("nan"))
    from nnr import ops
    ops.invalidate_packed_weights()
    trainer.train_step(data, it=2, epoch=0, scheduling_start=10000, render_path=None)      # produces the NaN; reported by the NEXT call
    with pytest.raises(FloatingPointError):
        trainer.train_step(data, it=3, epoch=0, scheduling_start=10000, render_path=None)
        trainer.flush_nan_check()
