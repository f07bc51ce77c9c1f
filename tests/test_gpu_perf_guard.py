"""GPU (-m gpu): coarse performance guards.  Parity tests cannot see a kernel that computes the right numbers ten times too slowly --
it has happened twice: a launch sized for one workgroup (D = 128 weight gradient) and a `#pragma unroll` loop that
silently stopped unrolling, putting the fp32 input-gradient kernel's register arrays into scratch memory (8x slower, every parity test
green; csrc/build.py now checks the compiler's scratch report as well).  Bounds are ~2x the measured durations on an MI355X."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# ms per launch: (R, N, bf16) -> kernel -> bound
BOUNDS = {
    (1024, 192, False): {"mlp_fwd": 3.6, "mlp_dgrad": 3.3, "mlp_wgrad": 3.2, "mlp_fwd_infer": 3.0},      # measured 1.82 / 1.62 / 1.58 / 1.46
    (4096, 128, True): {"mlp_fwd": 1.7, "mlp_dgrad": 1.7, "mlp_wgrad": 1.5, "mlp_fwd_infer": 1.0},       # measured 0.78 / 0.75 / 0.87-0.94 / 0.45
    (1000, 100, False): {"mlp_fwd": 2.6, "mlp_dgrad": 2.4, "mlp_wgrad": 2.4, "mlp_fwd_infer": 2.2},      # flat decomposition (N % 32 != 0)
}


@pytest.mark.parametrize("shape", sorted(BOUNDS))
def test_mlp_kernels_are_not_an_order_of_magnitude_off(shape):
    import bench
    import model as mdl
    R, N, bf16 = shape
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    net = mdl.OfficialStaticNerf(bench.full_cfg(R, bf16=bf16, n_samples=N)).to(dev)
    out = bench.kernel_roofline(net, dev, reps=4, bf16=bf16, rays=R, n_samples=N)
    for k, bound in BOUNDS[shape].items():
        ms = out["kernels"][k]["ms"]
        assert ms <= bound, (shape, k, ms, bound)


def test_in_step_kernel_timing_hooks():
    """nnr_prof_begin / nnr_prof_end (what bench.py's roofline block is based on): every launch of a main MLP kernel between the two
    calls is timed with HIP events on its launch stream; counts and plausibility of the means, and the error paths."""
    import ctypes as C

    import bench
    from nnr import lib as L
    lib = L.load()
    dev = torch.device("cuda", 0)
    trainer, net = bench.build_trainer(dev, 1, False, False, 256, 64)
    data = bench.synthetic_batch(dev)
    for i in range(2):
        trainer.train_step(data, it=i, epoch=0, scheduling_start=10000, render_path=None)
    ms, n = (C.c_float * 4)(), (C.c_int32 * 4)()
    assert lib.nnr_prof_end(ms, n) != 0                       # not started
    L.check(lib.nnr_prof_begin(3), "nnr_prof_begin")
    assert lib.nnr_prof_begin(3) != 0                         # already on
    for i in range(5):                                        # more launches than the capacity: the surplus is simply not timed
        trainer.train_step(data, it=2 + i, epoch=0, scheduling_start=10000, render_path=None)
    L.check(lib.nnr_prof_end(ms, n), "nnr_prof_end")
    assert list(n) == [3, 3, 3, 0], list(n)                   # training forward, input gradient, weight gradient; no inference forward
    assert all(0.005 < ms[k] < 5.0 for k in range(3)), list(ms)
    trainer.flush_nan_check()
