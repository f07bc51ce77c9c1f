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
    (4096, 128, True): {"mlp_fwd": 1.7, "mlp_dgrad": 1.7, "mlp_wgrad": 2.2, "mlp_fwd_infer": 1.0},       # measured 0.78 / 0.75 / 1.02 / 0.45
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
