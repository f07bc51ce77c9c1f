"""CPU: the oracle reproduces every golden vector minted from the real reference
(oracle/gen_golden.py).  This is what pins oracle/nerf_oracle.py."""
import numpy as np
import pytest
import torch

import golden_util as gu


@pytest.mark.parametrize("name", gu.GRAD_CASES + gu.EVAL_CASES)
def test_oracle_matches_reference_golden(name):
    case = gu.load_case(name)
    out, grads = gu.run_oracle(case)
    for k in ("rgb", "depth_pred", "depth_gt", "alpha", "z_vals"):
        ref = case["out." + k]
        got = out[k].detach().numpy()
        assert got.shape == ref.shape, k
        # same code, same torch build -> bit-exact here; 1e-6 leaves room for a different BLAS on the GPU box
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-6, err_msg=k)
    for k, (kind, ref, norm) in gu.golden_grads(case).items():
        gu.compare_grad(k, grads[k], kind, ref, norm, 2e-5)


def test_invariants():
    """De-facto invariants of the reference path."""
    case = gu.load_case("uniform_distalpha_masked_d128")
    out, _ = gu.run_oracle(case)
    alpha = out["alpha"]
    assert torch.all(alpha[:, -1] == 1.0)                      # model/rendering.py:128
    z = out["z_vals"]
    assert torch.all(z[:, 1:] >= z[:, :-1])                    # jittered samples stay ordered (:186-190)
    assert torch.all((out["rgb"] > 0) & (out["rgb"] < 1))
    import nerf_oracle as orc
    r = torch.zeros(3, requires_grad=True)
    R = orc.so3_exp(r)
    assert torch.allclose(R, torch.eye(3))
    R.sum().backward()
    assert torch.isfinite(r.grad).all()                        # +1e-15 keeps Exp differentiable at 0
