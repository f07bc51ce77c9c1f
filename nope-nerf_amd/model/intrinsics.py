"""Learnable focal length(s), API of reference model/intrinsics.py:5-70 (off by default: pose.learn_focal False).
The parameter is the focal itself (order 1) or its square root (order 2); gradients reach it through camera_mat,
which the renderer inverts inside the autograd graph."""
import numpy as np
import torch
import torch.nn as nn


class LearnFocal(nn.Module):
    def __init__(self, req_grad, fx_only, order=2, init_focal=None):
        super().__init__()
        if order not in (1, 2):
            raise ValueError('Focal init order need to be 1 or 2')
        self.fx_only, self.order = fx_only, order

        def coeff(f):
            if f is None:
                return torch.tensor(1.0)
            return torch.tensor(np.sqrt(f) if order == 2 else f).float()

        if isinstance(init_focal, list):
            fx0, fy0 = init_focal[0], init_focal[1]
        else:
            fx0 = fy0 = init_focal
        self.fx = nn.Parameter(coeff(fx0), requires_grad=req_grad)
        if not fx_only:
            self.fy = nn.Parameter(coeff(fy0), requires_grad=req_grad)

    def forward(self, i=None):
        fx = self.fx
        fy = self.fx if self.fx_only else self.fy
        if self.order == 2:
            fx, fy = fx ** 2, fy ** 2
        return torch.stack([fx, fy])
