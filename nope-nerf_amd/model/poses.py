"""Learnable per-image camera poses: so(3) vector + translation -> 4x4 camera-to-world.
Same parameters (`r`, `t`, optional frozen `init_c2w`) and call signature as reference model/poses.py:6-33;
once per step, stays in torch -- the kernels receive rays derived from it and return
ray gradients that autograd carries back to r and t."""
import torch
import torch.nn as nn

from model.common import make_c2w
from nnr import camera


class LearnPose(nn.Module):
    def __init__(self, num_cams, learn_R, learn_t, cfg, init_c2w=None):
        super().__init__()
        self.num_cams = num_cams
        self.init_c2w = nn.Parameter(init_c2w, requires_grad=False) if init_c2w is not None else None
        self.r = nn.Parameter(torch.zeros(num_cams, 3), requires_grad=learn_R)
        self.t = nn.Parameter(torch.zeros(num_cams, 3), requires_grad=learn_t)

    def forward(self, cam_id):
        i = int(cam_id)
        if self.r.is_cuda:      # one HIP launch forward, one backward (nnr_se3_exp_*), instead of ~60 ATen kernels
            pose = camera.se3_exp(self.r, self.t, i)
        else:
            pose = make_c2w(self.r[i], self.t[i])
        return pose if self.init_c2w is None else pose @ self.init_c2w[i]   # delta pose on top of the initial one

    def get_t(self):
        return self.t
