"""Per-image affine correction (scale, shift) of the mono-depth maps; parameters and semantics of reference
model/distortions.py:4-26: scale is floored at the *constant* 0.01 (no gradient below it) and the last camera's
scale is pinned to 1 when distortion.fix_scaleN is set."""
import torch
import torch.nn as nn


class Learn_Distortion(nn.Module):
    def __init__(self, num_cams, learn_scale, learn_shift, cfg):
        super().__init__()
        self.global_scales = nn.Parameter(torch.ones(num_cams, 1), requires_grad=learn_scale)
        self.global_shifts = nn.Parameter(torch.zeros(num_cams, 1), requires_grad=learn_shift)
        self.fix_scaleN = cfg['distortion']['fix_scaleN']
        self.num_cams = num_cams

    def forward(self, cam_id):
        cid = int(cam_id)
        scale = self.global_scales[cam_id]
        # value AND gradient of the reference's `if scale < 0.01: scale = tensor(0.01)` without the device->host sync
        scale = torch.where(scale < 0.01, torch.full_like(scale, 0.01), scale)
        if self.fix_scaleN and cid == self.num_cams - 1:
            scale = torch.ones_like(scale).detach()
        return scale, self.global_shifts[cam_id]
