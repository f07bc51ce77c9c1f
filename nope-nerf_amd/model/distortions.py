"""Per-image affine correction (scale, shift) of the mono-depth maps; parameters and semantics of reference
model/distortions.py:4-26: scale is floored at the *constant* 0.01 (no gradient below it) and the last camera's
scale is pinned to 1 when distortion.fix_scaleN is set."""
import torch
from torch import nn

_SCALE_FLOOR = 0.01


class Learn_Distortion(nn.Module):
    """state_dict keys: global_scales, global_shifts -- both (num_cams, 1)."""

    def __init__(self, num_cams, learn_scale, learn_shift, cfg):
        super().__init__()
        self.num_cams = num_cams
        self.fix_scaleN = cfg['distortion']['fix_scaleN']
        self.global_shifts = nn.Parameter(torch.zeros(num_cams, 1), requires_grad=learn_shift)
        self.global_scales = nn.Parameter(torch.ones(num_cams, 1), requires_grad=learn_scale)

    def forward(self, cam_id):
        """-> (scale, shift) of one camera, (1,) each."""
        shift = self.global_shifts[cam_id]
        if self.fix_scaleN and int(cam_id) == self.num_cams - 1:
            return torch.ones_like(shift), shift                      # the gauge: the last view's depth scale is 1
        raw = self.global_scales[cam_id]
        # value AND gradient of the reference's `if scale < 0.01: scale = tensor(0.01)` without its device->host sync
        # (below the floor the parameter receives a ZERO gradient and Adam keeps moving it by momentum -- what the reference does under the torch it
        # pins, 1.7, whose zero_grad() zero-fills; under torch >= 2.0 its gradient would be None and Adam would skip the tensor for that step)
        return torch.where(raw < _SCALE_FLOOR, torch.full_like(raw, _SCALE_FLOOR), raw), shift
