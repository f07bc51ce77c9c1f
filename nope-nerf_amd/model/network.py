"""nope_nerf -- the object `get_model` returns; thin wrapper that picks the per-ray mono-depth values and calls the
renderer (reference model/network.py:8-33)."""
import torch
import torch.nn as nn

from nnr import camera


def nearest_source_index(dst_idx, dst_size, src_size):
    """Source index F.interpolate(mode='nearest') reads for destination index `dst_idx` along one axis:
    floor(dst * (src/dst_size)) computed in float32 like ATen's nearest kernel, clamped to the source."""
    scale = torch.tensor(float(src_size) / float(dst_size), dtype=torch.float32, device=dst_idx.device)
    src = torch.floor(dst_idx.to(torch.float32) * scale).long()
    return torch.clamp(src, max=src_size - 1)


class nope_nerf(nn.Module):
    def __init__(self, cfg, renderer, depth_estimator=None, device=None, **kwargs):
        super().__init__()
        self.renderer = renderer.to(device)
        self.depth_estimator = depth_estimator.to(device) if depth_estimator is not None else None
        self.device = device

    def forward(self, p, ray_idx, camera_mat, world_mat, scale_mat, rendering_technique, it=0, eval_mode=False,
                depth_img=None, add_noise=True, img_size=None, depth_affine=None, rays=None):
        """`depth_affine` = (scale, shift, shift_first) is an extension for the trainer: `depth_img` is then the RAW mono-depth
        map and the frame's distortion is applied to the gathered values only (the same numbers as distorting the map first).
        `rays` (another trainer extension) = the per-ray tensors of nnr.camera.step_rays, which has already gathered the depths
        and generated the rays in its one launch: nothing is left to do here."""
        depth = None
        if rays is not None:
            return self.renderer(p, None, camera_mat, world_mat, scale_mat, rendering_technique, eval_=eval_mode, it=it,
                                 add_noise=add_noise, rays=rays)
        if rendering_technique == 'nope_nerf':
            # The reference nearest-resizes the whole depth map to the image size every step and then gathers R
            # values (network.py:22-24).  Gather-then-nothing is the same thing: index the source pixel directly.
            h, w = img_size
            if depth_img.is_cuda and depth_affine is not None:
                depth = camera.depth_gather_affine(depth_img, ray_idx, depth_affine[0], depth_affine[1], h, w, depth_affine[2])
            elif depth_img.is_cuda:
                depth = camera.depth_gather(depth_img, ray_idx, h, w)               # one launch (nnr_depth_gather_*)
            else:
                hd, wd = depth_img.shape[-2:]
                ys = nearest_source_index(torch.div(ray_idx, w, rounding_mode='floor'), h, hd)
                xs = nearest_source_index(ray_idx % w, w, wd)
                depth = depth_img[0, 0][ys, xs].view(1, -1, 1)
                if depth_affine is not None:
                    sc, sh, first = depth_affine
                    depth = (depth + sh) * sc if first else depth * sc + sh
        return self.renderer(p, depth, camera_mat, world_mat, scale_mat, rendering_technique, eval_=eval_mode, it=it,
                             add_noise=add_noise)
