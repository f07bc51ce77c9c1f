"""Trainer_pose -- test-time pose optimisation step: network frozen in eval mode, no jitter, MSE on rgb, only the pose
optimiser steps (reference model/eval_pose_one_epoch.py:10-98).  Needs d loss / d rays through the fused kernel with the
weights held fixed."""
import logging

import torch

from model import imaging
from model.common import arange_pixels
from model.losses import Loss_Eval

logger_py = logging.getLogger(__name__)


def _pixel_batch(img, n_points):
    """n_points random pixels of the (b,3,h,w) frame: (ray_idx, rgb_gt (b,n,3), p (b,n,2) in [-1,1])."""
    b, _, h, w = img.shape
    ray_idx = torch.randperm(h * w, device=img.device)[:n_points]
    colours = img.view(b, 3, h * w).permute(0, 2, 1)[:, ray_idx]
    return ray_idx, colours, arange_pixels((h, w), b, device=img.device)[1][:, ray_idx]


class Trainer_pose(object):
    def __init__(self, model, cfg, device=None, optimizer_pose=None, pose_param_net=None, focal_net=None, **kwargs):
        self.loss = Loss_Eval()
        self.rendering_technique, self.n_points = cfg['type'], cfg['n_points']
        self.model, self.device = model, device
        self.pose_param_net, self.optimizer_pose, self.focal_net = pose_param_net, optimizer_pose, focal_net

    def train_step(self, data, it=100000):
        """One Adam step on the pose table only; the radiance field (and a learned focal) stay frozen in eval mode."""
        self.model.eval()
        if self.focal_net is not None:
            self.focal_net.eval()
        self.pose_param_net.train()
        self.optimizer_pose.zero_grad()
        losses = self.compute_loss(data, it=it)
        losses['loss'].backward()
        self.optimizer_pose.step()
        return losses

    def process_data_dict(self, data):
        on = lambda key: data.get(key).to(self.device)
        img = on('img')
        b, _, h, w = img.shape
        depth_img = data.get('img.depth', torch.ones(b, h, w)).unsqueeze(1).to(self.device)     # all ones: no ray masked
        return img, depth_img, on('img.camera_mat'), on('img.scale_mat'), on('img.idx')

    def compute_loss(self, data, eval_mode=False, it=100000):
        img, depth_img, camera_mat, scale_mat, img_idx = self.process_data_dict(data)
        if self.focal_net is not None:
            camera_mat = imaging.camera_from_focal(self.focal_net(0), self.device)
        world_mat = imaging.inverse_pose(self.pose_param_net(img_idx))
        ray_idx, rgb_gt, p = _pixel_batch(img, self.n_points)
        out = self.model(p, ray_idx, camera_mat, world_mat, scale_mat, self.rendering_technique, it=it, eval_mode=True,
                         depth_img=depth_img, add_noise=False, img_size=tuple(img.shape[-2:]))
        return self.loss(out['rgb'], rgb_gt)
