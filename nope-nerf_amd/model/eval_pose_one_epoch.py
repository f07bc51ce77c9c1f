"""Trainer_pose -- test-time pose optimisation step: network frozen in eval mode, no jitter, MSE on rgb, only the pose
optimiser steps (reference model/eval_pose_one_epoch.py:10-98).  Needs d loss / d rays through the fused kernel with the
weights held fixed."""
import logging

import torch

from nnr import camera
from model.common import arange_pixels
from model.losses import Loss_Eval

logger_py = logging.getLogger(__name__)


class Trainer_pose(object):
    def __init__(self, model, cfg, device=None, optimizer_pose=None, pose_param_net=None, focal_net=None, **kwargs):
        self.model, self.device = model, device
        self.optimizer_pose, self.pose_param_net, self.focal_net = optimizer_pose, pose_param_net, focal_net
        self.n_points = cfg['n_points']
        self.rendering_technique = cfg['type']
        self.loss = Loss_Eval()

    def train_step(self, data, it=100000):
        self.model.eval()
        self.pose_param_net.train()
        self.optimizer_pose.zero_grad()
        if self.focal_net is not None:
            self.focal_net.eval()
        loss_dict = self.compute_loss(data, it=it)
        loss_dict['loss'].backward()
        self.optimizer_pose.step()
        return loss_dict

    def process_data_dict(self, data):
        dev = self.device
        img = data.get('img').to(dev)
        b, _, h, w = img.shape
        depth_img = data.get('img.depth', torch.ones(b, h, w)).unsqueeze(1).to(dev)
        return (img, depth_img, data.get('img.camera_mat').to(dev), data.get('img.scale_mat').to(dev),
                data.get('img.idx').to(dev))

    def compute_loss(self, data, eval_mode=False, it=100000):
        img, depth_img, camera_mat, scale_mat, img_idx = self.process_data_dict(data)
        dev = self.device
        b, _, h, w = img.shape
        c2w = self.pose_param_net(img_idx)
        world_mat = (camera.inverse4(c2w) if c2w.is_cuda else torch.inverse(c2w)).unsqueeze(0)
        if self.focal_net is not None:
            fxfy = self.focal_net(0)
            camera_mat = torch.zeros(4, 4, device=dev)
            camera_mat[0, 0], camera_mat[1, 1], camera_mat[2, 2], camera_mat[3, 3] = fxfy[0], -fxfy[1], -1.0, 1.0
            camera_mat = camera_mat.unsqueeze(0)
        ray_idx = torch.randperm(h * w, device=dev)[:self.n_points]
        rgb_gt = img.view(b, 3, h * w).permute(0, 2, 1)[:, ray_idx]
        p = arange_pixels((h, w), b, device=dev)[1][:, ray_idx]
        out = self.model(p, ray_idx, camera_mat, world_mat, scale_mat, self.rendering_technique, it=it, eval_mode=True,
                         depth_img=depth_img, add_noise=False, img_size=(h, w))
        return self.loss(out['rgb'], rgb_gt)
