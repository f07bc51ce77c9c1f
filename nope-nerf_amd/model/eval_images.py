"""Eval_Images -- full-image rendering + PSNR / SSIM / LPIPS for evaluation/eval.py; constructor, `eval_images`
signature and the returned dictionary follow reference model/eval_images.py:16-137.  The render loop runs the
forward-only fused HIP kernel; SSIM is the Gaussian-window metric of third_party/pytorch_ssim restated in model/imaging.py
(pinned to the reference function's values), LPIPS is whatever callable the caller passes."""
import logging
import os

import torch
import torch.nn.functional as F
from tqdm import tqdm

from model import imaging
from model.common import mse2psnr

logger_py = logging.getLogger(__name__)


class Eval_Images(object):
    def __init__(self, renderer, cfg, points_batch_size=100000, use_learnt_poses=True, use_learnt_focal=True, device=None,
                 render_type=None, c2ws=None, img_list=None):
        self.points_batch_size = points_batch_size
        self.renderer = renderer
        self.resolution = cfg['extract_images']['resolution']
        self.device = device
        self.use_learnt_poses, self.use_learnt_focal = use_learnt_poses, use_learnt_focal
        self.render_type = render_type
        self.c2ws, self.img_list = c2ws, img_list

    def process_data_dict(self, data):
        img = data.get('img').to(self.device)
        b, _, h, w = img.shape
        return (img, data.get('img.depth', torch.ones(b, h, w)), data.get('img.camera_mat').to(self.device),
                data.get('img.scale_mat').to(self.device), data.get('img.idx'))

    def eval_images(self, data, render_dir, fxfy, lpips_vgg_fn, logger, min_depth=0.1, max_depth=20, it=0):
        self.renderer.eval()
        img_gt, depth_gt, camera_mat, scale_mat, img_idx = self.process_data_dict(data)
        img_idx = int(img_idx)
        img_gt = img_gt.squeeze(0).permute(1, 2, 0)
        depth_gt = depth_gt.squeeze(0).numpy()
        mask = (depth_gt > min_depth) * (depth_gt < max_depth)
        if self.use_learnt_poses:
            world_mat = imaging.inverse_pose(self.c2ws[img_idx])
        if self.use_learnt_focal:
            camera_mat = imaging.camera_from_focal(fxfy, self.device)
        img_out, depth_out = imaging.render_full_image(self.renderer, self.resolution, camera_mat, world_mat, scale_mat,
                                                       self.render_type, self.device, self.points_batch_size, it)
        mse = F.mse_loss(img_out, img_gt).item()
        psnr = mse2psnr(mse)
        chw = lambda t: t.permute(2, 0, 1).unsqueeze(0).contiguous()
        ssim = imaging.ssim_gaussian(chw(img_out), chw(img_gt)).item()
        lpips_loss = lpips_vgg_fn(chw(img_out), chw(img_gt), normalize=True).item()
        tqdm.write('{0:4d} img: PSNR: {1:.2f}, SSIM: {2:.2f},  LPIPS {3:.2f}'.format(img_idx, psnr, ssim, lpips_loss))

        depth_out = imaging.resize_nearest(depth_out, depth_gt.shape[:2])
        dirs = {k: os.path.join(render_dir, k) for k in ('img_out', 'depth_out', 'img_gt_out')}
        for d in dirs.values():
            os.makedirs(d, exist_ok=True)
        depth_out = imaging.depth_to_u8(depth_out)
        img_u8 = (img_out.cpu().numpy() * 255).astype('uint8')
        gt_u8 = (img_gt.cpu().numpy() * 255).astype('uint8')
        name = str(img_idx).zfill(4) + '.png'
        imaging.save_png(img_u8, os.path.join(dirs['img_out'], name))
        imaging.save_png(depth_out, os.path.join(dirs['depth_out'], name))
        imaging.save_png(gt_u8, os.path.join(dirs['img_gt_out'], name))
        depth_out, depth_gt = depth_out[mask], depth_gt[mask]
        return {'img': img_u8, 'depth': depth_out, 'mse': mse, 'psnr': psnr, 'ssim': ssim, 'lpips': lpips_loss,
                'depth_pred': depth_out, 'depth_gt': depth_gt}
