"""Extract_Images -- novel-view rendering for vis/render.py; API of reference model/extracting_images.py:14-124.
`output_geo` (phong geometry visualisation) is outside the HIP hot path and is skipped with a warning."""
import logging
import os

import numpy as np

from model import imaging

logger_py = logging.getLogger(__name__)


class Extract_Images(object):
    def __init__(self, renderer, cfg, use_learnt_poses=True, use_learnt_focal=True, device=None, render_type=None):
        self.points_batch_size = 100000
        self.renderer = renderer
        self.resolution = cfg['extract_images']['resolution']
        self.device = device
        self.use_learnt_poses, self.use_learnt_focal = use_learnt_poses, use_learnt_focal
        self.render_type = render_type
        self._warned = False

    def process_data_dict(self, data):
        return data.get('img.camera_mat').to(self.device), data.get('img.scale_mat').to(self.device), data.get('img.idx')

    def generate_images(self, data, render_dir, c2ws, fxfy, it, output_geo):
        self.renderer.eval()
        camera_mat, scale_mat, img_idx = self.process_data_dict(data)
        img_idx = int(img_idx)
        if self.use_learnt_poses:
            world_mat = imaging.inverse_pose(c2ws[img_idx])
        if self.use_learnt_focal:
            camera_mat = imaging.camera_from_focal(fxfy, self.device)
        rgb, depth_out = imaging.render_full_image(self.renderer, self.resolution, camera_mat, world_mat, scale_mat,
                                                   self.render_type, self.device, self.points_batch_size, it)
        img_out = (rgb.cpu().numpy() * 255).astype(np.uint8)
        if output_geo and not self._warned:
            logger_py.warning("output_geo: the phong geometry renderer is outside the HIP hot path; no geo_out images")
            self._warned = True
        img_dir, depth_dir = os.path.join(render_dir, 'img_out'), os.path.join(render_dir, 'depth_out')
        os.makedirs(img_dir, exist_ok=True)
        os.makedirs(depth_dir, exist_ok=True)
        np.save(os.path.join(depth_dir, '{}.npy'.format(img_idx)), depth_out)
        depth_u8 = imaging.depth_to_u8(depth_out)
        name = str(img_idx).zfill(4) + '.png'
        imaging.save_png(img_out, os.path.join(img_dir, name))
        imaging.save_png(depth_u8, os.path.join(depth_dir, name))
        return {'img': img_out, 'depth': depth_u8, 'geo': None}
