"""One scene as arrays in host memory + per-index access (reference dataloading/dataset.py:13-227).  Attribute and key names are
the reference's: callers read .N_imgs, .c2ws, .c2ws_colmap, .K, .H, .W, .focal, .i_train, .i_test, .img_list, .dpt_depth."""
import logging
import os
import random

import numpy as np
import torch

from dataloading.common import _load_data, load_depths_npz, load_gt_depths, recenter_poses, spherify_poses

logger = logging.getLogger(__name__)

_BD_FACTOR = 0.75                    # nearest scene bound lands at 1 / 0.75 after rescaling (dataset.py:59-63)
_OPENCV_TO_OPENGL = np.diag([1.0, -1.0, -1.0, 1.0]).astype(np.float32)
_IDENTITY = np.eye(4, dtype=np.float32)


def _normalised_llff_poses(poses, bds, spherify):
    """LLFF pose block (3,5,n) -> (n,3,5) float32 in the render path's axes [right, up, back], rescaled so that the closest
    depth bound is 1/0.75, recentred on the average camera and optionally spherified (dataset.py:55-68)."""
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    sc = 1. / (bds.min() * _BD_FACTOR)
    poses[:, :3, 3] *= sc
    bds *= sc
    poses = recenter_poses(poses)
    if spherify:
        poses, _, bds = spherify_poses(poses, bds)
    return poses.astype(np.float32)


class DataField(object):
    def __init__(self, model_path, transform=None, with_camera=False, with_depth=False, use_DPT=False, scene_name=[' '],
                 mode='train', spherify=False, load_ref_img=False, customized_poses=False, customized_focal=False,
                 resize_factor=2, depth_net='dpt', crop_size=0, random_ref=False, norm_depth=False, load_colmap_poses=True,
                 sample_rate=8, **kwargs):
        """Arguments as reference dataset.py:14-42.  mode: 'train' | 'eval_trained' | 'render' (training views), 'eval' (every
        sample_rate-th view, starting at sample_rate // 2), 'all'."""
        if use_DPT:
            raise NotImplementedError("depth.type 'DPT' runs the DPT network inside the data loader; precompute the maps with "
                                      "preprocess/dpt_depth.py (writes <scene>/dpt/depth_*.npz) and set depth.type: None")
        self.transform, self.with_camera, self.with_depth, self.use_DPT = transform, with_camera, with_depth, use_DPT
        self.mode, self.ref_img, self.random_ref, self.sample_rate = mode, load_ref_img, random_ref, sample_rate

        load_dir = os.path.join(model_path, scene_name[0])
        if crop_size != 0:
            depth_net = depth_net + '_' + str(crop_size)
        poses, bds, imgs, img_names, crop_ratio, focal_crop_factor = _load_data(
            load_dir, factor=resize_factor, crop_size=crop_size, load_colmap_poses=load_colmap_poses)
        c2ws_colmap = None
        if load_colmap_poses:
            llff = _normalised_llff_poses(poses, bds, spherify)
            self.hwf = llff[:, :3, :]
            focal = llff[0, 2, -1]
            c2ws_colmap = torch.from_numpy(np.concatenate(
                [llff[:, :3, :4], np.broadcast_to(_IDENTITY[3:], (llff.shape[0], 1, 4))], 1))

        imgs = np.transpose(np.moveaxis(imgs, -1, 0).astype(np.float32), (0, 3, 1, 2))    # (n, 3, h, w)
        h, w = imgs.shape[2:]
        if customized_focal:
            K_file = np.load(os.path.join(load_dir, 'intrinsics.npz'))['K'].astype(np.float32)
            div = 1 if resize_factor is None else resize_factor
            fx, fy = K_file[0, 0] / div, K_file[1, 1] / div
        elif load_colmap_poses:
            fx = fy = focal
        else:
            print('No focal provided, use image size as default')
            fx, fy = w, h
        fx, fy = fx / focal_crop_factor, fy / focal_crop_factor
        self.H, self.W, self.focal = h, w, fx
        self.K = np.diag([2 * fx / w, -2 * fy / h, -1, 1]).astype(np.float32)

        ids = np.arange(imgs.shape[0])
        self.i_test = ids[int(sample_rate / 2)::sample_rate]
        self.i_train = np.array([i for i in ids if i not in self.i_test])
        train_names = [img_names[i] for i in self.i_train]
        test_names = [img_names[i] for i in self.i_test]
        print('test set: ', test_names)
        self.N_imgs_train, self.N_imgs_test = len(self.i_train), len(self.i_test)

        if customized_poses:   # ground-truth file is in OpenCV axes (x right, y down, z forward)
            gt = np.load(os.path.join(load_dir, 'gt_poses.npz'))['poses'].astype(np.float32)
            c2ws = torch.from_numpy(gt) @ torch.from_numpy(_OPENCV_TO_OPENGL)
        else:
            c2ws = c2ws_colmap

        if mode in ('train', 'eval_trained', 'render'):
            idx_list, self.img_list = self.i_train, train_names
        elif mode == 'eval':
            idx_list, self.img_list = self.i_test, test_names
        elif mode == 'all':
            idx_list, self.img_list = ids, img_names
        else:
            raise ValueError('unknown mode %r' % (mode,))
        self.imgs = imgs[idx_list]
        self.N_imgs = len(idx_list)
        if c2ws is not None:
            self.c2ws = c2ws[idx_list]
        if load_colmap_poses:
            self.c2ws_colmap = c2ws_colmap[self.i_train]
        # depth maps are always those of the TRAINING views, whatever the mode (dataset.py:146-149)
        self.dpt_depth = load_depths_npz(train_names, os.path.join(load_dir, depth_net), norm=norm_depth)
        if with_depth:
            self.depth = load_gt_depths(train_names, load_dir, crop_ratio=crop_ratio)

    # ------------------------------------------------------------------ per-view access
    def load(self, input_idx_img=None):
        return self.load_field(input_idx_img)

    def load_image(self, idx, data={}):
        data[None] = self.imgs[idx]
        data['idx'] = idx

    def pick_reference(self, idx):
        """A later neighbour at most `random_ref` frames ahead; the last view looks back (dataset.py:170-176)."""
        if not self.random_ref:
            raise ValueError('load_ref_img needs dataloading.random_ref >= 1 (the reference leaves ref_idx undefined otherwise)')
        if idx == self.N_imgs - 1:
            return idx - 1
        return idx + random.randint(1, min(self.random_ref, self.N_imgs - idx - 1))

    def load_ref_img(self, idx, data={}):
        ref_idx = self.pick_reference(idx)
        if self.dpt_depth is not None:
            data['ref_dpts'] = self.dpt_depth[ref_idx]
        if self.with_depth:
            data['ref_depths'] = self.depth[ref_idx]
        data['ref_imgs'] = self.imgs[ref_idx]
        data['ref_idxs'] = ref_idx

    def load_depth(self, idx, data={}):
        data['depth'] = self.depth[idx]

    def load_DPT_depth(self, idx, data={}):
        data['dpt'] = self.dpt_depth[idx]

    def load_camera(self, idx, data={}):
        data['camera_mat'] = self.K
        data['scale_mat'] = _IDENTITY.copy()
        data['idx'] = idx

    def load_field(self, input_idx_img=None):
        idx = 0 if input_idx_img is None else input_idx_img
        data = {}
        if self.mode != 'render':
            self.load_image(idx, data)
            if self.ref_img:
                self.load_ref_img(idx, data)
            if self.with_depth:
                self.load_depth(idx, data)
            if self.dpt_depth is not None:
                self.load_DPT_depth(idx, data)
        if self.with_camera:
            self.load_camera(idx, data)
        return data
