"""cfg -> (loader, fields)  (reference dataloading/dataloading.py:13-159).

Two loaders behind the same call:
  * the reference's: a torch DataLoader over host arrays, default collate, one (1,3,h,w) frame (+ neighbour, depth maps) per
    step -- every step then pays a host-to-device copy of 10-25 MB in Trainer.process_data_dict (model/training.py:165-185);
  * the resident one: the whole scene is uploaded once (a 150-frame 540x960 scene with depth maps is about 1.2 GB of the 288 GB of
    HBM) and every step receives views of the resident tensors, so the `.to(device)` calls of the trainer are no-ops and nothing
    crosses PCIe inside the training loop.  The order of the views still comes from a torch sampler with the same `shuffle` flag,
    and the batches are bit-identical to the host loader's (tests/test_dataloading.py).
Which one: `dataloading.resident` = True / False forces either; ABSENT (every YAML of the reference: train.py, unmodified, with its own
configs) means AUTO -- resident when a GPU is there, the batch size is 1 (every config of the reference) and the scene takes at most
half of the free device memory; otherwise the reference's loader.  The reference's loop then runs at the rate of the step, not of the
host collate + PCIe copy of 10-25 MB per step (profiles/r04/scene_loop_*.json)."""
import logging
import os
import weakref

import numpy as np
import torch
from torch.utils import data

from dataloading.dataset import DataField

logger = logging.getLogger(__name__)


def get_data_fields(cfg, mode='train'):
    dcfg, tcfg = cfg['dataloading'], cfg['training']
    if dcfg['dataset_name'] != 'any':
        raise ValueError("dataset_name %r does not exist (only 'any')" % (dcfg['dataset_name'],))
    field = DataField(
        model_path=dcfg['path'], transform=None, with_camera=True, with_depth=dcfg['with_depth'], scene_name=dcfg['scene'],
        use_DPT=(cfg['depth']['type'] == 'DPT'), mode=mode, spherify=dcfg['spherify'],
        load_ref_img=(tcfg['pc_weight'] != 0.0) or (tcfg['rgb_s_weight'] != 0.0),
        customized_poses=dcfg['customized_poses'], customized_focal=dcfg['customized_focal'],
        resize_factor=dcfg['resize_factor'], depth_net=dcfg['depth_net'], crop_size=dcfg['crop_size'],
        random_ref=dcfg['random_ref'], norm_depth=dcfg['norm_depth'], load_colmap_poses=dcfg['load_colmap_poses'],
        sample_rate=dcfg['sample_rate'])
    return {'img': field}


class OurDataset(data.Dataset):
    """index -> flat dict: the unnamed entry of a field becomes '<field>', the others '<field>.<key>'."""

    def __init__(self, fields, n_views=0, mode='train'):
        self.fields = fields
        print(mode, ': ', n_views, ' views')
        self.n_views = n_views

    def __len__(self):
        return self.n_views

    def __getitem__(self, idx):
        out = {}
        for fname, field in self.fields.items():
            item = field.load(idx)
            if isinstance(item, dict):
                for k, v in item.items():
                    out[fname if k is None else '%s.%s' % (fname, k)] = v
            else:
                out[fname] = item
        return out


class _ViewIndices(data.Dataset):
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return i


class ResidentLoader(object):
    """Iterates like the reference loader at batchsize 1, but the frames / depth maps live on `device` for the whole run."""

    # (device, content key) -> device tensor: train.py builds two loaders of the same scene (train.py:35-36), one upload serves both.  Weak
    # values: the loaders hold the tensors, the cache only finds them -- a scene is freed with its last loader
    _uploaded = weakref.WeakValueDictionary()

    def __init__(self, field, n_views, shuffle, device):
        self.field, self.n_views, self.device = field, n_views, device
        self.order = data.DataLoader(_ViewIndices(n_views), batch_size=1, shuffle=shuffle, num_workers=0)

        def up(a):
            a = np.ascontiguousarray(a)
            if a.nbytes < (1 << 20):
                return torch.from_numpy(a).to(device)
            import hashlib
            key = (str(device), a.shape, str(a.dtype), hashlib.blake2b(a.view(np.uint8).reshape(-1), digest_size=16).hexdigest())   # the whole content (~1 GB/s, once)
            t = ResidentLoader._uploaded.get(key)
            if t is None:
                t = torch.from_numpy(a).to(device)
                ResidentLoader._uploaded[key] = t
            return t
        self.imgs = up(field.imgs) if field.mode != 'render' else None
        if self.imgs is not None:
            self.imgs._nnr_resident = True      # views of it may be cached per frame by their consumers (model.Trainer._resized)
        self.dpt = up(field.dpt_depth) if (field.dpt_depth is not None and field.mode != 'render') else None
        self.depth = up(field.depth) if (field.with_depth and field.mode != 'render') else None
        self.K = up(field.K).unsqueeze(0)
        self.K._nnr_resident = True
        self.eye = torch.eye(4, device=device).unsqueeze(0)
        self.dataset = self.order.dataset

    def __len__(self):
        return self.n_views

    def __iter__(self):
        f = self.field
        for batch in self.order:
            i = int(batch[0])
            idx = torch.tensor([i])
            out = {}
            if f.mode != 'render':
                out['img'] = self.imgs[i:i + 1]
                out['img.idx'] = idx
                if f.ref_img:
                    j = f.pick_reference(i)
                    if self.dpt is not None:
                        out['img.ref_dpts'] = self.dpt[j:j + 1]
                    if self.depth is not None:
                        out['img.ref_depths'] = self.depth[j:j + 1]
                    out['img.ref_imgs'] = self.imgs[j:j + 1]
                    out['img.ref_idxs'] = torch.tensor([j])
                if self.depth is not None:
                    out['img.depth'] = self.depth[i:i + 1]
                if self.dpt is not None:
                    out['img.dpt'] = self.dpt[i:i + 1]
            if f.with_camera:
                out['img.camera_mat'], out['img.scale_mat'], out['img.idx'] = self.K, self.eye, idx
            yield out


def _scene_bytes(field):
    return sum(int(a.nbytes) for a in (getattr(field, 'imgs', None), getattr(field, 'dpt_depth', None),
                                       getattr(field, 'depth', None) if getattr(field, 'with_depth', False) else None) if a is not None)


def _auto_resident(field, dcfg, mode):
    """`dataloading.resident` absent: (use the resident loader?, why not)."""
    if os.environ.get('NNR_RESIDENT', '') in ('0', 'off', 'false'):
        return False, 'NNR_RESIDENT=0'
    if not torch.cuda.is_available():
        return False, 'no GPU'
    if dcfg['batchsize'] != 1:
        return False, 'batchsize %d' % dcfg['batchsize']
    need = _scene_bytes(field)
    # (the device's TOTAL memory from its properties: torch.cuda.mem_get_info() would create a CUDA context in this process, and when the
    # answer is "host loader" that process goes on to fork DataLoader workers -- ADVICE r04)
    total = torch.cuda.get_device_properties(torch.cuda.current_device()).total_memory
    # every local rank uploads its own copy: ranks that SHARE a device (the gloo dry run of nnr.parallel.auto_init) share the budget too
    sharing = max(1, -(-int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1) // max(1, torch.cuda.device_count())))
    if need * sharing > total // 4:
        return False, 'the scene (%.1f GB x %d rank(s) on this device) does not fit in a quarter of the device memory (%.1f GB)' % (
            need / 2 ** 30, sharing, total / 2 ** 30)
    return True, ''


def get_dataloader(cfg, mode='train', shuffle=True, n_views=None):
    """-> (iterable of batches, {'img': DataField}).  mode 'render' with n_views yields that many camera-only batches."""
    dcfg = cfg['dataloading']
    fields = get_data_fields(cfg, mode)
    if not (n_views is not None and mode == 'render'):
        n_views = fields['img'].N_imgs
    resident = dcfg.get('resident', 'auto')
    if resident == 'auto' or resident is None:
        resident, why = _auto_resident(fields['img'], dcfg, mode)
        if not resident:
            logger.info('host loader (%s)', why)
    if resident:
        if dcfg['batchsize'] != 1:
            raise ValueError('dataloading.resident serves one view per step (batchsize 1), as every config of the reference does')
        # dataloading.resident_device names the device (the one the Trainer runs on, if not the current one); default: the CURRENT device --
        # under torchrun `import model` has made that this rank's GPU (nnr.parallel.auto_init)
        device = torch.device(dcfg.get('resident_device') or (('cuda:%d' % torch.cuda.current_device()) if torch.cuda.is_available() else 'cpu'))
        print(mode, ': ', n_views, ' views (resident on %s: %.1f MB)' % (device, _scene_bytes(fields['img']) / 2 ** 20))
        return ResidentLoader(fields['img'], n_views, shuffle, device), fields
    dataset = OurDataset(fields, n_views=n_views, mode=mode)
    loader = data.DataLoader(dataset, batch_size=dcfg['batchsize'], num_workers=dcfg['n_workers'], shuffle=shuffle,
                             pin_memory=torch.cuda.is_available())
    return loader, fields


def collate_remove_none(batch):
    return data.dataloader.default_collate([b for b in batch if b is not None])


def worker_init_fn(worker_id):
    np.random.seed(int.from_bytes(os.urandom(4), byteorder='big') + worker_id)
