"""CheckpointIO -- one torch.save file per module group, same on-disk format as reference model/checkpoints.py
({registered name: state_dict, ...scalars}); files interchange with the reference because every module here keeps
the reference's parameter names and shapes."""
import datetime
import os
import shutil
import urllib.parse

import torch
from torch.utils import model_zoo


# Callables run before every CheckpointIO.save (model.Trainer registers its deferred NaN check here: the training step reads the
# loss's NaN flag one step late to stay free of host syncs, so a NaN in the very last step before a save would otherwise be
# written to model.pt unnoticed -- the reference stops on the NaN before its backward, reference model/losses.py:204).
PRE_SAVE_HOOKS = []


def is_url(url):
    return urllib.parse.urlparse(url).scheme in ('http', 'https')


class CheckpointIO(object):
    def __init__(self, checkpoint_dir='./chkpts', **kwargs):
        self.module_dict = kwargs
        self.checkpoint_dir = checkpoint_dir
        os.makedirs(checkpoint_dir, exist_ok=True)

    def register_modules(self, **kwargs):
        self.module_dict.update(kwargs)

    def _path(self, filename):
        return filename if os.path.isabs(filename) else os.path.join(self.checkpoint_dir, filename)

    def save(self, filename, **kwargs):
        for hook in list(PRE_SAVE_HOOKS):
            hook()
        from nnr import parallel
        if not parallel.is_writer():      # data parallel: every rank holds the same parameters and optimiser state; rank 0 writes them
            return
        blob = dict(kwargs)
        blob.update({name: mod.state_dict() for name, mod in self.module_dict.items()})
        torch.save(blob, self._path(filename))

    def backup_model_best(self, filename, **kwargs):
        from nnr import parallel
        if not parallel.is_writer():
            return
        src = self._path(filename)
        if os.path.exists(src):
            dst = os.path.join(self.checkpoint_dir, 'backup_model_best')
            os.makedirs(dst, exist_ok=True)
            shutil.copy(src, os.path.join(dst, '%s.pt' % datetime.datetime.now().timestamp()))

    def load(self, filename, device=None, load_model_only=False):
        if is_url(filename):
            return self.load_url(filename)
        return self.load_file(filename, device, load_model_only)

    def load_file(self, filename, device=None, load_model_only=False):
        path = self._path(filename)
        if not os.path.exists(path):
            raise FileExistsError          # sic: train.py:64-67 resumes on exactly this exception type
        print(path)
        print('=> Loading checkpoint from local file...')
        # weights_only=False: these are trusted local checkpoints, and train.py stores numpy scalars in them
        # (`loss_val_best = np.array(psnr_window).mean()`), which torch >= 2.6's default weights_only=True refuses to unpickle
        # with an UnpicklingError -- not the FileExistsError train.py:64-67 knows how to handle.  The reference's torch 1.7
        # has no such restriction.
        blob = torch.load(path, map_location=device, weights_only=False)
        if load_model_only:
            blob = {'model': blob['model']}
        return self.parse_state_dict(blob)

    def load_url(self, url):
        print(url)
        print('=> Loading checkpoint from url...')
        return self.parse_state_dict(model_zoo.load_url(url, progress=True, check_hash=False, weights_only=False))

    def parse_state_dict(self, state_dict):
        for name, mod in self.module_dict.items():
            if name in state_dict:
                mod.load_state_dict(state_dict[name])
            else:
                print('Warning: Could not find %s in checkpoint!' % name)
        return {k: v for k, v in state_dict.items() if k not in self.module_dict}
