"""`import model as mdl` -- the reference's public surface (same ten names as reference model/__init__.py), backed by the
MI355X-native render path in `nnr` (libnnr.so).  The names are resolved from one table so that the surface is stated once
and checked at import time."""
import importlib

from nnr import parallel as _parallel

_parallel.auto_init()      # under torchrun: bind the GPU of this rank, join the process group (nnr/parallel.py) -- train.py itself never does

_EXPORTS = {
    'CheckpointIO': 'checkpoints', 'nope_nerf': 'network', 'Trainer': 'training', 'Renderer': 'rendering',
    'get_model': 'config', 'OfficialStaticNerf': 'official_nerf', 'LearnPose': 'poses', 'LearnFocal': 'intrinsics',
    'Trainer_pose': 'eval_pose_one_epoch', 'Learn_Distortion': 'distortions',
}
__all__ = sorted(_EXPORTS)

for _name, _module in _EXPORTS.items():
    globals()[_name] = getattr(importlib.import_module('model.' + _module), _name)
del _name, _module
