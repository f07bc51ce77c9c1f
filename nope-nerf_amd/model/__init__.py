"""`import model as mdl` -- the reference's public surface (reference model/__init__.py:1-10), backed by the
MI355X-native render path in `nnr` (libnnr.so)."""
from model.checkpoints import CheckpointIO
from model.network import nope_nerf
from model.training import Trainer
from model.rendering import Renderer
from model.config import get_model
from model.official_nerf import OfficialStaticNerf
from model.poses import LearnPose
from model.intrinsics import LearnFocal
from model.eval_pose_one_epoch import Trainer_pose
from model.distortions import Learn_Distortion
