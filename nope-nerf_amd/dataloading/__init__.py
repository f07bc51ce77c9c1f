"""`import dataloading as dl` -- the reference's data side (dataloading/__init__.py:1-2): `get_dataloader`, `load_config`.
Reads the same on-disk scene layout (images[/_F], poses_bounds.npy, <depth_net>/depth_*.npz, depth/*.png, gt_poses.npz,
intrinsics.npz) with PIL instead of imageio / cv2, and can keep a whole scene resident in HBM (`dataloading.resident`)."""
from dataloading.configloading import load_config  # noqa: F401
from dataloading.dataloading import get_dataloader  # noqa: F401
