"""Golden vectors for the data side (on-disk formats poses_bounds.npy / depth_*.npz, dataloading/common.py:59-141,
286-314 and dataloading/dataset.py:53-227).  Writes the deterministic synthetic scene (tools/scene_writer.py), loads it with the
REFERENCE DataField in several configurations (imageio / cv2 are absent here: both are replaced by minimal PIL-backed readers, the
only two calls the loader makes) and freezes what it returns in tests/golden/scene_field.npz.  Authoring container only:
    python oracle/gen_golden_data.py"""
import hashlib
import importlib.machinery
import os
import shutil
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("NNR_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "scene_field.npz")
sys.path.insert(0, os.path.join(ROOT, "tools"))

SCENE = dict(scene="synthetic", frames=9, size=(48, 64), factor=2, seed=5)
CASES = {   # name -> DataField keyword arguments
    "tanks_train": dict(resize_factor=None, spherify=True, load_ref_img=True, random_ref=1, mode="train"),
    "llff_eval": dict(resize_factor=2, spherify=True, load_ref_img=False, random_ref=1, mode="eval", sample_rate=4),
    "plain_all": dict(resize_factor=None, spherify=False, load_ref_img=True, random_ref=2, mode="all", norm_depth=True),
    "custom": dict(resize_factor=2, spherify=False, customized_poses=True, customized_focal=True, with_depth=True,
                   load_ref_img=True, random_ref=1, mode="train"),
    "cropped": dict(resize_factor=None, spherify=True, crop_size=4, load_ref_img=False, mode="train"),
    "no_colmap": dict(resize_factor=None, load_colmap_poses=False, load_ref_img=False, mode="render"),
}


def stub_readers():
    imageio = types.ModuleType("imageio")
    imageio.imread = lambda f, ignoregamma=False: np.asarray(Image.open(f))
    cv2 = types.ModuleType("cv2")
    cv2.IMREAD_UNCHANGED, cv2.INTER_AREA, cv2.INTER_CUBIC, cv2.INTER_NEAREST = -1, 3, 2, 0     # constants named at import time
    cv2.imread = lambda f, flags=None: np.asarray(Image.open(f))
    for m in (imageio, cv2):
        m.__spec__ = importlib.machinery.ModuleSpec(m.__name__, None)
        sys.modules[m.__name__] = m
    for name in ("torchvision", "torchvision.transforms", "timm", "timm.models", "timm.models.layers"):
        m = MagicMock()
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m


def prepare(tmp):
    import scene_writer
    meta = scene_writer.write_scene(tmp, **SCENE)
    root = os.path.join(tmp, SCENE["scene"])
    shutil.copytree(os.path.join(root, "dpt"), os.path.join(root, "dpt_4"))     # crop_size=4 reads <depth_net>_4/
    return meta


def digest(a):
    """SHA-1 of the raw bytes (dtype + layout included): whole-array bit-exactness without storing the array."""
    a = np.ascontiguousarray(a)
    return np.array(hashlib.sha1(str(a.dtype).encode() + a.tobytes()).hexdigest())


def snapshot(field, picks):
    import random
    out = {"imgs_sha1": digest(field.imgs), "imgs_shape": np.array(field.imgs.shape), "imgs_corner": field.imgs[:, :, :4, :4], "K": field.K, "H": field.H, "W": field.W,
           "focal": np.float64(field.focal), "i_train": field.i_train, "i_test": field.i_test, "N_imgs": field.N_imgs,
           "img_list": np.array(field.img_list)}
    for k in ("c2ws", "c2ws_colmap"):
        if hasattr(field, k) and getattr(field, k) is not None:
            out[k] = getattr(field, k).numpy()
    if hasattr(field, "hwf"):
        out["hwf"] = field.hwf
    if field.dpt_depth is not None:
        out["dpt_sha1"], out["dpt_shape"], out["dpt_corner"] = digest(field.dpt_depth), np.array(field.dpt_depth.shape), field.dpt_depth[:, :4, :4]
    if field.with_depth:
        out["depth_sha1"], out["depth_corner"] = digest(field.depth), field.depth[:, :4, :4]
    random.seed(11)
    for i in picks:
        try:
            d = field.load(i)
        except IndexError:      # mode 'all' / 'eval' index the TRAINING views' depth maps (dataset.py:146-149): frames beyond
            out[f"load{i}.index_error"] = np.int64(1)      # their count cannot be served
            continue
        out[f"load{i}.keys"] = np.array(sorted("" if k is None else k for k in d))
        for k in ("ref_idxs", "idx"):
            if k in d:
                out[f"load{i}.{k}"] = np.int64(d[k])
    return out


def main():
    stub_readers()
    sys.path.insert(0, REF)
    from dataloading.dataset import DataField
    blob = {}
    with tempfile.TemporaryDirectory() as tmp:
        prepare(tmp)
        for name, kw in CASES.items():
            kw = dict(kw)
            field = DataField(tmp, with_camera=True, scene_name=[SCENE["scene"]], use_DPT=False, depth_net="dpt", **kw)
            snap = snapshot(field, picks=(0, field.N_imgs // 2, field.N_imgs - 1))
            blob.update({f"{name}.{k}": v for k, v in snap.items()})
            print(f"{name:12s} N_imgs {field.N_imgs}  {field.H}x{field.W}  focal {float(field.focal):.4f}  keys {len(snap)}")
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
