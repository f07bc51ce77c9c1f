"""CPU: the data side (dataloading/) against golden vectors the reference's own DataField produced on the deterministic synthetic
scene (oracle/gen_golden_data.py -> tests/golden/scene_field.npz); the scene writer's files against what was drawn; the loaders
(host DataLoader and resident) against each other; layered YAML configs."""
import hashlib
import json
import os
import random
import shutil
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("nope-nerf_amd", "tools", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))

import dataloading as dl  # noqa: E402
import scene_writer  # noqa: E402
from dataloading.dataset import DataField  # noqa: E402
from gen_golden_data import CASES, SCENE  # noqa: E402  (the case table only; nothing of the reference is imported)

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "scene_field.npz"))


def digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha1(str(a.dtype).encode() + a.tobytes()).hexdigest()


@pytest.fixture(scope="module")
def scene_dir(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("scene"))
    meta = scene_writer.write_scene(tmp, **SCENE)
    root = os.path.join(tmp, SCENE["scene"])
    shutil.copytree(os.path.join(root, "dpt"), os.path.join(root, "dpt_4"))
    return tmp, meta


@pytest.mark.parametrize("name", sorted(CASES))
def test_datafield_matches_the_reference(scene_dir, name):
    tmp, _ = scene_dir
    g = lambda k: GOLD[f"{name}.{k}"]
    has = lambda k: f"{name}.{k}" in GOLD.files
    f = DataField(tmp, with_camera=True, scene_name=[SCENE["scene"]], use_DPT=False, depth_net="dpt", **CASES[name])
    assert (f.N_imgs, f.H, f.W) == (int(g("N_imgs")), int(g("H")), int(g("W")))
    assert abs(float(f.focal) - float(g("focal"))) <= 1e-6 * float(g("focal"))
    np.testing.assert_array_equal(f.K, g("K"))
    np.testing.assert_array_equal(f.i_train, g("i_train"))
    np.testing.assert_array_equal(f.i_test, g("i_test"))
    assert list(f.img_list) == list(g("img_list"))
    assert f.imgs.dtype == np.float32 and list(f.imgs.shape) == list(g("imgs_shape"))
    assert digest(f.imgs) == str(g("imgs_sha1"))                     # every pixel of every frame, bit for bit
    for k in ("c2ws", "c2ws_colmap"):
        assert has(k) == (getattr(f, k, None) is not None)
        if has(k):
            np.testing.assert_allclose(getattr(f, k).numpy(), g(k), rtol=0, atol=1e-6)
    if has("hwf"):
        np.testing.assert_allclose(f.hwf, g("hwf"), rtol=0, atol=1e-6)
    assert digest(f.dpt_depth) == str(g("dpt_sha1"))
    if has("depth_sha1"):
        assert digest(f.depth) == str(g("depth_sha1"))
    random.seed(11)                                                  # the neighbour pick draws from Python's `random`
    for i in (0, f.N_imgs // 2, f.N_imgs - 1):
        if has(f"load{i}.index_error"):
            with pytest.raises(IndexError):
                f.load(i)
            continue
        d = f.load(i)
        assert sorted("" if k is None else k for k in d) == list(g(f"load{i}.keys"))
        for k in ("ref_idxs", "idx"):
            if has(f"load{i}.{k}"):
                assert int(d[k]) == int(g(f"load{i}.{k}"))


def test_written_files_hold_what_was_drawn(scene_dir):
    tmp, meta = scene_dir
    root = os.path.join(tmp, SCENE["scene"])
    h, w = meta["size"]
    arr = np.load(os.path.join(root, "poses_bounds.npy"))
    assert arr.shape == (SCENE["frames"], 17)
    blk = arr[:, :15].reshape(-1, 3, 5)
    c2w = np.array(meta["c2w"])
    np.testing.assert_allclose(blk[:, :, 1], c2w[:, :3, 0], atol=1e-12)      # LLFF column 1 = right
    np.testing.assert_allclose(-blk[:, :, 0], c2w[:, :3, 1], atol=1e-12)     # LLFF column 0 = down
    np.testing.assert_allclose(blk[:, :, 4], np.tile([h, w, meta["focal"]], (SCENE["frames"], 1)))
    R = c2w[:, :3, :3]
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.tile(np.eye(3), (len(R), 1, 1)), atol=1e-12)
    assert np.allclose(np.linalg.det(R), 1.0)
    # monocular depth = affine distortion of the sensor depth; the mm PNG is the z-depth itself
    f = DataField(tmp, with_camera=True, scene_name=[SCENE["scene"]], resize_factor=None, with_depth=True, sample_rate=10 ** 6,
                  depth_net="dpt")
    assert f.N_imgs == SCENE["frames"]             # a sample_rate beyond the frame count holds out no view
    z = np.asarray(f.depth)
    k = 0
    s, t = meta["scales"][int(f.i_train[k])], meta["shifts"][int(f.i_train[k])]
    hd, wd = meta["depth_size"]
    assert f.dpt_depth.shape[1:] == (hd, wd)
    # depth maps are drawn at the training resolution (factor 2): compare against the full-resolution map through its corners
    full = z[k]
    np.testing.assert_allclose(f.dpt_depth[k][0, 0] * s + t, full[0, 0], atol=2e-3)
    np.testing.assert_allclose(f.dpt_depth[k][-1, -1] * s + t, full[-1, -1], atol=2e-3)
    assert json.load(open(os.path.join(root, "scene.json")))["seed"] == SCENE["seed"]


def test_frames_are_consistent_with_the_render_path_rays(scene_dir):
    """Unproject a pixel of view a with its true depth through the loader's K and pose, reproject into view b with
    model.common.project_to_cam-style arithmetic: colours must agree (Lambertian scene, no occlusion at the picked pixels)."""
    tmp, meta = scene_dir
    h, w = meta["size"]
    f = meta["focal"]
    world = scene_writer.Scene(SCENE["seed"])
    c2w = np.array(meta["c2w"])
    a, b = 3, 4
    rgb_a, z_a = scene_writer.render_view(world, c2w[a], h, w, f)
    rgb_b, z_b = scene_writer.render_view(world, c2w[b], h, w, f)
    d = scene_writer.pixel_dirs(h, w, f).reshape(h, w, 3)
    hits = 0
    for (y, x) in ((10, 12), (24, 32), (40, 50), (30, 8)):
        p = c2w[a, :3, 3] + z_a[y, x] * (c2w[a, :3, :3] @ d[y, x])
        q = c2w[b, :3, :3].T @ (p - c2w[b, :3, 3])               # camera b coordinates, looking down -z
        u = (q[0] / -q[2]) * (2 * f / w)                           # normalised image coordinates in [-1, 1]
        v = (q[1] / -q[2]) * (-2 * f / h)
        xb, yb = (u + 1) * (w - 1) / 2, (v + 1) * (h - 1) / 2
        xi, yi = int(round(xb)), int(round(yb))
        if not (0 <= xi < w and 0 <= yi < h) or abs(z_b[yi, xi] - (-q[2])) > 0.05:
            continue                                               # left the frame or occluded in b
        assert np.abs(rgb_b[yi, xi] - rgb_a[y, x]).max() < 0.08     # nearest-pixel lookup of a smooth texture
        hits += 1
    assert hits >= 2


def _cfg(tmp, **over):
    cfg = {"dataloading": {"dataset_name": "any", "path": tmp, "scene": [SCENE["scene"]], "batchsize": 1, "n_workers": 0,
                           "with_depth": False, "spherify": True, "customized_poses": False, "customized_focal": False,
                           "resize_factor": None, "depth_net": "dpt", "crop_size": 0, "random_ref": 1, "norm_depth": False,
                           "load_colmap_poses": True, "shuffle": True, "sample_rate": 8},
           "training": {"pc_weight": [1.0, 0.0], "rgb_s_weight": [1.0, 0.0]}, "depth": {"type": "None"}}
    cfg["dataloading"].update(over)
    return cfg


def test_host_and_resident_loaders_serve_the_same_batches(scene_dir):
    tmp, _ = scene_dir
    batches = {}
    for resident in (False, True):
        torch.manual_seed(3)
        random.seed(3)
        loader, fields = dl.get_dataloader(_cfg(tmp, resident=resident, resident_device="cpu"), mode="train", shuffle=True)
        assert len(loader) == fields["img"].N_imgs == 8
        batches[resident] = list(loader)
    assert len(batches[False]) == len(batches[True]) == 8
    order = [int(b["img.idx"]) for b in batches[False]]
    assert sorted(order) == list(range(8)) and order != list(range(8))           # shuffled permutation of the views
    for host, res in zip(batches[False], batches[True]):
        assert sorted(host) == sorted(res)
        for k in host:
            assert torch.equal(torch.as_tensor(host[k]), torch.as_tensor(res[k])), k
        assert host["img"].shape == (1, 3, 48, 64) and host["img.dpt"].shape == (1, 24, 32) or host["img.dpt"].shape == (1, 48, 64)
        assert host["img.camera_mat"].shape == (1, 4, 4) and int(host["img.ref_idxs"]) in (int(host["img.idx"]) + 1, 6)


def test_render_mode_serves_cameras_only(scene_dir):
    tmp, _ = scene_dir
    for resident in (False, True):
        loader, _ = dl.get_dataloader(_cfg(tmp, resident=resident, resident_device="cpu"), mode="render", shuffle=False, n_views=5)
        got = list(loader)
        assert len(got) == 5 and sorted(got[0]) == ["img.camera_mat", "img.idx", "img.scale_mat"]
        assert [int(b["img.idx"]) for b in got] == [0, 1, 2, 3, 4]


def test_unsupported_requests_fail_loudly(scene_dir):
    tmp, _ = scene_dir
    cfg = _cfg(tmp)
    cfg["depth"]["type"] = "DPT"
    with pytest.raises(NotImplementedError):
        dl.get_dataloader(cfg)
    with pytest.raises(ValueError):
        dl.get_dataloader(_cfg(tmp, dataset_name="DTU"))
    with pytest.raises(ValueError):
        dl.get_dataloader(_cfg(tmp, resident=True, batchsize=2))


def test_minify_creates_the_downsized_folder(scene_dir, tmp_path):
    tmp, _ = scene_dir
    dst = str(tmp_path / "copy")
    shutil.copytree(os.path.join(tmp, SCENE["scene"]), os.path.join(dst, SCENE["scene"]))
    f = DataField(dst, with_camera=True, scene_name=[SCENE["scene"]], resize_factor=4, depth_net="dpt")
    assert os.path.isdir(os.path.join(dst, SCENE["scene"], "images_4")) and (f.H, f.W) == (12, 16)
    assert abs(float(f.focal) - 0.9 * 64 / 4) < 1e-5


def test_layered_configs(tmp_path):
    base, top, mid = tmp_path / "default.yaml", tmp_path / "scene.yaml", tmp_path / "mid.yaml"
    base.write_text("a: {x: 1, y: {z: 2}}\nb: 5\nempty:\n")
    mid.write_text("a: {y: {z: 3}}\n")
    top.write_text("a: {x: 7}\nc: [1, 2]\nempty: {k: 1}\n")
    cfg = dl.load_config(str(top), str(base))
    assert cfg == {"a": {"x": 7, "y": {"z": 2}}, "b": 5, "c": [1, 2], "empty": {"k": 1}}
    cfg = dl.load_config(str(top), str(base), inherit_from=str(mid))
    assert cfg["a"] == {"x": 7, "y": {"z": 3}} and cfg["b"] == 5
    assert dl.load_config(str(mid)) == {"a": {"y": {"z": 3}}}
