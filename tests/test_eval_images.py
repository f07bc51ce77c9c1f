"""The full-image drivers against goldens from the reference classes (oracle/gen_golden_evalimg.py -> tests/golden/eval_images.npz):
Eval_Images.eval_images (render with a given pose, MSE / PSNR / SSIM, nearest depth resize, masked depths, PNGs) and
Extract_Images.generate_images.  CPU: oracle-backed operator; gpu: the forward-only HIP kernel."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in ("nope-nerf_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))

GOLD = np.load(os.path.join(HERE, "golden", "eval_images.npz"))
H, W, N = 12, 16, 8


def _devices():
    return [pytest.param("cpu"), pytest.param("cuda", marks=pytest.mark.gpu)]


def _renderer(dev, monkeypatch):
    import model as mdl
    from test_host_logic import make_cfg
    if dev == "cpu":
        import oracle_backend
        from model import rendering
        monkeypatch.setattr(rendering.nnr, "render_rays", oracle_backend.render_rays)
    cfg = make_cfg(128, num_points=N)
    cfg["extract_images"] = {"resolution": [H, W]}
    net = mdl.OfficialStaticNerf(cfg)
    w = np.load(os.path.join(HERE, "golden", "weights_d128.npz"))
    net.load_state_dict({k: torch.from_numpy(w[k]) for k in w.files})
    return mdl.Renderer(net, cfg["rendering"], device=torch.device(dev)), cfg


def _u8_close(a, b):
    """uint8 images of two fp32 renders that agree to 1e-4: a value on a rounding boundary may differ by one step."""
    return np.abs(a.astype(np.int32) - b.astype(np.int32)).max() <= 1


@pytest.mark.parametrize("dev", _devices())
def test_eval_images_matches_the_reference(dev, monkeypatch, tmp_path):
    from model.eval_images import Eval_Images
    renderer, cfg = _renderer(dev, monkeypatch)
    d = torch.device(dev)
    c2ws = torch.from_numpy(GOLD["c2ws"]).to(d)
    zero = lambda a, b, normalize=True: torch.zeros(())
    ev = Eval_Images(renderer, cfg, use_learnt_poses=True, use_learnt_focal=False, device=d, render_type="nope_nerf", c2ws=c2ws,
                     img_list=["a.png", "b.png", "c.png"])
    data = {"img": torch.from_numpy(GOLD["img"]), "img.depth": torch.from_numpy(GOLD["depth_gt"]), "img.idx": torch.tensor([1]),
            "img.camera_mat": torch.from_numpy(GOLD["K"]), "img.scale_mat": torch.eye(4).unsqueeze(0)}
    out = ev.eval_images(data, str(tmp_path), None, zero, logger=None, min_depth=0.1, max_depth=20)
    assert abs(out["mse"] - float(GOLD["eval.mse"])) <= 2e-6 and abs(out["psnr"] - float(GOLD["eval.psnr"])) <= 2e-4
    assert abs(out["ssim"] - float(GOLD["eval.ssim"])) <= 5e-5 and out["lpips"] == 0.0
    assert _u8_close(out["img"], GOLD["eval.img"])
    assert out["depth"].shape == GOLD["eval.depth"].shape and out["depth_gt"].shape == GOLD["eval.depth_gt"].shape
    np.testing.assert_allclose(out["depth_gt"], GOLD["eval.depth_gt"], rtol=0, atol=0)            # same mask, same sensor depths
    assert _u8_close(out["depth"], GOLD["eval.depth"]) and _u8_close(out["depth_pred"], GOLD["eval.depth_pred"])
    for sub in ("img_out", "depth_out", "img_gt_out"):
        png = np.asarray(Image.open(os.path.join(str(tmp_path), sub, "0001.png")))
        assert png.shape == GOLD[f"eval.png.{sub}"].shape and _u8_close(png, GOLD[f"eval.png.{sub}"]), sub
    ev2 = Eval_Images(renderer, cfg, use_learnt_poses=True, use_learnt_focal=True, device=d, render_type="nope_nerf", c2ws=c2ws)
    out2 = ev2.eval_images(dict(data, **{"img.idx": torch.tensor([2])}), str(tmp_path / "f"), torch.tensor([1.5, 2.1]), zero, logger=None)
    assert _u8_close(out2["img"], GOLD["eval_focal.img"]) and abs(out2["psnr"] - float(GOLD["eval_focal.psnr"])) <= 2e-4


@pytest.mark.parametrize("dev", _devices())
def test_extract_images_matches_the_reference(dev, monkeypatch, tmp_path):
    from model.extracting_images import Extract_Images
    renderer, cfg = _renderer(dev, monkeypatch)
    d = torch.device(dev)
    ex = Extract_Images(renderer, cfg, use_learnt_poses=True, use_learnt_focal=False, device=d, render_type="nope_nerf")
    cam = {"img.idx": torch.tensor([0]), "img.camera_mat": torch.from_numpy(GOLD["K"]), "img.scale_mat": torch.eye(4).unsqueeze(0)}
    out = ex.generate_images(cam, str(tmp_path), torch.from_numpy(GOLD["c2ws"]).to(d), None, 0, False)
    assert out["geo"] is None and _u8_close(out["img"], GOLD["extract.img"]) and _u8_close(out["depth"], GOLD["extract.depth"])
    np.testing.assert_allclose(np.load(os.path.join(str(tmp_path), "depth_out", "0.npy")), GOLD["extract.depth_npy"], rtol=0, atol=2e-4)


@pytest.mark.parametrize("dev", _devices())
def test_render_visdata_matches_the_reference(dev, monkeypatch, tmp_path):
    """Trainer.render_visdata (reference model/training.py:100-163): returned frame and the two PNGs."""
    import model as mdl
    renderer, cfg = _renderer(dev, monkeypatch)
    d = torch.device(dev)
    model = mdl.get_model(renderer, cfg, device=d)
    pose = mdl.LearnPose(3, True, True, cfg, init_c2w=torch.from_numpy(GOLD["c2ws"])).to(d)
    tcfg = {'type': 'nope_nerf', 'n_training_points': 16, 'vis_geo': False, 'detach_gt_depth': False, 'pc_ratio': 4, 'match_method': 'dense',
            'shift_first': False, 'detach_ref_img': True, 'scale_pcs': True, 'detach_rgbs_scale': False, 'vis_reprojection_every': 5000,
            'nearest_limit': 0.01, 'annealing_epochs': 2000, 'rgb_weight': [1.0, 1.0], 'depth_weight': [0.04, 0.0], 'pc_weight': [0.0, 0.0],
            'rgb_s_weight': [0.0, 0.0], 'depth_consistency_weight': [0.0, 0.0], 'weight_dist_2nd_loss': [0.0, 0.0],
            'weight_dist_1st_loss': [0.0, 0.0], 'depth_loss_type': 'l1', 'with_ssim': False, 'with_auto_mask': False}
    sgd = lambda m: torch.optim.SGD(m.parameters(), lr=0.0)
    tr = mdl.Trainer(model, sgd(model), tcfg, device=d, optimizer_pose=sgd(pose), pose_param_net=pose)
    data = {"img": torch.from_numpy(GOLD["img"]), "img.dpt": torch.from_numpy(GOLD["vis.dpt"]), "img.idx": 2,
            "img.camera_mat": torch.from_numpy(GOLD["K"]), "img.scale_mat": torch.eye(4).unsqueeze(0)}
    ret = tr.render_visdata(data, (6, 8), 100, str(tmp_path))
    assert ret.dtype == np.uint8 and ret.shape == GOLD["vis.ret"].shape and _u8_close(ret, GOLD["vis.ret"])
    for name, key in (("0002_img.png", "vis.png.img"), ("0002_depth.png", "vis.png.depth")):
        png = np.asarray(Image.open(os.path.join(str(tmp_path), name)))
        assert png.shape == GOLD[key].shape and _u8_close(png, GOLD[key]), name
