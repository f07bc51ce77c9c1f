"""Golden vectors for the full-image drivers: reference Eval_Images.eval_images (model/eval_images.py:44-137:
full-frame render with a given pose, MSE / PSNR / SSIM, depth resize, PNGs) and Extract_Images.generate_images
(model/extracting_images.py:38-123) on a tiny frame with the seed-42 D=128 network.  imageio / cv2 are absent here: minimal
PIL / numpy stand-ins for imwrite and for cv2.resize(..., INTER_NEAREST).  tests/golden/eval_images.npz.
Authoring container only:  python oracle/gen_golden_evalimg.py"""
import importlib.machinery
import os
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402
from gen_golden_poses import trajectory  # noqa: E402

H, W, N = 12, 16, 8


def stubs():
    imageio = types.ModuleType("imageio")
    imageio.imwrite = lambda path, arr: Image.fromarray(np.asarray(arr)).save(path)
    imageio.imread = lambda f, **k: np.asarray(Image.open(f))
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.INTER_AREA, cv2.INTER_CUBIC, cv2.IMREAD_UNCHANGED = 0, 3, 2, -1

    def resize(a, size, interpolation=None):       # cv2.resize(src, (w, h), INTER_NEAREST): source index floor(dst * scale)
        gw, gh = size
        ys = np.minimum((np.arange(gh) * (a.shape[0] / gh)).astype(np.int64), a.shape[0] - 1)
        xs = np.minimum((np.arange(gw) * (a.shape[1] / gw)).astype(np.int64), a.shape[1] - 1)
        return a[ys][:, xs]
    cv2.resize = resize
    for m in (imageio, cv2):
        m.__spec__ = importlib.machinery.ModuleSpec(m.__name__, None)
        sys.modules[m.__name__] = m


def main():
    stubs()
    ref = gg.import_reference()
    from model.eval_images import Eval_Images
    from model.extracting_images import Extract_Images
    cfg = gg.base_cfg(128)
    cfg["rendering"]["num_points"] = N
    cfg["extract_images"]["resolution"] = [H, W]
    dev = torch.device("cpu")
    torch.manual_seed(42)
    net = ref.OfficialStaticNerf(cfg)
    renderer = ref.Renderer(net, cfg["rendering"], device=dev)
    c2ws, _ = trajectory(3, 21, 0.0)
    c2ws[:, :3, 3] *= 0.1
    g = torch.Generator().manual_seed(3)
    img = torch.rand(1, 3, H, W, generator=g)
    depth_gt = 0.05 + 3 * torch.rand(1, 8, 10, generator=g)          # sensor depth at its own resolution; some below min_depth
    f = 0.8 * W
    K = torch.diag(torch.tensor([2 * f / W, -2 * f / H, -1.0, 1.0])).unsqueeze(0)
    zero = lambda a, b, normalize=True: torch.zeros(())
    blob = {"img": img.numpy(), "depth_gt": depth_gt.numpy(), "K": K.numpy(), "c2ws": c2ws.numpy()}
    with tempfile.TemporaryDirectory() as tmp:
        ev = Eval_Images(renderer, cfg, use_learnt_poses=True, use_learnt_focal=False, device=dev, render_type="nope_nerf", c2ws=c2ws,
                         img_list=["a.png", "b.png", "c.png"])
        data = {"img": img, "img.depth": depth_gt, "img.idx": torch.tensor([1]), "img.camera_mat": K, "img.scale_mat": torch.eye(4).unsqueeze(0)}
        out = ev.eval_images(data, tmp, None, zero, logger=None, min_depth=0.1, max_depth=20)
        for k in ("img", "depth", "depth_pred", "depth_gt"):
            blob["eval." + k] = np.asarray(out[k])
        for k in ("mse", "psnr", "ssim", "lpips"):
            blob["eval." + k] = np.float64(out[k])
        for sub in ("img_out", "depth_out", "img_gt_out"):
            blob[f"eval.png.{sub}"] = np.asarray(Image.open(os.path.join(tmp, sub, "0001.png")))
        # learnt focal: K rebuilt from (fx, fy)
        ev2 = Eval_Images(renderer, cfg, use_learnt_poses=True, use_learnt_focal=True, device=dev, render_type="nope_nerf", c2ws=c2ws)
        out2 = ev2.eval_images(dict(data, **{"img.idx": torch.tensor([2])}), os.path.join(tmp, "f"), torch.tensor([1.5, 2.1]), zero, logger=None)
        blob["eval_focal.img"], blob["eval_focal.psnr"] = np.asarray(out2["img"]), np.float64(out2["psnr"])
        ex = Extract_Images(renderer, cfg, use_learnt_poses=True, use_learnt_focal=False, device=dev, render_type="nope_nerf")
        cam = {"img.idx": torch.tensor([0]), "img.camera_mat": K, "img.scale_mat": torch.eye(4).unsqueeze(0)}
        out3 = ex.generate_images(cam, os.path.join(tmp, "x"), c2ws, None, 0, False)
        blob["extract.img"], blob["extract.depth"] = np.asarray(out3["img"]), np.asarray(out3["depth"])
        blob["extract.depth_npy"] = np.load(os.path.join(tmp, "x", "depth_out", "0.npy"))
        assert out3["geo"] is None
        # Trainer.render_visdata (model/training.py:100-163): the periodic visualisation render of a training view
        tcfg = dict(cfg["training"], vis_geo=False)
        model = ref.get_model(renderer, cfg, device=dev)
        pose = ref.LearnPose(3, True, True, cfg, init_c2w=c2ws)
        sgd = lambda m: torch.optim.SGD(m.parameters(), lr=0.0)
        tr = ref.Trainer(model, sgd(model), tcfg, device=dev, optimizer_pose=sgd(pose), pose_param_net=pose)
        vis = os.path.join(tmp, "vis")
        os.makedirs(vis)
        dpt = 1 + 2 * torch.rand(1, 9, 12, generator=g)
        blob["vis.dpt"] = dpt.numpy()
        vdata = {"img": img, "img.dpt": dpt, "img.idx": 2, "img.camera_mat": K, "img.scale_mat": torch.eye(4).unsqueeze(0)}
        ret = tr.render_visdata(vdata, (6, 8), 100, vis)
        blob["vis.ret"] = np.asarray(ret)
        blob["vis.png.img"] = np.asarray(Image.open(os.path.join(vis, "0002_img.png")))
        blob["vis.png.depth"] = np.asarray(Image.open(os.path.join(vis, "0002_depth.png")))
    print({k: float(blob["eval." + k]) for k in ("mse", "psnr", "ssim")}, "valid depths", blob["eval.depth_gt"].shape)
    out_path = os.path.join(gg.OUT, "eval_images.npz")
    np.savez_compressed(out_path, **blob)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
