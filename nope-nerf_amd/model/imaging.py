"""Full-image inference helper shared by Eval_Images / Extract_Images: renders an (h, w) frame through the forward-only
fused kernel in chunks of `points_batch_size` rays (reference model/eval_images.py:66-87,
model/extracting_images.py:50-74)."""
import numpy as np
import torch
from PIL import Image

from model.common import arange_pixels


def camera_from_focal(fxfy, device):
    """(1,4,4) K = diag(fx, -fy, -1, 1) from a (frozen) learned focal pair; device-side copies, no host sync."""
    k = torch.zeros(1, 4, 4, device=device)
    fxfy = fxfy.detach().to(device)
    k[0, 0, 0], k[0, 1, 1], k[0, 2, 2], k[0, 3, 3] = fxfy[0], -fxfy[1], -1.0, 1.0
    return k


def inverse_pose(c2w):
    from nnr import camera
    return (camera.inverse4(c2w) if c2w.is_cuda else torch.inverse(c2w)).unsqueeze(0)


def render_full_image(renderer, resolution, camera_mat, world_mat, scale_mat, render_type, device, points_batch_size=100000, it=0):
    """-> rgb (h,w,3) float tensor on `device`, depth (h,w) numpy.  The mono-depth input is all ones so that no ray is
    masked (the reference builds the same constant through a grid_sample of zeros)."""
    h, w = resolution
    pixels = arange_pixels(resolution=(h, w), device=device)[1]
    depth = torch.ones(1, h * w, 1, device=device)
    rgb, dep = [], []
    with torch.no_grad():
        for pix_i, d_i in zip(torch.split(pixels, points_batch_size, dim=1), torch.split(depth, points_batch_size, dim=1)):
            out = renderer(pix_i, d_i, camera_mat, world_mat, scale_mat, render_type, eval_=True, it=it, add_noise=False)
            rgb.append(out['rgb'])
            dep.append(out['depth_pred'])
    return torch.cat(rgb, dim=1).view(h, w, 3), torch.cat(dep, dim=0).view(h, w).cpu().numpy()


def depth_to_u8(depth):
    return np.clip(255.0 / depth.max() * (depth - depth.min()), 0, 255).astype(np.uint8)


def resize_nearest(arr, size_hw):
    """cv2.resize(..., INTER_NEAREST) for a 2-D array: source index floor(dst * scale)."""
    gh, gw = size_hw
    ys = np.minimum((np.arange(gh) * (arr.shape[0] / gh)).astype(np.int64), arr.shape[0] - 1)
    xs = np.minimum((np.arange(gw) * (arr.shape[1] / gw)).astype(np.int64), arr.shape[1] - 1)
    return arr[ys][:, xs]


def save_png(arr_u8, path):
    Image.fromarray(arr_u8).save(path)


def ssim_gaussian(a, b, window_size=11, sigma=1.5):
    """Mean SSIM of two (B,C,H,W) images in [0,1] (Wang et al. 2004: Gaussian window, zero-padded 'same' filtering, K1 = 0.01,
    K2 = 0.03, dynamic range 1, mean over pixels and channels) -- the metric evaluation/eval.py reports through
    third_party/pytorch_ssim.ssim; the window is applied as two 1-D passes."""
    x = torch.arange(window_size, dtype=torch.float32, device=a.device) - window_size // 2
    g = torch.exp(-x * x / (2.0 * sigma * sigma))
    g = g / g.sum()
    c = a.shape[1]
    kh, kw = g.view(1, 1, -1, 1).expand(c, 1, -1, 1), g.view(1, 1, 1, -1).expand(c, 1, 1, -1)
    pad = window_size // 2

    def blur(t):
        t = torch.nn.functional.conv2d(t, kh, padding=(pad, 0), groups=c)
        return torch.nn.functional.conv2d(t, kw, padding=(0, pad), groups=c)

    mu_a, mu_b = blur(a), blur(b)
    var_a, var_b, cov = blur(a * a) - mu_a * mu_a, blur(b * b) - mu_b * mu_b, blur(a * b) - mu_a * mu_b
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu_a * mu_b + c1) * (2 * cov + c2)) / ((mu_a * mu_a + mu_b * mu_b + c1) * (var_a + var_b + c2))).mean()
