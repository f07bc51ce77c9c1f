"""Time the point-cloud loss (fwd+bwd) at the trainer's size: HIP op vs the reference's dense torch expression on the GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch
from model.losses import Loss

def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 96 * 168
    g = torch.Generator().manual_seed(0)
    if len(sys.argv) > 2:   # "grid": two back-projected depth maps (pixel grid x random depth), like the trainer's clouds
        w = int(round((S * 16 / 9) ** 0.5)); h = S // w; S = h * w
        ys, xs = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing='ij')
        def cloud():
            d = 1 + 2 * torch.rand(h, w, generator=g)
            return torch.stack([xs * d * 0.7, ys * d * 0.4, -d], -1).view(1, S, 3)
        x, y = cloud().cuda().requires_grad_(True), cloud().cuda().requires_grad_(True)
    else:
        x = (torch.rand(1, S, 3, generator=g) * 4).cuda().requires_grad_(True)
        y = (torch.rand(1, S, 3, generator=g) * 4).cuda().requires_grad_(True)
    lm = Loss({'depth_loss_type': 'l1', 'match_method': 'dense', 'with_ssim': False})
    def hip():
        l = lm.get_pc_loss(x, y); l.backward()
    def dense():
        xt, yt = x[0].permute(1, 0), y[0].permute(1, 0)
        def ppe(a, b):
            idx = lm.comp_closest_pts_idx_with_split(a, b)
            return torch.linalg.norm(a - b[:, idx], dim=0).mean()
        (ppe(xt, yt) + ppe(yt, xt)).backward()
    print("S = D = %d: HIP %.3f ms, dense torch on the same GPU %.3f ms" % (S, bench(hip), bench(dense, 3)))
