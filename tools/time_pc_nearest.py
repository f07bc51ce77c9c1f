"""The nearest-neighbour search alone (nnr_pc_nearest), HIP events:  python tools/time_pc_nearest.py [S]   -- two back-projected depth maps of S
points each (pixel grid x random depth, like the trainer's clouds at 540 x 960 / pc_ratio 4: 32 400).  NNR_PC_PER / NNR_PC_WGS: the kernel's knobs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch
from nnr import pointcloud

if __name__ == "__main__":
    S = int(sys.argv[1]) if len(sys.argv) > 1 else 32400
    g = torch.Generator().manual_seed(0)
    w = int(round((S * 16 / 9) ** 0.5)); h = S // w; S = h * w
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing='ij')
    smooth = len(sys.argv) > 2 and sys.argv[2] == "smooth"      # a smooth depth map (+ 1 % noise) instead of white-noise depth
    def cloud():
        d = (2 + 0.6 * torch.sin(3 * xs + torch.rand(1, generator=g) * 6) * torch.cos(2 * ys) + 0.02 * torch.rand(h, w, generator=g)) if smooth \
            else 1 + 2 * torch.rand(h, w, generator=g)
        return torch.stack([xs * d * 0.7, ys * d * 0.4, -d], -1).view(S, 3).cuda()
    x, y = cloud(), cloud()
    for _ in range(3):
        pointcloud.nearest(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        idx, dist = pointcloud.nearest(x, y)
    e1.record()
    torch.cuda.synchronize()
    print(("smooth depth, " if smooth else "white-noise depth, ") + "S = D = %d: nnr_pc_nearest %.1f us per call (fill + search + decode); index checksum %d" % (S, e0.elapsed_time(e1) / 20 * 1e3, int(idx.sum())))
