"""How large are the search windows of the per-image nearest-neighbour search on a REAL scene at the start of training (poses at the
identity, distortions at 1 / 0)?  Writes the loop_rate scene (tools/scene_writer.py, 16 frames of 540 x 960), lifts neighbouring
mono-depth maps as the trainer does and prints the distribution of nearest-neighbour distances and of the pixel radius the ray-aware
search needs for them (rho B / (p_z pitch)).   python tools/pc_window_stats.py"""
import ctypes as C
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "nope-nerf_amd"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import scene_writer
import test_gpu_pc_search as T
from nnr import lib as L, pointcloud

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    lib = L.load()
    with tempfile.TemporaryDirectory() as d:
        scene_writer.write_scene(d, scene="s", frames=4, size=(540, 960), seed=0)
        dpt = [torch.from_numpy(np.load(os.path.join(d, "s", "dpt", "depth_%03d.npz" % i))["pred"][0]).to(dev) for i in range(4)]
        Kpix = np.load(os.path.join(d, "s", "intrinsics.npz"))["K"]
    hd, wd, hr, wr = 540, 960, 135, 240
    fx, fy = float(Kpix[0, 0]), float(Kpix[1, 1])
    K = torch.diag(torch.tensor([2 * fx / wd, -2 * fy / hd, -1.0, 1.0]))
    print("depth maps: min %.3f median %.3f max %.3f; focal %.1f px" % (float(dpt[0].min()), float(dpt[0].median()), float(dpt[0].max()), fx))
    for (i, j) in ((0, 1), (1, 2), (2, 3)):
        Kinv = torch.linalg.inv(K.double()).float()
        K_c, Kinv_c, rel_c = (t.reshape(16).contiguous().float().to(dev) for t in (K, Kinv, T._rel()))
        s2 = torch.tensor([1.0], dtype=torch.float32, device=dev)
        cfg = L.AuxCfg(hd, wd, hr, wr, 0.05, L.AUX_PC | L.AUX_SCALE_PCS, 0, 0)
        ws = torch.zeros(lib.nnr_aux_workspace_floats(C.byref(cfg)) + 2, dtype=torch.float32, device=dev)
        ws = ws[(ws.data_ptr() % 8) // 4:]
        out = torch.empty(4, dtype=torch.float32, device=dev)
        L.check(lib.nnr_aux_terms_fwd(C.byref(cfg), L.ptr(dpt[i]), L.ptr(dpt[j]), None, None, L.ptr(K_c), L.ptr(Kinv_c), L.ptr(rel_c), L.ptr(s2), None,
                                      L.ptr(out), L.ptr(ws), L.stream()), "fwd")
        torch.cuda.synchronize()
        S = hr * wr
        X, Y = ws[20 * S:23 * S].view(S, 3).clone(), ws[23 * S:26 * S].view(S, 3).clone()
        _, dist = pointcloud.nearest(X, Y)
        pz = -X[:, 2]                                            # depth along the camera axis (K's third row is -1)
        pitch = 2.0 / (wr - 1)
        B = float(torch.linalg.norm(Kinv[:3, :3] @ torch.tensor([1.0, 1.0, 1.0])))
        radius = dist * B / (pz.abs() * pitch) / float(abs(Kinv[0, 0]))     # rough: pixels in x
        q = lambda t, p: float(torch.quantile(t.float(), p))
        print("frames %d -> %d: NN distance median %.4f p90 %.4f max %.4f | depth median %.2f | radius [px] median %.1f p90 %.1f p99 %.1f max %.1f | loss_pc %.4f"
              % (i, j, q(dist, 0.5), q(dist, 0.9), float(dist.max()), q(pz, 0.5), q(radius, 0.5), q(radius, 0.9), q(radius, 0.99), float(radius.max()), float(out[0])))
