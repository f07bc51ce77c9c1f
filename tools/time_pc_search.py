"""Time of the per-image forward (clouds + ray-aware nearest-neighbour search + sums) at the first-phase grid, noise and smooth depths.
   python tools/time_pc_search.py        (NNR_PC_SEARCH=brute for the exhaustive search)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_pc_search as T
from nnr import lib as L

if __name__ == "__main__":
    lib = L.load()
    dev = torch.device("cuda", 0)
    hd, wd, hr, wr = 540, 960, 135, 240
    scene = None
    if "--scene" in sys.argv:      # + two neighbouring frames of the loop_rate scene, poses at the identity: the start of a real training run
        import tempfile
        import numpy as np
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import scene_writer
        with tempfile.TemporaryDirectory() as d:
            scene_writer.write_scene(d, scene="s", frames=3, size=(540, 960), seed=0)
            scene = [torch.from_numpy(np.load(os.path.join(d, "s", "dpt", "depth_%03d.npz" % i))["pred"][0]).to(dev) for i in (1, 2)]
            fpx = float(np.load(os.path.join(d, "s", "intrinsics.npz"))["K"][0, 0])
    for kind in ("noise", "smooth") + (("scene",) if scene else ()):
        g = torch.Generator().manual_seed(1)
        if kind == "scene":
            d1, d2 = scene
            f = fpx
            rel = T._rel()
        else:
            d1, d2 = T._depths(kind, hd, wd, g).to(dev), T._depths(kind, hd, wd, g).to(dev)
            f = 0.7 * wd
            rel = T._rel((0.2, 1.0, 0.1), 0.04, (0.05, -0.02, 0.03))
        K = torch.diag(torch.tensor([2 * f / wd, -2 * f / hd, -1.0, 1.0]))
        Kinv = torch.linalg.inv(K.double()).float()
        K_c, Kinv_c, rel_c = (t.reshape(16).contiguous().float().to(dev) for t in (K, Kinv, rel))
        s2 = torch.tensor([1.0], dtype=torch.float32, device=dev)
        cfg = L.AuxCfg(hd, wd, hr, wr, 0.05, L.AUX_PC | L.AUX_SCALE_PCS, 0, 0)
        ws = torch.zeros(lib.nnr_aux_workspace_floats(C.byref(cfg)) + 2, dtype=torch.float32, device=dev)
        ws = ws[(ws.data_ptr() % 8) // 4:]
        out = torch.empty(4, dtype=torch.float32, device=dev)
        run = lambda: L.check(lib.nnr_aux_terms_fwd(C.byref(cfg), L.ptr(d1), L.ptr(d2), None, None, L.ptr(K_c), L.ptr(Kinv_c), L.ptr(rel_c), L.ptr(s2), None,
                                                    L.ptr(out), L.ptr(ws), L.stream()), "fwd")
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100):
            run()
        b.record()
        torch.cuda.synchronize()
        print("%-6s depths, 135 x 240 grid: per-image forward %.1f us (loss_pc %.6f)" % (kind, a.elapsed_time(b) * 10, float(out[0])))
        if os.environ.get("NNR_PC_DEBUG_COUNTS"):      # a -DNNR_PC_DEBUG build: tile evaluations and sphere tests per wave of the search
            S = hr * wr
            ws[31 * S + 4:31 * S + 7] = 0
            run()
            torch.cuda.synchronize()
            waves = 2 * 4 * ((hr + 7) // 8) * ((wr + 7) // 8)
            print("       per wave of the tile kernel: %.1f turns, %.1f per-source tests, %.1f tile evaluations" % (float(ws[31 * S + 4]) / waves, float(ws[31 * S + 6]) / waves, float(ws[31 * S + 5]) / waves))
