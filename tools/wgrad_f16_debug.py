"""Debug aid: the fp16-term weight gradient against float64 products of the stashed planes, with the table of plane maxima."""
import ctypes as C
import sys
import torch
sys.path.insert(0, "tests"); sys.path.insert(0, "nope-nerf_amd"); sys.path.insert(0, "oracle")
import test_gpu_layer_local as T

for shape in [(256, 37, 64), (256, 64, 192)]:
    r = T.run_passes(*shape, bf16=False)
    lib, cfg, ws = r["lib"], r["cfg"], r["ws"]
    n = lib.nnr_workspace_floats(C.byref(cfg))
    print(shape, "plane maxima (last 32 floats of the workspace):")
    print(ws[n - 32:].cpu().numpy())
    X, Dl, gw = r["X"], r["Dl"], r["gw"]
    for l in (1, 2, 3, 5, 6, 7):
        exact = Dl[l].T @ X[l]
        got = gw[l].double()
        print(l, "max|X| %.4g max|D| %.4g" % (float(X[l].abs().max()), float(Dl[l].abs().max())),
              "got/exact median %.6g" % float((got / exact).median()), "max|got| %.4g max|exact| %.4g" % (float(got.abs().max()), float(exact.abs().max())))
