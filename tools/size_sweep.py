"""Kernel times of the three fused-MLP kernels across problem sizes (HIP events on the launch stream), to catch launch-sizing
pathologies away from the benchmark shape (the D=128 weight-gradient plan once ran on a single workgroup), plus the full-image
inference rate of SURVEY 8 row f3 (540x960 frame through model.imaging.render_full_image, fp32 and bf16 products).

    python tools/size_sweep.py > gpurun_out/size_sweep.json"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch

import bench

SHAPES = [(32, 64, 128), (512, 64, 128), (1024, 128, 128), (4096, 128, 128), (256, 64, 256), (1024, 128, 256), (1024, 192, 256),
          (4096, 128, 256), (8192, 128, 256), (1000, 100, 256)]


def macs_per_sample(D):
    return 63 * D + 3 * D * D + (D + 63) * D + 3 * D * D + D + D * D + (D + 27) * (D // 2) + (D // 2) * 3


def kernels(lib, L, ops, net, R, N, D, dev):
    g = torch.Generator().manual_seed(1)
    d = torch.randn(R, 3, generator=g)
    d = (d / d.norm(dim=-1, keepdim=True)).to(dev)
    o, view = torch.zeros(R, 3, device=dev), (-d).contiguous()
    z = torch.linspace(0, 1, N)
    z = 0.01 * (1 - z) + 10 * z
    mid = 0.5 * (z[1:] + z[:-1])
    lo, hi = torch.cat([z[:1], mid]).to(dev), torch.cat([mid, z[-1:]]).to(dev)
    jit = torch.rand(R, N, generator=g).to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    cfg = L.make_cfg(R, N, D, train=True)
    packed = ops._packed_for(cfg, net.weights(), net.biases())
    ws = torch.zeros(lib.nnr_workspace_floats(C.byref(cfg)), device=dev)
    gw = [torch.zeros_like(x) for x in net.weights()]
    gb = [torch.zeros_like(x) for x in net.biases()]
    gs = L.params_struct(gw, gb)
    plan = ops._plan_for(cfg, dev)
    fns = {"fwd": lambda: lib.nnr_mlp_fwd(C.byref(cfg), L.ptr(o), L.ptr(d), L.ptr(view), L.ptr(lo), L.ptr(hi), L.ptr(jit),
                                          L.ptr(packed), L.ptr(ws), st),
           "dgrad": lambda: lib.nnr_mlp_dgrad(C.byref(cfg), L.ptr(packed), L.ptr(ws), st),
           "wgrad": lambda: lib.nnr_mlp_wgrad(C.byref(cfg), L.ptr(packed), C.byref(gs), L.ptr(plan), L.ptr(ws), st)}
    flop = 2.0 * macs_per_sample(D) * R * N
    row = {"R": R, "N": N, "D": D}
    for name, fn in fns.items():
        L.check(fn(), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.check(fn(), name)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        row[name + "_ms"] = round(ms, 4)
        row[name + "_tflops"] = round(flop / ms * 1e-9, 1)
    return row


def full_image(mdl, dev, bf16):
    from model import imaging
    cfg = bench.full_cfg(1024, bf16=bf16)
    cfg["rendering"]["num_points"] = 128
    torch.manual_seed(0)
    net = mdl.OfficialStaticNerf(cfg).to(dev)
    renderer = mdl.Renderer(net, cfg["rendering"], device=dev)
    f = 0.7 * 960
    K = torch.diag(torch.tensor([2 * f / 960, -2 * f / 540, -1.0, 1.0])).unsqueeze(0).to(dev)
    eye = torch.eye(4, device=dev).unsqueeze(0)
    out = {}
    for batch in (100000, 518400):
        imaging.render_full_image(renderer, (540, 960), K, eye, eye, "nope_nerf", dev, points_batch_size=batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            imaging.render_full_image(renderer, (540, 960), K, eye, eye, "nope_nerf", dev, points_batch_size=batch)
        torch.cuda.synchronize()
        out["batch_%d_ms" % batch] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    return out


if __name__ == "__main__":
    from nnr import lib as L
    from nnr import ops
    import model as mdl
    dev = torch.device("cuda", 0)
    lib = L.load()
    nets = {}
    rows = []
    for R, N, D in SHAPES:
        if D not in nets:
            c = bench.full_cfg(R)
            c["model"]["hidden_dim"] = D
            torch.manual_seed(42)
            nets[D] = mdl.OfficialStaticNerf(c).to(dev)
        rows.append(kernels(lib, L, ops, nets[D], R, N, D, dev))
        print(json.dumps(rows[-1]), file=sys.stderr, flush=True)
    res = {"kernels": rows, "full_image_540x960_n128": {"fp32": full_image(mdl, dev, False), "bf16": full_image(mdl, dev, True)}}
    print(json.dumps(res, indent=1))
