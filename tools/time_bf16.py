"""Time the bf16-MFMA variants (NNR_F_BF16) of the MLP forward / input-gradient kernels at the benchmark shape next to the
fp32 ones (HIP events on the launch stream)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch

import bench

if __name__ == "__main__":
    from nnr import lib as L
    from nnr import ops
    import model as mdl
    dev = torch.device("cuda", 0)
    lib = L.load()
    R, N, D = bench.R_PER_GPU, bench.N_SAMPLES, bench.HIDDEN
    torch.manual_seed(42)
    net = mdl.OfficialStaticNerf(bench.full_cfg(R)).to(dev)
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
    out = {}
    for bf16 in (False, True):
        for train in (True, False):
            cfg = L.make_cfg(R, N, D, train=train, bf16=bf16)
            packed = ops._packed_for(cfg, net.weights(), net.biases())
            ws = torch.zeros(lib.nnr_workspace_floats(C.byref(cfg)), device=dev)
            fns = {"fwd": lambda: lib.nnr_mlp_fwd(C.byref(cfg), L.ptr(o), L.ptr(d), L.ptr(view), L.ptr(lo), L.ptr(hi), L.ptr(jit),
                                                  L.ptr(packed), L.ptr(ws), st)}
            if train:
                fns["dgrad"] = lambda: lib.nnr_mlp_dgrad(C.byref(cfg), L.ptr(packed), L.ptr(ws), st)
                gw = [torch.zeros_like(x) for x in net.weights()]
                gb = [torch.zeros_like(x) for x in net.biases()]
                gs = L.params_struct(gw, gb)
                plan = ops._plan_for(cfg, dev)
                fns["wgrad"] = lambda: lib.nnr_mlp_wgrad(C.byref(cfg), L.ptr(packed), C.byref(gs), L.ptr(plan), L.ptr(ws), st)
            for name, fn in fns.items():
                L.check(fn(), name)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    L.check(fn(), name)
                e1.record()
                torch.cuda.synchronize()
                out["%s_%s_%s" % (name, "train" if train else "infer", "bf16" if bf16 else "fp32")] = round(e0.elapsed_time(e1) / 5, 4)
    print(json.dumps(out))
