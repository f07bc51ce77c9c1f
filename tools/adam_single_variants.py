"""Which arithmetic does torch.optim.Adam(foreach=False, fused=False) carry out in this torch build?  Compiles
tools/ubench/adam_single_variants.hip on the GPU box and compares each candidate expression of every link of the update BITWISE with
torch's own kernels over a few steps (counts of differing elements out of 2^20), link by link.      python tools/adam_single_variants.py"""
import ctypes as C
import itertools
import os
import subprocess

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "ubench", "adam_single_variants.hip")
SO = "/tmp/adam_single_variants.so"


def main():
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO, SRC], check=True)
    lib = C.CDLL(SO)
    lib.adam_single.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_double] * 5 + [C.c_int] * 4 + [C.c_void_p]
    dev = torch.device("cuda")
    n = 1 << 20
    gen = torch.Generator().manual_seed(3)
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    p0 = torch.randn(n, generator=gen).to(dev)

    def grads(k):
        g = torch.Generator().manual_seed(100 + k)
        mag = 10.0 ** (torch.rand(n, generator=g) * 10 - 7)
        return (torch.randn(n, generator=g) * mag).to(dev)

    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=eps, foreach=False, fused=False)
    traj = []
    for k in range(5):
        p.grad = grads(k)
        opt.step()
        st = opt.state[p]
        traj.append((p.detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone()))
    print("torch", torch.__version__, "state step:", opt.state[p]['step'])
    st_ptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(k, va, vb, vc, vd):
        if k == 0:
            pp, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        else:
            pp, m, v = (t.clone() for t in traj[k - 1])
        step = k + 1
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step          # torch/optim/adam.py: python floats
        rc = lib.adam_single(pp.data_ptr(), grads(k).data_ptr(), m.data_ptr(), v.data_ptr(), n, b1, b2, eps, -(lr / bc1), bc2 ** 0.5, va, vb, vc, vd, st_ptr)
        assert rc == 0
        torch.cuda.synchronize()
        tp, tm, tv = traj[k]
        return int((pp != tp).sum()), int((m != tm).sum()), int((v != tv).sum())

    ks = (0, 1, 2, 3, 4)
    print("first moment (elements differing from torch, steps 1..5):")
    for va in range(4):
        print("  va=%d" % va, [run(k, va, 0, 0, 0)[1] for k in ks])
    print("second moment:")
    for vb in range(4):
        print("  vb=%d" % vb, [run(k, 0, vb, 0, 0)[2] for k in ks])
    best_a = min(range(4), key=lambda va: sum(run(k, va, 0, 0, 0)[1] for k in ks))
    best_b = min(range(4), key=lambda vb: sum(run(k, 0, vb, 0, 0)[2] for k in ks))
    print("best va", best_a, "best vb", best_b)
    print("denominator x update (parameters differing):")
    for vc, vd in itertools.product(range(4), range(2)):
        print("  vc=%d vd=%d" % (vc, vd), [run(k, best_a, best_b, vc, vd)[0] for k in ks])


if __name__ == "__main__":
    main()
