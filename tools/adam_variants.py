"""Which arithmetic does torch.optim.Adam(fused=True) carry out in this torch build?  Compiles tools/ubench/adam_variants.hip on the GPU
box and compares each candidate expression BITWISE with torch's kernel over a few steps (counts of differing elements out of 2^20).

    python tools/adam_variants.py
"""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "ubench", "adam_variants.hip")
SO = "/tmp/adam_variants.so"


def main():
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", SO, SRC], check=True)
    lib = C.CDLL(SO)
    lib.adam_variant.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_double] * 4 + [C.c_float] + [C.c_int] * 3 + [C.c_void_p]
    dev = torch.device("cuda")
    n = 1 << 20
    gen = torch.Generator().manual_seed(3)
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    p0 = torch.randn(n, generator=gen).to(dev)

    def grads(k):
        g = torch.Generator().manual_seed(100 + k)
        mag = 10.0 ** (torch.rand(n, generator=g) * 10 - 7)
        return (torch.randn(n, generator=g) * mag).to(dev)

    # torch's trajectory: states after each of 4 steps
    p = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=eps, fused=True)
    traj = []
    for k in range(4):
        p.grad = grads(k)
        opt.step()
        st = opt.state[p]
        traj.append((p.detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone()))
    st_ptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(k, va, vb, vc):
        """one step of variant (va, vb, vc) from torch's state after step k-1; returns mismatch counts vs torch's state after step k"""
        if k == 0:
            pp, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        else:
            pp, m, v = (t.clone() for t in traj[k - 1])
        g = grads(k)
        rc = lib.adam_variant(pp.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, b1, b2, eps, float(k + 1), va, vb, vc, st_ptr)
        assert rc == 0
        torch.cuda.synchronize()
        tp, tm, tv = traj[k]
        return int((pp != tp).sum()), int((m != tm).sum()), int((v != tv).sum())

    print("first moment variants (elements differing from torch, steps 2..4):")
    for va in range(20):
        print("  va=%d" % va, [run(k, va, 0, 0)[1] for k in (1, 2, 3)])
    # a few elements where the closest variants disagree with torch: old moment, gradient, torch's result, the variant's, and the exactly
    # rounded value of beta1 * m + (1 - beta1) * g in rational arithmetic
    from fractions import Fraction
    import struct
    f2h = lambda x: struct.pack('>f', x).hex()
    for va in (0, 6):
        k = 2
        pp, m, v = (t.clone() for t in traj[k - 1])
        g = grads(k)
        lib.adam_variant(pp.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr, b1, b2, eps, float(k + 1), va, 0, 0, st_ptr)
        torch.cuda.synchronize()
        bad = (m != traj[k][1]).nonzero().reshape(-1)[:6].cpu()
        m_old = traj[k - 1][1].cpu()
        for i in bad.tolist():
            mo, gg, tn, vn = float(m_old[i]), float(g[i].cpu()), float(traj[k][1][i].cpu()), float(m[i].cpu())
            exact = Fraction(b1) * Fraction(mo) + (1 - Fraction(b1)) * Fraction(gg)
            exact_w = Fraction(mo) + Fraction(1 - b1) * (Fraction(gg) - Fraction(mo))
            print("  va=%d i=%d m_old %s (%.9g) g %s (%.9g) torch %s variant %s exact(beta form) %s exact(lerp, w=double(1-b1)) %s"
                  % (va, i, f2h(mo), mo, f2h(gg), gg, f2h(tn), f2h(vn), f2h(float(exact)), f2h(float(exact_w))))
    print("second moment variants:")
    for vb in range(9):
        print("  vb=%d" % vb, [run(k, 0, vb, 0)[2] for k in (1, 2, 3)])
    best_a = min(range(20), key=lambda va: sum(run(k, va, 0, 0)[1] for k in (1, 2, 3)))
    best_b = min(range(9), key=lambda vb: sum(run(k, 0, vb, 0)[2] for k in (1, 2, 3)))
    print("best va", best_a, "best vb", best_b)
    print("parameter update variants (with the best moments):")
    for vc in list(range(5)) + [8 + i for i in range(5)]:
        print("  vc=%d" % vc, [run(k, best_a, best_b, vc)[0] for k in (0, 1, 2, 3)])


if __name__ == "__main__":
    main()
