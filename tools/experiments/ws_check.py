"""GPU check of the wave-specialised inference forward (nnr_mlp_fwd_ws.hip, NNR_FWD_WS=1) against the one-wave-per-SIMD kernel
(nnr_mlp_fwd.hip): same inputs, same packed weights -- outputs must agree to rounding (the only arithmetic difference is where the bias
is added) -- and the time of both, launches alternating.   python tools/ws_check.py [--timing-only]

Every GPU call of a shape runs in a child process with a timeout: a barrier mismatch between the two roles would hang the kernel."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))

SHAPES = [  # rays, samples, jitter, fused compositing (samples=False), dist_alpha, white_bg
    (4, 32, True, False, False, False),        # one workgroup, one chunk per ray
    (4, 64, True, True, False, False),         # two chunks: the colour layer of chunk 0 arrives during chunk 1
    (64, 96, True, False, True, False),        # odd chunk count
    (64, 96, False, True, True, True),
    (1000, 100, True, False, False, False),    # flat decomposition (N % 32 != 0), ragged tail
    (1024, 192, True, False, False, False),    # BASELINE configs[1]
    (1024, 192, True, True, False, False),
]


def child(idx, reps):
    import torch
    import bench
    import model as mdl
    from nnr import ops
    dev = torch.device("cuda", 0)
    R, N, jit, fused, da, wb = SHAPES[idx]
    cfg = bench.full_cfg(R)
    torch.manual_seed(42)
    net = mdl.OfficialStaticNerf(cfg).to(dev)
    names = ["layers0.0", "layers0.2", "layers0.4", "layers0.6", "layers1.0", "layers1.2", "layers1.4", "layers1.6", "fc_density", "fc_feature",
             "rgb_layers.0", "fc_rgb"]
    mods = dict(net.named_modules())
    weights = [mods[n].weight for n in names]
    biases = [mods[n].bias for n in names]
    g = torch.Generator(device="cpu").manual_seed(7)
    o = (0.3 * torch.randn(R, 3, generator=g)).to(dev)
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1).to(dev)
    v = (-d).contiguous()
    s = torch.linspace(0.0, 1.0, N)
    z = 0.01 * (1 - s) + 4.0 * s
    mid = 0.5 * (z[1:] + z[:-1])
    lo, hi = torch.cat([z[:1], mid]).to(dev), torch.cat([mid, z[-1:]]).to(dev)
    u = torch.rand(R, N, generator=g).to(dev) if jit else None

    def run(ws):
        os.environ["NNR_FWD_WS"] = "1" if ws else "0"
        with torch.no_grad():
            out = ops.render_rays(o, d, v, lo, hi, u, weights, biases, hidden=256, dist_alpha=da, white_bg=wb, relu_sigma=False,
                                  samples=not fused)
        torch.cuda.synchronize()
        return [t.clone() if t is not None else None for t in out]

    ref = run(False)
    got = run(True)
    res = {"shape": [R, N], "jitter": jit, "fused": fused}
    for name, a, b in zip(("rgb", "dist", "alpha", "z"), ref, got):
        if a is None:
            continue
        res["maxdiff_" + name] = float((a - b).abs().max())
        res["nan_" + name] = bool(torch.isnan(b).any())
    if reps:
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(2 * reps)]
        for w in (False, True):      # warm both
            run(w)
        for i in range(2 * reps):
            ws = i % 2 == 1
            os.environ["NNR_FWD_WS"] = "1" if ws else "0"
            with torch.no_grad():
                ev[i][0].record()
                ops.render_rays(o, d, v, lo, hi, u, weights, biases, hidden=256, dist_alpha=da, white_bg=wb, relu_sigma=False, samples=not fused)
                ev[i][1].record()
        torch.cuda.synchronize()
        ms = [e[0].elapsed_time(e[1]) for e in ev]
        res["ms_product"] = sorted(ms[0::2])[reps // 2]
        res["ms_ws"] = sorted(ms[1::2])[reps // 2]
        res["ms_product_min"] = min(ms[0::2])
        res["ms_ws_min"] = min(ms[1::2])
    print("WSCHECK " + json.dumps(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
        sys.exit(0)
    timing_only = "--timing-only" in sys.argv
    ok = True
    for i, sh in enumerate(SHAPES):
        if timing_only and sh[0] < 1024:
            continue
        reps = 20 if sh[0] >= 1024 else 0
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(i), str(reps)], capture_output=True, text=True, timeout=240)
            lines = [l for l in r.stdout.splitlines() if l.startswith("WSCHECK ")]
            if r.returncode != 0 or not lines:
                ok = False
                print("shape", sh, "FAILED rc", r.returncode, (r.stderr or "")[-1500:], flush=True)
                continue
            res = json.loads(lines[0][8:])
            bad = any(v for k, v in res.items() if k.startswith("nan_")) or any(v > 2e-5 for k, v in res.items() if k.startswith("maxdiff_"))
            ok &= not bad
            print(("BAD " if bad else "ok  ") + json.dumps(res), "%.0f s" % (time.time() - t0), flush=True)
        except subprocess.TimeoutExpired:
            ok = False
            print("shape", sh, "TIMEOUT (hang?)", flush=True)
            break       # a hung kernel may have left the GPU unusable: stop here
    print("ws_check:", "PASS" if ok else "FAIL")
    sys.exit(0 if ok else 1)
