"""Profiling helper: run the fused-MLP kernels with a -DNNR_TIMELINE library (NNR_LIB) and print the per-stage shader-clock
deltas of one mid-grid wave (forward: train then inference)."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
import torch

import bench

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    import model as mdl
    from nnr import lib as nnrlib
    cfg = bench.full_cfg(bench.R_PER_GPU)
    torch.manual_seed(42)
    net = mdl.OfficialStaticNerf(cfg).to(dev)
    # python tools/timeline.py [R N [bf16]]
    R, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (bench.R_PER_GPU, bench.N_SAMPLES)
    bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"
    out = bench.kernel_roofline(net, dev, reps=2, bf16=bf16, rays=R, n_samples=N)
    print(json.dumps(out["kernels"]))
    handle = ctypes.CDLL(nnrlib.LIB_PATH)
    for name in ("nnr_timeline_fwd", "nnr_timeline_dgrad", "nnr_timeline_fwd16", "nnr_timeline_dgrad16", "nnr_timeline_wgrad"):
        if not hasattr(handle, name):
            continue
        buf = (ctypes.c_ulonglong * 32)()
        torch.cuda.synchronize()
        rc = getattr(handle, name)(buf)
        for base, tag in ((0, "train"), (16, "infer")):
            v = [int(x) for x in buf[base:base + 16]]
            if not any(v):
                continue
            if name.endswith("wgrad"):
                print(name, "loop", v[1] - v[0], "flush", v[2] - v[1], "samples", v[4], "MI*8+NI", v[5],
                      "cycles/16 samples (ideal %d)" % (64 * 4 * (v[5] // 8) * (v[5] % 8)), (v[1] - v[0]) / max(v[4], 1) * 16)
                continue
            n = max(i for i, x in enumerate(v) if x) + 1
            print(name, tag, "rc", rc, "total", v[n - 1] - v[0], "deltas", [v[i + 1] - v[i] for i in range(n - 1)])
    if hasattr(handle, "nnr_timeline_wgrad_all") and not bf16:
        import numpy as np
        buf = (ctypes.c_ulonglong * 4096)()
        handle.nnr_timeline_wgrad_all(buf)
        v = np.array(buf, dtype=np.int64).reshape(-1, 2)
        v = v[v[:, 0] > 0]
        t0 = v[:, 0].min()
        dur = v[:, 1] - v[:, 0]
        print("wgrad waves", len(v), "start spread", int(v[:, 0].max() - t0), "end max", int(v[:, 1].max() - t0),
              "dur min/mean/max", int(dur.min()), int(dur.mean()), int(dur.max()))
        nb = len(v) // 4
        per_blk = dur[:nb * 4].reshape(nb, 4).max(axis=1)
        order = np.argsort(per_blk)
        print("slowest blocks", [(int(b), int(per_blk[b])) for b in order[-6:]], "fastest", [(int(b), int(per_blk[b])) for b in order[:4]])
        from nnr import ops
        from nnr import lib as L2
        jobs, first = ops.plan_jobs(L2.make_cfg(bench.R_PER_GPU, bench.N_SAMPLES, bench.HIDDEN, train=True), with_waves=True)
        import collections
        bytype = collections.defaultdict(list)
        for wv in range(len(first) - 1):
            js = jobs[first[wv]:first[wv + 1]]
            key = tuple(sorted({(j.MI, j.NI, j.bias) for j in js}))
            cost = sum(j.MI * j.NI * (j.k1 - j.k0) // 16 for j in js)
            bytype[key].append(dur[wv] / max(cost, 1))
        for k, x in sorted(bytype.items()):
            print("class B/A wave types", k, "n", len(x), "cycles per cost-granule min/mean/max %.1f %.1f %.1f" % (min(x), sum(x) / len(x), max(x)))
        # the slowest and the fastest workgroups: which jobs their four waves ran, and how long each wave took
        for b in list(order[-8:]) + list(order[:4]):
            desc = []
            for wv in range(4 * b, 4 * b + 4):
                js = jobs[first[wv]:first[wv + 1]]
                desc.append("%d:" % dur[wv] + "+".join("(%d,%d,b%d,L%d,%d)" % (j.MI, j.NI, j.bias, j.layer, (j.k1 - j.k0) // 16) for j in js))
            print("block", int(b), int(per_blk[b]), " | ".join(desc))
    if hasattr(handle, "nnr_timeline_fwd16_all") and bf16:
        import numpy as np
        buf = (ctypes.c_ulonglong * (3 * 4096))()
        handle.nnr_timeline_fwd16_all(buf)
        v = np.array(buf, dtype=np.int64).reshape(-1, 3)
        v = v[v[:, 0] > 0]
        t0 = v[:, 0].min()
        dur = v[:, 1] - v[:, 0]
        print("fwd16 workgroups", len(v), "kernel span", int(v[:, 1].max() - t0), "dur min/mean/max", int(dur.min()), int(dur.mean()), int(dur.max()))
        hw = v[:, 2]
        cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
        key = se * 64 + sh * 16 + cu
        starts = np.sort(v[:, 0] - t0)
        print("start times of the first/last 8 workgroups", starts[:8].tolist(), starts[-8:].tolist())
        ends = np.sort(v[:, 1] - t0)
        print("end time percentiles 10/50/90/100", [int(np.percentile(ends, q)) for q in (10, 50, 90, 100)])
        # gaps between consecutive workgroups on the same (se, sh, cu) -- XCDs share ids, so this mixes 8 of them; still shows the idle time
        order = np.argsort(v[:, 0])
        print("durations of the first 4 and last 4 started", dur[order[:4]].tolist(), dur[order[-4:]].tolist())
