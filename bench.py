#!/usr/bin/env python
"""bench.py -- training rays/s of the NoPe-NeRF render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One *step* = one `model.Trainer.train_step` over one synthetic image batch resident in HBM: pose / distortion forward,
random pixel pick, ray generation, stratified sampling, fused MLP forward, compositing, rgb-L1 + depth-L1 loss heads,
full backward to the MLP / pose / distortion gradients, (N>1: one flat RCCL all-reduce), and the three Adam steps.
The per-image auxiliary losses (point-cloud NN, surface reprojection) are SURVEY.md section 8(f) "next" rows and are
switched off (their weights anneal to 0 in the reference too).

Workload = BASELINE.json configs[1]: 1024 rays x (64+128) samples per GPU, 8-layer-256 MLP, pose + distortion learnable,
fp32.  The reference has no coarse/fine resampling (SURVEY.md header), so "64 coarse + 128 fine" is pinned as 192
stratified samples per ray in a single pass.  Weak scaling: every rank renders its own 1024-ray shard of a
(1024*N)-ray step.

The JSON line carries `roofline` (dominant fused-MLP kernel timed with HIP events on the launch stream; algorithmic
FLOPs = 1 186 816 per sample per pass) and `cpu_baseline` (the CPU oracle -- a port of the
reference's PyTorch path -- timed on this host's cores on a bounded sample of the same workload).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))

import numpy as np
import torch
import torch.distributed as dist

R_PER_GPU, N_SAMPLES, HIDDEN = 1024, 192, 256
IMG_H, IMG_W, N_CAMS = 540, 960, 16
FLOP_PER_SAMPLE_PASS = 2 * 593408            # forward == dgrad == wgrad, BASELINE.md section 2
PEAK_FP32_MFMA_TFLOPS = 157.3                # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU x 256 CU x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0               # dense bf16 MFMA (not the 2:1-sparsity headline figure)
PEAK_HBM_GBS = 8000.0                        # HBM3E


def full_cfg(rays_total, aux=False, bf16=False):
    cfg = {
        'model': {'hidden_dim': HIDDEN, 'pos_enc_levels': 10, 'dir_enc_levels': 4, 'occ_activation': 'softplus'},
        'rendering': {'type': 'nope_nerf', 'n_max_network_queries': 64000, 'white_background': False, 'radius': 4.0,
                      'num_points': N_SAMPLES, 'depth_range': [0.01, 10], 'dist_alpha': False, 'use_ray_dir': True,
                      'normalise_ray': True, 'normal_loss': False, 'sample_option': 'uniform', 'outside_steps': 0},
        'depth': {'type': 'None'},
        'distortion': {'fix_scaleN': True},
        'training': {
            'type': 'nope_nerf', 'n_training_points': rays_total, 'vis_geo': False, 'detach_gt_depth': False, 'pc_ratio': 4,
            'match_method': 'dense', 'shift_first': False, 'detach_ref_img': True, 'scale_pcs': True,
            'detach_rgbs_scale': False, 'vis_reprojection_every': 5000, 'nearest_limit': 0.01, 'annealing_epochs': 2000,
            'rgb_weight': [1.0, 1.0], 'depth_weight': [0.04, 0.0], 'pc_weight': [0.0, 0.0], 'rgb_s_weight': [0.0, 0.0],
            'depth_consistency_weight': [0.0, 0.0], 'weight_dist_2nd_loss': [0.0, 0.0], 'weight_dist_1st_loss': [0.0, 0.0],
            'depth_loss_type': 'l1', 'with_ssim': False, 'with_auto_mask': False,
        },
    }
    if bf16:  # BASELINE configs[2] arithmetic: bf16 MFMA products, fp32 accumulation (not the headline metric)
        cfg['rendering']['mfma_dtype'] = 'bf16'
    if aux:   # configs/default.yaml:99-100 -- the first training phase: point-cloud + surface-reprojection losses on
        cfg['training']['pc_weight'] = [1.0, 0.0]
        cfg['training']['rgb_s_weight'] = [1.0, 0.0]
    return cfg


def synthetic_batch(device, seed=42):
    g = torch.Generator().manual_seed(seed)
    f = 0.7 * IMG_W
    K = torch.diag(torch.tensor([2 * f / IMG_W, -2 * f / IMG_H, -1.0, 1.0])).unsqueeze(0)
    return {
        'img': torch.rand(1, 3, IMG_H, IMG_W, generator=g).to(device),
        'img.idx': 3,
        'img.dpt': (1 + 2 * torch.rand(1, IMG_H, IMG_W, generator=g)).to(device),
        'img.camera_mat': K.to(device),
        'img.scale_mat': torch.eye(4).unsqueeze(0).to(device),
        # the neighbouring frame the per-image losses compare against (dataloading: ref_imgs / ref_dpts / ref_idxs)
        'img.ref_imgs': torch.rand(1, 3, IMG_H, IMG_W, generator=g).to(device),
        'img.ref_dpts': (1 + 2 * torch.rand(1, IMG_H, IMG_W, generator=g)).to(device),
        'img.ref_idxs': 4,
    }


def build_trainer(device, world, aux=False, bf16=False):
    import model as mdl
    cfg = full_cfg(R_PER_GPU * world, aux, bf16)
    torch.manual_seed(42)
    net = mdl.OfficialStaticNerf(cfg)
    model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device=device), cfg, device=device)
    pose = mdl.LearnPose(N_CAMS, True, True, cfg).to(device)
    distn = mdl.Learn_Distortion(N_CAMS, True, True, cfg).to(device)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        pose.r.copy_(0.01 * torch.randn(N_CAMS, 3, generator=g))
        pose.t.copy_(0.01 * torch.randn(N_CAMS, 3, generator=g))
        distn.global_scales.copy_(1 + 0.05 * torch.randn(N_CAMS, 1, generator=g))
        distn.global_shifts.copy_(0.05 * torch.randn(N_CAMS, 1, generator=g))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt_pose = torch.optim.Adam(pose.parameters(), lr=5e-4)
    opt_dist = torch.optim.Adam(distn.parameters(), lr=5e-4)
    trainer = mdl.Trainer(model, opt, cfg['training'], device=device, optimizer_pose=opt_pose, pose_param_net=pose,
                          optimizer_distortion=opt_dist, distortion_net=distn)
    return trainer, net


def _hbm_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r01/hbm_traffic.json: rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE, separate runs of tools/profile_kernels.py at this very shape).  Counters cannot be
    collected from inside this process; None if the summary is not there."""
    path = os.path.join(ROOT, 'profiles', 'r01', 'hbm_traffic.json')
    if not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)['kernels']
    key = {'mlp_fwd': 'mlp_fwd_kernel<256, true, false>', 'mlp_dgrad': 'mlp_dgrad_kernel<256, false>', 'mlp_wgrad': 'wgrad_kernel<false>'}[kernel]
    for name, v in table.items():
        if key in name:
            return int(v['fetch_bytes'] + v['write_bytes'])
    return None


def kernel_roofline(net, device, reps=5, bf16=False):
    """Time the three fused-MLP kernels of one 1024x192 step individually with HIP events on the launch stream.
    bf16=True (bench.py --bf16, not the headline): the bf16-product kernels, whose bound is HBM, not the matrix pipe."""
    from nnr import lib as L
    from nnr import ops
    lib = L.load()
    R, N, D = R_PER_GPU, N_SAMPLES, HIDDEN
    cfg = L.make_cfg(R, N, D, train=True, bf16=bf16)
    g = torch.Generator().manual_seed(1)
    d = torch.randn(R, 3, generator=g)
    d = (d / d.norm(dim=-1, keepdim=True)).to(device)
    o = torch.zeros(R, 3, device=device)
    z = torch.linspace(0, 1, N)
    z = 0.01 * (1 - z) + 10 * z
    mid = 0.5 * (z[1:] + z[:-1])
    lo, hi = torch.cat([z[:1], mid]).to(device), torch.cat([mid, z[-1:]]).to(device)
    jit = torch.rand(R, N, generator=g).to(device)
    w, b = net.weights(), net.biases()
    packed = ops._packed_for(cfg, w, b)
    ws = torch.empty(lib.nnr_workspace_floats(C.byref(cfg)), device=device)
    rgb, dst = torch.empty(R, 3, device=device), torch.empty(R, device=device)
    d_rgb, d_dst = torch.randn(R, 3, device=device) / R, torch.randn(R, device=device) / R
    gw = [torch.zeros_like(x) for x in w]
    gb = [torch.zeros_like(x) for x in b]
    gs = L.params_struct(gw, gb)
    plan = ops._plan_for(cfg, device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    view = (-d).contiguous()
    stages = {
        'mlp_fwd': lambda: lib.nnr_mlp_fwd(C.byref(cfg), L.ptr(o), L.ptr(d), L.ptr(view), L.ptr(lo), L.ptr(hi), L.ptr(jit),
                                           L.ptr(packed), L.ptr(ws), st),
        'composite_fwd': lambda: lib.nnr_composite_fwd(C.byref(cfg), L.ptr(rgb), L.ptr(dst), None, None, L.ptr(ws), st),
        'composite_bwd': lambda: lib.nnr_composite_bwd(C.byref(cfg), L.ptr(d_rgb), L.ptr(d_dst), L.ptr(ws), st),
        'mlp_dgrad': lambda: lib.nnr_mlp_dgrad(C.byref(cfg), L.ptr(packed), L.ptr(ws), st),
        'mlp_wgrad': lambda: lib.nnr_mlp_wgrad(C.byref(cfg), L.ptr(packed), C.byref(gs), L.ptr(plan), L.ptr(ws), st),
    }
    cfg_inf = L.make_cfg(R, N, D, bf16=bf16)      # forward-only variant (eval / visualisation): no stash
    ws_inf = torch.empty(lib.nnr_workspace_floats(C.byref(cfg_inf)), device=device)
    stages['mlp_fwd_infer'] = lambda: lib.nnr_mlp_fwd(C.byref(cfg_inf), L.ptr(o), L.ptr(d), L.ptr(view), L.ptr(lo), L.ptr(hi),
                                                      L.ptr(jit), L.ptr(packed), L.ptr(ws_inf), st)
    times = {}
    for name, fn in stages.items():
        L.check(fn(), name)            # warm-up + makes the workspace valid for the next stage
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.check(fn(), name)
        e1.record()
        torch.cuda.synchronize()
        times[name] = e0.elapsed_time(e1) / reps   # ms per launch
    flops = FLOP_PER_SAMPLE_PASS * R * N
    per = {k: {'ms': round(v, 4), 'tflops': round(flops / (v * 1e-3) / 1e12, 2) if k.startswith('mlp_') else None}
           for k, v in times.items()}
    dom = max(('mlp_fwd', 'mlp_dgrad', 'mlp_wgrad'), key=lambda k: times[k])
    achieved = flops / (times[dom] * 1e-3) / 1e12
    mlp_ms = times['mlp_fwd'] + times['mlp_dgrad'] + times['mlp_wgrad']
    if bf16:
        # Algorithmic HBM bytes per sample of the bf16-mode kernels: hidden activations and their
        # gradients are bf16 planes (8 x D + D/2 elements each), encodings (64 + 32) and the 4-wide planes fp32, masks 1 bit
        # per activation.  fwd writes the stash; dgrad reads masks + dout4 + encodings and writes the gradient planes;
        # wgrad reads activations + gradients + encodings + dout4 once.
        act = (8 * D + D // 2) * 2
        enc, masks, four = (64 + 32) * 4, 9 * 2 * (D // 64) * 4, 16
        byts = {'mlp_fwd': act + enc + masks + four + 4, 'mlp_dgrad': act + masks + four + enc + 2 * four,
                'mlp_wgrad': 2 * act + enc + four}
        gbs = {k: byts[k] * R * N / (times[k] * 1e-3) / 1e9 for k in byts}
        dom = max(byts, key=lambda k: times[k])
        for k in byts:
            per[k]['gbytes_per_s'] = round(gbs[k], 1)
        return {'bound': 'hbm', 'kernel': dom, 'achieved': round(gbs[dom], 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'frac': round(gbs[dom] / PEAK_HBM_GBS, 4), 'traffic': None, 'bytes_per_launch': byts[dom] * R * N, 'kernels': per,
                'fused_mlp_all_three': {'ms': round(mlp_ms, 4), 'tflops': round(3 * flops / (mlp_ms * 1e-3) / 1e12, 2),
                                        'frac_of_bf16_mfma_peak': round(3 * flops / (mlp_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4)}}
    return {
        'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), 'traffic': _hbm_traffic(dom),
        'flop_per_launch': flops, 'kernels': per,
        'fused_mlp_all_three': {'ms': round(mlp_ms, 4), 'tflops': round(3 * flops / (mlp_ms * 1e-3) / 1e12, 2),
                                'frac': round(3 * flops / (mlp_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
    }


def cpu_baseline(sample_rays=192, steps=3):
    """The CPU oracle (oracle/nerf_oracle.py, a port of the reference's PyTorch path) on this host's cores, on a bounded
    sample of the same workload: `sample_rays` rays x 192 samples, D=256, forward + loss heads + backward."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as orc
    cores = min(os.cpu_count() or 1, 32)   # the GEMMs here are (12k x 256) x (256 x 256): more threads only add sync cost
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    params = {k: v.requires_grad_(True) for k, v in orc.init_params(HIDDEN, 1).items()}
    pose_r = (0.01 * torch.randn(N_CAMS, 3, generator=g)).requires_grad_(True)
    pose_t = (0.01 * torch.randn(N_CAMS, 3, generator=g)).requires_grad_(True)
    scales = (1 + 0.05 * torch.randn(N_CAMS, 1, generator=g)).requires_grad_(True)
    shifts = (0.05 * torch.randn(N_CAMS, 1, generator=g)).requires_grad_(True)
    f = 0.7 * IMG_W
    K = torch.diag(torch.tensor([2 * f / IMG_W, -2 * f / IMG_H, -1.0, 1.0])).unsqueeze(0)
    depth = 1 + 2 * torch.rand(1, 1, IMG_H, IMG_W, generator=g)
    img = torch.rand(1, 3, IMG_H, IMG_W, generator=g)
    rcfg = full_cfg(sample_rays)['rendering']
    rcfg['occ_activation'] = 'softplus'
    ts = []
    for i in range(steps + 1):
        ray_idx = torch.randperm(IMG_H * IMG_W, generator=g)[:sample_rays]
        jitter = torch.rand(1, sample_rays, N_SAMPLES, generator=g)
        t0 = time.perf_counter()
        loss, _ = orc.train_step_scope(params, pose_r, pose_t, scales, shifts, 3, K, depth, img, (IMG_H, IMG_W), ray_idx,
                                       jitter, rcfg)
        loss.backward()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts[1:]))
    return {'value': round(sample_rays / med, 2), 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
            'sample': f'{sample_rays} rays x {N_SAMPLES} samples, D={HIDDEN}, fwd+loss+bwd, median of {steps} steps after 1 warm-up, '
                      f'torch {torch.__version__} CPU, {cores} threads'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--aux', action='store_true',
                    help='also run the per-image point-cloud / reprojection losses of the first training phase (not the headline metric)')
    ap.add_argument('--bf16', action='store_true',
                    help='bf16-MFMA mode of the MLP forward / input-gradient kernels (BASELINE configs[2] arithmetic; not the headline metric)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the HIP render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1"

    trainer, net = build_trainer(device, world, args.aux, args.bf16)
    data = synthetic_batch(device)

    def step(i):
        return trainer.train_step(data, it=i, epoch=0, scheduling_start=10000, render_path=None)

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ld = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(ld['loss'].detach())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        rays = R_PER_GPU * world
        out = {
            'metric': 'training rays/sec', 'value': round(rays / (ms * 1e-3), 1), 'unit': 'rays/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16 products / f32 accumulate' if args.bf16 else 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: 1024 rays/GPU x 192 samples (64 coarse + 128 fine pinned as one '
                                   '192-sample stratified pass), 8-layer-256 MLP, pose + distortion learnable, fp32; '
                                   'full Trainer.train_step incl. 3 Adam steps; aux per-image losses ' + ('ON (pc + rgb_s)' if args.aux else 'off'),
                       'rays_per_gpu': R_PER_GPU, 'n_samples': N_SAMPLES, 'hidden': HIDDEN, 'image': [IMG_H, IMG_W],
                       'parallelism': f'dp{world} (ray-sharded, one flat all-reduce)'},
            'final_loss': round(loss_val, 6),
        }
        out['roofline'] = kernel_roofline(net, device, bf16=args.bf16)
        out['cpu_baseline'] = None if (args.no_cpu_baseline or world > 1) else cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
