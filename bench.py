#!/usr/bin/env python
"""bench.py -- training rays/s of the NoPe-NeRF render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 starts itself: without WORLD_SIZE in the environment the script re-executes under `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU, RCCL); launched by torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as usual.

One *step* = one `model.Trainer.train_step` over one synthetic image batch resident in HBM: pose / distortion forward,
random pixel pick, ray generation, stratified sampling, fused MLP forward, compositing, rgb-L1 + depth-L1 loss heads,
full backward to the MLP / pose / distortion gradients, (N>1: one flat RCCL all-reduce), and the three Adam steps.
The per-image auxiliary losses (point-cloud NN, surface reprojection) are SURVEY.md section 8(f) "next" rows and are
switched off in the headline (their weights anneal to 0 in the reference too); `--aux` turns them on.

Headline workload = BASELINE.json configs[1]: 1024 rays x (64+128) samples per GPU, 8-layer-256 MLP, pose + distortion
learnable, fp32.  The reference has no coarse/fine resampling (SURVEY.md header), so "64 coarse + 128 fine" is pinned as 192
stratified samples per ray in a single pass.  Weak scaling: every rank renders its own shard of a (rays_per_gpu * N)-ray step.

The ONE JSON line carries
  * `roofline`      -- the dominant fused-MLP kernel timed with HIP events on the launch stream; algorithmic FLOPs =
                       1 186 816 per sample per pass;
  * `cpu_baseline`  -- the CPU oracle (a port of the reference's PyTorch path) timed on this host's cores on a bounded
                       sample of the same workload (2 warm-ups + 5 timed steps);
  * `configs`       -- (N = 1 only) the other BASELINE.json configurations measured in the same run: `bf16_4096x128`
                       (configs[2]: bf16 MFMA products, fp32 accumulate), `fp32_1024x128` (the reference's stock default
                       N = 128) and `cpu_32x64_d128` (configs[0]: the reference's own CPU-runnable case, full step with the
                       per-image losses on, timed on the host through the oracle).
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))

import numpy as np
import torch
import torch.distributed as dist

R_PER_GPU, N_SAMPLES, HIDDEN = 1024, 192, 256     # BASELINE configs[1]
IMG_H, IMG_W, N_CAMS = 540, 960, 16
MACS_PER_SAMPLE = {256: 593408, 128: 157440}   # sum of in x out over the 12 nn.Linear
FLOP_PER_SAMPLE_PASS = 2 * MACS_PER_SAMPLE[256]   # forward == dgrad == wgrad
# MACs the kernels EXECUTE per sample and pass: fc_feature folded into the colour-hidden layer by the pack kernel (an exact algebraic
# rewrite, DESIGN.md section 4.1 "merged layer": -D*D), odd widths padded inside the MFMA fragments (63 -> 64, 27 -> 32)
EXECUTED_MACS_PER_SAMPLE = {D: 64 * D + 3 * D * D + (D + 64) * D + 3 * D * D + D + (D + 32) * (D // 2) + (D // 2) * 3 for D in (128, 256)}
PEAK_FP32_MFMA_TFLOPS = 157.3                # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU x 256 CU x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0               # dense bf16 MFMA (not the 2:1-sparsity headline figure)
PEAK_HBM_GBS = 8000.0                        # HBM3E
# BASELINE configs[2]: "bf16 MFMA with fp32 accumulate, tolerance vs fp32 ref" -- the tolerance, stated.  Measured at the benchmark shape itself
# (4096 x 128, D = 256; 256-ray subset against the fp32 oracle, tests/test_gpu_bench_shape_parity.py prints the figures into the suite log):
# outputs to a few 1e-4 absolute, gradient tensors 5 - 15 % in relative L2 -- bf16 products flip the ReLU gates of near-zero pre-activations.
# Against an oracle with the SAME arithmetic (every product bf16 x bf16, fp32 accumulate) the kernels are held to 1e-4 / 2.5e-2.
BF16_TOLERANCE_VS_FP32 = {'rgb_max_abs': 5e-4, 'depth_max_abs': 2e-3, 'grad_rel_l2': 0.25,
                          'measured_at': '4096 x 128, D = 256, 256-ray subset vs the fp32 oracle (tests/test_gpu_bench_shape_parity.py)'}
FP32_HOW = {'mfma': 'fp32 (fp32 MFMAs)', 'split3': 'fp32 via six bf16 MFMA terms per product',
            'split2': 'fp32 via three fp16 MFMA terms per product (fwd, dgrad, wgrad 4x4 + encoding tiles)'}      # (short: the driver's record cuts strings at ~100 characters)
FP32_NOTE = {'mfma': 'v_mfma_f32_32x32x2_f32 products in all three MLP kernels',
             'split2': 'fp32 results: every product of the forward, the input gradient and the 4 x 4 weight-gradient tiles as three fp16 MFMA terms of '
                       'two-term operands (power-of-two scaled, residual at 2^11: csrc/nnr_split2.h; as close to fp64 as fp32 MFMAs: '
                       'tests/test_gpu_split3.py), fp32 accumulate; the 128 x 64 encoding tiles of the weight gradient likewise, its other narrow tiles on fp32 MFMAs',
             'split3': 'fp32 results: every product of the three MLP kernels as six bf16 MFMA terms of three-term (exact) operands, fp32 accumulate '
                       '(as close to fp64 as fp32 MFMAs: tests/test_gpu_split3.py); the narrow weight-gradient tiles on fp32 MFMAs'}


def full_cfg(rays_total, aux=False, bf16=False, n_samples=None, hidden=None):
    cfg = {
        'model': {'hidden_dim': hidden or HIDDEN, 'pos_enc_levels': 10, 'dir_enc_levels': 4, 'occ_activation': 'softplus'},
        'rendering': {'type': 'nope_nerf', 'n_max_network_queries': 64000, 'white_background': False, 'radius': 4.0,
                      'num_points': n_samples or N_SAMPLES, 'depth_range': [0.01, 10], 'dist_alpha': False, 'use_ray_dir': True,
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
    if bf16:  # BASELINE configs[2] arithmetic: bf16 MFMA products, fp32 accumulation
        cfg['rendering']['mfma_dtype'] = 'bf16'
    if aux:   # configs/default.yaml:99-100 -- the first training phase: point-cloud + surface-reprojection losses on
        cfg['training']['pc_weight'] = [1.0, 0.0]
        cfg['training']['rgb_s_weight'] = [1.0, 0.0]
    return cfg


def synthetic_batch(device, seed=42, depth_hw=None):
    g = torch.Generator().manual_seed(seed)
    f = 0.7 * IMG_W
    K = torch.diag(torch.tensor([2 * f / IMG_W, -2 * f / IMG_H, -1.0, 1.0])).unsqueeze(0)
    dh, dw = depth_hw or (IMG_H, IMG_W)
    # the batch of a scene that is RESIDENT in HBM, as dataloading.ResidentLoader serves it (section 5 of DESIGN.md: inputs are device-resident):
    # frame and neighbour are views of one scene tensor, the camera matrix is the scene's own tensor -- both flagged the way the loader flags
    # them, so that the trainer may keep per-frame constants (the resized frames of the per-image losses, the inverse camera matrix)
    frames = torch.rand(2, 3, IMG_H, IMG_W, generator=g).to(device)
    frames._nnr_resident = True
    Kd = K.to(device)
    Kd._nnr_resident = True
    return {
        'img': frames[0:1],
        'img.idx': 3,
        'img.dpt': (1 + 2 * torch.rand(1, dh, dw, generator=g)).to(device),
        'img.camera_mat': Kd,
        'img.scale_mat': torch.eye(4).unsqueeze(0).to(device),
        # the neighbouring frame the per-image losses compare against (dataloading: ref_imgs / ref_dpts / ref_idxs)
        'img.ref_imgs': frames[1:2],
        'img.ref_dpts': (1 + 2 * torch.rand(1, dh, dw, generator=g)).to(device),
        'img.ref_idxs': 4,
    }


def build_trainer(device, world, aux=False, bf16=False, rays_per_gpu=None, n_samples=None, hidden=None):
    import model as mdl
    cfg = full_cfg((rays_per_gpu or R_PER_GPU) * world, aux, bf16, n_samples, hidden)
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


_TRAFFIC_FILES = ('profiles/r06/hbm_traffic.json', 'profiles/r06/hbm_traffic_bf16_4096x128.json', 'profiles/r06/hbm_traffic_fp32_1024x128.json',
                  'profiles/r05/hbm_traffic.json', 'profiles/r05/hbm_traffic_bf16_4096x128.json', 'profiles/r05/hbm_traffic_fp32_1024x128.json',
                  'profiles/r04/hbm_traffic.json', 'profiles/r04/hbm_traffic_bf16_4096x128.json',
                  'profiles/r03/hbm_traffic.json', 'profiles/r03/hbm_traffic_bf16_4096x128.json',
                  'profiles/r02/hbm_traffic.json', 'profiles/r02/hbm_traffic_bf16_4096x128.json')
_KERNEL_KEYS = {
    False: {'mlp_fwd': 'mlp_fwd_kernel<256, true', 'mlp_dgrad': 'mlp_dgrad_kernel<256', 'mlp_wgrad': 'nnr::wgrad_kernel'},
    True: {'mlp_fwd': 'mlp_fwd_bf16_kernel<256, true,', 'mlp_dgrad': 'mlp_dgrad_bf16_kernel<256,', 'mlp_wgrad': 'wgrad_b_kernel'},
}


def _hbm_traffic(kernel, bf16=False, shape=None):
    """(HBM bytes per launch of `kernel`, where the number comes from).  PMC counters cannot be collected from inside this process:
    the figure is read from the committed summary of separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of
    tools/profile_kernels.py at the very shape named in the file (FETCH_SIZE doubled per MI355X_MICROARCH.md).  (None, None) if no
    summary holds this kernel at this shape."""
    for rel in _TRAFFIC_FILES:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            table = json.load(f)
        if shape is not None and tuple(table.get('shape', (1024, 192))) != tuple(shape):
            continue
        if bool(table.get('bf16', False)) != bool(bf16):
            continue
        key = _KERNEL_KEYS[bool(bf16)][kernel]
        if not bf16 and kernel != 'mlp_wgrad':      # the fp32 mode's forward / input gradient have one kernel per product mode: take the running one's figure only
            from nnr import lib as nnr_lib
            if nnr_lib.fp32_products() == 'split2':
                key = key.replace('_kernel<', '_f16_kernel<')      # nnr::mlp_fwd_f16_kernel<256, true> / nnr::mlp_dgrad_f16_kernel<256>
        for name, v in table['kernels'].items():
            if key in name:
                return int(v['fetch_bytes'] + v['write_bytes']), rel + ' (offline PMC passes, not measured in this run)'
    return None, None


def _pmc_clock(kernel, bf16=False, shape=None):
    """Shader clock and MFMA-busy share of `kernel` from the committed SQ counter pass of the same binaries (profiles/r06/pmc_clock.json:
    GRBM_GUI_ACTIVE / 8 XCDs / kernel duration) -- an OFFLINE figure, like `traffic`; the live clock of this run is `roofline.clock`."""
    path = os.path.join(ROOT, 'profiles', 'r06', 'pmc_clock.json')
    if bf16 or not os.path.exists(path):
        return None
    with open(path) as f:
        table = json.load(f)
    if shape is not None and tuple(table.get('shape', (1024, 192))) != tuple(shape):
        return None
    key = _KERNEL_KEYS[False][kernel] if kernel in _KERNEL_KEYS[False] else None
    if key is None:
        return None
    from nnr import lib as nnr_lib
    if kernel != 'mlp_wgrad' and nnr_lib.fp32_products() == 'split2':
        key = key.replace('_kernel<', '_f16_kernel<')
    for name, v in table['kernels'].items():
        if key in name and not (kernel == 'mlp_fwd' and 'false>' in name):
            return dict(v, source='profiles/r06/pmc_clock.json (offline PMC pass of the same binaries, isolated launches)')
    return None


def box_probe(device, gib=1, reps=8):
    """What this box's HBM delivers to plain streaming kernels, next to the step it just timed: boxes of the pool differ -- twice in ~60 runs of
    round 5 the forward and the input gradient (the two kernels that WRITE 1.7-1.9 GB of stash each) took 1.63 / 1.46 ms instead of 0.96 / 0.89
    while the weight gradient (a reader) was unchanged, a 4.6 ms step on otherwise identical binaries.  A 1 GiB fill (write) and a 1 GiB sum
    (read), `reps` each between two events; outside the timed region."""
    n = gib * (1 << 28)
    x = torch.empty(n, dtype=torch.float32, device=device)
    res = {}
    for name, fn in (('write', lambda: x.fill_(1.0)), ('read', lambda: x.sum())):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        res['hbm_%s_GBps' % name] = round(4.0 * n * reps / (a.elapsed_time(b) * 1e-3) / 1e9, 1)
    del x
    res['what'] = '%d GiB torch fill_ / sum on this box after the timed steps (a streaming write / read of HBM; not part of any timed region)' % gib
    return res


def _sclk_file(device):
    """hwmon `freq1_input` (shader clock, Hz) of the card `device` is, located through its PCI bus id in sysfs; None where the box does not
    expose it (other cards of the node are visible in /sys/class/drm but not readable from inside the container)."""
    import glob
    try:
        bus = '%02x' % torch.cuda.get_device_properties(device).pci_bus_id
    except Exception:
        bus = None
    readable = []
    for card in sorted(glob.glob('/sys/class/drm/card*/device')):
        if not os.access(os.path.join(card, 'pp_dpm_sclk'), os.R_OK):
            continue
        f = [x for x in glob.glob(os.path.join(card, 'hwmon', '*', 'freq1_input')) if os.access(x, os.R_OK)]
        if f:
            readable.append((os.path.realpath(card), f[0]))
    for real, f in readable:
        if bus is not None and (':%s:' % bus) in real.rsplit('/', 1)[-1]:
            return f
    return readable[0][1] if len(readable) == 1 else None


def clock_probe(device, step, seconds=0.5, period=0.004):
    """The shader clock the chip holds UNDER the training step: `step()` repeated for `seconds` (untimed, after the timed region) while a
    thread reads the card's hwmon shader-clock file every `period`.  The roofline peaks are quoted at the nominal 2.4 GHz; this is what the
    same binaries had available on this box."""
    import threading
    f = _sclk_file(device)
    if f is None:
        return None
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            try:
                samples.append(int(open(f).read()) / 1e6)
            except Exception:
                pass
            time.sleep(period)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t = threading.Thread(target=sampler, daemon=True)
    t0 = time.time()
    t.start()
    n = 0
    while time.time() - t0 < seconds:
        step()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    stop.set()
    t.join()
    if not samples:
        return None
    samples.sort()
    return {'sclk_mhz_median': round(samples[len(samples) // 2], 1), 'sclk_mhz_min': round(samples[0], 1), 'sclk_mhz_max': round(samples[-1], 1),
            'samples': len(samples), 'steps': n, 'nominal_mhz': 2400,
            'what': 'hwmon freq1_input of this card sampled every %g ms while the training step repeats for %g s after the timed region' % (period * 1e3, seconds)}


def kernel_roofline(net, device, reps=5, bf16=False, rays=None, n_samples=None, in_step=None, sequence_reps=0):
    """Roofline block of the three fused-MLP kernels.  `in_step` = their mean durations INSIDE the timed training steps (HIP events
    on the launch stream around every launch, nnr_prof_begin / nnr_prof_end: _timed_steps): the basis of `achieved` when given.
    Each kernel is also timed in isolation (`reps` back-to-back launches between two events; reported as `isolated_ms`): that
    figure is systematically slower for the kernels that write the stash -- five forwards in a row push 10 GB to HBM and the chip
    clocks down, which the same kernel between the step's other kernels does not see.
    bf16=True: the bf16-product kernels, whose bound is HBM, not the matrix pipe."""
    from nnr import lib as L
    from nnr import ops
    lib = L.load()
    R, N, D = rays or R_PER_GPU, n_samples or N_SAMPLES, net.hidden_dim
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
    packed_inf = ops._packed_for(cfg_inf, w, b)
    # the product inference path (imaging.render_full_image): compositing in the kernel's epilogue, 16 bytes written per ray
    stages['mlp_fwd_infer'] = lambda: lib.nnr_render_fwd(C.byref(cfg_inf), L.ptr(o), L.ptr(d), L.ptr(view), L.ptr(lo), L.ptr(hi),
                                                         L.ptr(jit), L.ptr(packed_inf), L.ptr(rgb), L.ptr(dst), None, None, L.ptr(ws_inf), st)
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
    isolated = dict(times)
    sequence = None
    if sequence_reps:      # the five stages in the order of a training step, `sequence_reps` times, an event pair around every launch: what a
                           # kernel costs BETWEEN the others (clock, L2 and HBM state of a step) without a Trainer -- for experiment libraries
        order = ['mlp_fwd', 'composite_fwd', 'composite_bwd', 'mlp_dgrad', 'mlp_wgrad']
        marks = {k: [] for k in order}
        for _ in range(sequence_reps + 2):
            for k in order:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                L.check(stages[k](), k)
                e1.record()
                marks[k].append((e0, e1))
        torch.cuda.synchronize()
        sequence = {k: round(float(np.mean([a.elapsed_time(b) for a, b in v[2:]])), 4) for k, v in marks.items()}
    if in_step:
        times.update({k: in_step[k] for k in ('mlp_fwd', 'mlp_dgrad', 'mlp_wgrad')})
    how = ('HIP events on the launch stream around every launch of the kernel inside the %d timed training steps' % in_step['launches']
           if in_step else 'HIP events around %d back-to-back launches of the kernel alone' % reps)
    flops = 2 * MACS_PER_SAMPLE[D] * R * N
    executed = 2 * EXECUTED_MACS_PER_SAMPLE[D] * R * N
    per = {k: {'ms': round(v, 4), 'tflops': round(flops / (v * 1e-3) / 1e12, 2) if k.startswith('mlp_') else None,
               'executed_tflops': round(executed / (v * 1e-3) / 1e12, 2) if k.startswith('mlp_') else None,
               'isolated_ms': round(isolated[k], 4)} for k, v in times.items()}
    if sequence:
        for k, v in sequence.items():
            per[k]['sequence_ms'] = v
    dom = max(('mlp_fwd', 'mlp_dgrad', 'mlp_wgrad'), key=lambda k: times[k])
    achieved = flops / (times[dom] * 1e-3) / 1e12
    mlp_ms = times['mlp_fwd'] + times['mlp_dgrad'] + times['mlp_wgrad']
    three = {'ms': round(mlp_ms, 4), 'tflops': round(3 * flops / (mlp_ms * 1e-3) / 1e12, 2)}
    if bf16:
        # Algorithmic HBM bytes per sample of the bf16-mode kernels: hidden activations and their
        # gradients are bf16 planes (8 x D + D/2 elements each), encodings (64 + 32) and the 4-wide planes fp32, masks 1 bit
        # per activation.  fwd writes the stash; dgrad reads masks + dout4 + encodings and writes the gradient planes;
        # wgrad reads activations + gradients + encodings + dout4 once.
        byts = bf16_bytes_per_sample(D)
        gbs = {k: byts[k] * R * N / (times[k] * 1e-3) / 1e9 for k in byts}
        dom = max(byts, key=lambda k: times[k])
        for k in byts:
            per[k]['gbytes_per_s'] = round(gbs[k], 1)
            per[k]['bytes_per_sample'] = byts[k]
            per[k]['frac_of_hbm_peak'] = round(gbs[k] / PEAK_HBM_GBS, 4)      # every kernel against ITS bound, not only the dominant one
        traffic, src = _hbm_traffic(dom, True, (R, N))
        three['frac_of_bf16_mfma_peak'] = round(three['tflops'] / PEAK_BF16_MFMA_TFLOPS, 4)
        three['gbytes_per_s'] = round(sum(byts.values()) * R * N / (mlp_ms * 1e-3) / 1e9, 1)
        three['frac_of_hbm_peak'] = round(three['gbytes_per_s'] / PEAK_HBM_GBS, 4)
        for k in byts:
            per[k]['frac_of_bf16_mfma_peak'] = round(per[k]['executed_tflops'] / PEAK_BF16_MFMA_TFLOPS, 4)
        # SURVEY 8(d) prices this mode against the dense bf16 MFMA peak; the HBM figure of the same kernel stands beside it (the stash the
        # kernels' own layout defines is what they actually wait on: DESIGN 4.2)
        return {'bound': 'mfma', 'kernel': dom, 'achieved': per[dom]['executed_tflops'], 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(per[dom]['executed_tflops'] / PEAK_BF16_MFMA_TFLOPS, 4),
                'frac_algorithmic': round(per[dom]['tflops'] / PEAK_BF16_MFMA_TFLOPS, 4),
                'what': 'executed bf16 MFMA work (executed MACs x 2) / in-step time of the dominant kernel vs the dense bf16 peak',
                'hbm': {'achieved': round(gbs[dom], 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(gbs[dom] / PEAK_HBM_GBS, 4),
                        'bytes_per_launch': byts[dom] * R * N, 'what': 'algorithmic stash bytes of the same kernel / its duration'},
                'traffic': traffic, 'traffic_source': src, 'timing': how, 'flop_per_launch': flops, 'executed_flop_per_launch': executed,
                'kernels': per, 'fused_mlp_all_three': three}
    traffic, src = _hbm_traffic(dom, False, (R, N))
    products = L.fp32_products()
    if products == 'mfma':      # (term products run on the 16-bit pipe: an "fp32 MFMA fraction" of them would read above 1)
        three['frac'] = round(three['tflops'] / PEAK_FP32_MFMA_TFLOPS, 4)
    if products in ('split3', 'split2'):
        # Every fp32 product as 16-bit MFMA terms: six bf16 terms of three-term operands (csrc/nnr_split.h), or -- forward / input gradient of
        # 'split2' -- three fp16 terms of two-term operands (csrc/nnr_split2.h).  The kernels' bound is the 16-bit matrix pipe (bf16 and fp16:
        # the same dense 2.5 PFLOP/s), the work they issue is terms x the executed MACs.  Forward / input gradient: all of it; weight gradient
        # (six bf16 terms, or -- 'split2' at D = 256, the workgroup jobs of nnr_wgrad.hip wgrad_group_split2 -- three fp16 terms): the 4 x 4 tiles
        # (480 of the 528 tile-units of MFMA work at D = 256), the narrow tiles stay on fp32 MFMAs.
        # (round 6, two-term mode: + the 128 x 64 tiles against the position encoding, 32 more tile-units, as private two-term jobs -- unless
        # NNR_WGRAD_ENC2_WEIGHT=0 leaves them on fp32 MFMAs)
        enc2 = products == 'split2' and D == 256 and os.environ.get('NNR_WGRAD_ENC2_WEIGHT', '') != '0' and not os.environ.get('NNR_WGRAD_BF16_TERMS')
        share = {'mlp_fwd': 1.0, 'mlp_dgrad': 1.0, 'mlp_fwd_infer': 1.0, 'mlp_wgrad': ((512.0 if enc2 else 480.0) / 528.0) if D == 256 else 0.0}
        terms = {k: (6 if (products == 'split3' or (k == 'mlp_wgrad' and os.environ.get('NNR_WGRAD_BF16_TERMS'))) else 3) for k in share}
        for k, f in share.items():
            issued = terms[k] * f * executed / (times[k] * 1e-3) / 1e12
            per[k].update(mfma=('bf16, 6 terms per fp32 product' if terms[k] == 6 else 'fp16, 3 terms per fp32 product')
                          + ('' if f == 1.0 else ' (4 x 4 tiles%s: %.0f %% of the MACs; the other narrow tiles fp32)' % (' and the encoding tiles' if f > 0.95 else '', 100 * f)), terms_per_product=terms[k],
                          issued_bf16_tflops=round(issued, 1), frac_of_bf16_mfma_peak=round(issued / PEAK_BF16_MFMA_TFLOPS, 4),
                          # SURVEY 8(d)'s own figure beside it: algorithmic FLOPs / time, and its ratio to the fp32 MATRIX peak -- above 1 where the
                          # work does not run on that pipe, which is the point of the term products
                          algorithmic_tflops=round(flops / (times[k] * 1e-3) / 1e12, 2),
                          frac_of_fp32_matrix_peak=round(flops / (times[k] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4))
        three['issued_bf16_tflops'] = round(sum(terms[k] * share[k] * executed for k in ('mlp_fwd', 'mlp_dgrad', 'mlp_wgrad')) / (mlp_ms * 1e-3) / 1e12, 1)
        three['frac_of_bf16_mfma_peak'] = round(three['issued_bf16_tflops'] / PEAK_BF16_MFMA_TFLOPS, 4)
        three['frac_of_fp32_matrix_peak'] = round(three['tflops'] / PEAK_FP32_MFMA_TFLOPS, 4)
        issued_dom = per[dom]['issued_bf16_tflops']
        if products == 'split2':      # (the committed counter pass is of the default build)
            for k in ('mlp_fwd', 'mlp_dgrad', 'mlp_wgrad'):
                pc = _pmc_clock(k, False, (R, N))
                if pc is not None:      # the matrix-pipe fraction against the peak at the clock the kernel itself held in the PMC pass
                    per[k]['pmc'] = pc
                    per[k]['frac_of_bf16_mfma_peak_at_pmc_clock'] = round(per[k]['issued_bf16_tflops'] / (PEAK_BF16_MFMA_TFLOPS * pc['ghz'] / 2.4), 4)
        # ... and every kernel against the OTHER roof: algorithmic HBM bytes (the fp32 stash planes each kernel writes / reads exactly once: DESIGN 4.4 / 4.6)
        byts = fp32_bytes_per_sample(D)
        for k in byts:
            gb = byts[k] * R * N / (times[k] * 1e-3) / 1e9
            per[k].update(bytes_per_sample=byts[k], gbytes_per_s=round(gb, 1), frac_of_hbm_peak=round(gb / PEAK_HBM_GBS, 4))
        hbm = {'achieved': per[dom]['gbytes_per_s'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': per[dom]['frac_of_hbm_peak'],
               'bytes_per_launch': byts[dom] * R * N, 'what': 'algorithmic stash bytes of the same kernel (distinct planes, each once) / its in-step duration'}
        mfma = {'achieved': issued_dom, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(issued_dom / PEAK_BF16_MFMA_TFLOPS, 4),
                'what': 'issued 16-bit MFMA work (%d terms x executed MACs x 2%s) / in-step time vs the dense bf16 / fp16 peak'
                        % (terms[dom], '' if share[dom] == 1.0 else ' of the tiles on the 16-bit pipe')}
        if hbm['frac'] > mfma['frac']:
            # the dominant kernel sits nearer the HBM roof than the matrix pipe's (round 6: the weight gradient, once its products took three fp16 terms)
            return {
                'fp32_products': products, 'bound': 'hbm', 'kernel': dom, 'achieved': hbm['achieved'], 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                'frac': hbm['frac'], 'what': hbm['what'], 'bytes_per_launch': hbm['bytes_per_launch'], 'mfma': mfma,
                'notes': 'the nearer of the two roofs of the dominant kernel; its matrix-pipe figure is in `mfma`, every kernel carries both fractions '
                         '(`frac_of_hbm_peak`, `frac_of_bf16_mfma_peak`); `traffic` = what the kernel actually fetched + wrote (PMC): above the algorithmic '
                         'bytes by the planes its narrow tiles read again (DESIGN 4.6d)',
                'algorithmic_tflops': round(achieved, 2), 'frac_of_fp32_matrix_peak': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                'traffic': traffic, 'traffic_source': src, 'timing': how, 'flop_per_launch': flops, 'executed_flop_per_launch': executed,
                'issued_bf16_flop_per_launch': int(terms[dom] * share[dom] * executed), 'kernels': per, 'fused_mlp_all_three': three,
            }
        return {
            'fp32_products': products, 'bound': 'mfma', 'kernel': dom, 'achieved': issued_dom, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(issued_dom / PEAK_BF16_MFMA_TFLOPS, 4), 'hbm': hbm,
            'what': 'issued 16-bit MFMA work (%d terms x executed MACs x 2) / in-step time vs the dense bf16 / fp16 peak' % terms[dom],
            'notes': 'peak at the nominal 2.4 GHz; the chip holds ~1.8 GHz under these kernels (DESIGN 4.3).  frac_algorithmic = terms x ALGORITHMIC '
                     'MACs x 2 / time / peak (SURVEY 8d counts algorithmic work; the kernels execute 89 %% of it: feature layer folded).  '
                     'algorithmic_tflops (= fp32_equivalent_tflops) = algorithmic FLOPs / time; frac_of_fp32_matrix_peak = that / %.1f, above 1 by construction: '
                     'the products do not run on the fp32 matrix pipe' % PEAK_FP32_MFMA_TFLOPS,
            'frac_algorithmic': round(terms[dom] * share[dom] * flops / (times[dom] * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4),
            'algorithmic_tflops': round(achieved, 2), 'frac_of_fp32_matrix_peak': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
            'fp32_equivalent_tflops': round(achieved, 2), 'traffic': traffic, 'traffic_source': src, 'timing': how,
            'flop_per_launch': flops, 'executed_flop_per_launch': executed, 'issued_bf16_flop_per_launch': int(terms[dom] * share[dom] * executed),
            'kernels': per, 'fused_mlp_all_three': three,
        }
    return {
        'fp32_products': products,
        'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 2), 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
        'frac': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), 'frac_algorithmic': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
        'what': 'algorithmic FLOPs (SURVEY 8d) / in-step time of the dominant kernel vs the fp32 MFMA peak', 'traffic': traffic, 'traffic_source': src, 'timing': how,
        'flop_per_launch': flops, 'executed_flop_per_launch': executed, 'executed_tflops': round(executed / (times[dom] * 1e-3) / 1e12, 2),
        'executed_frac': round(executed / (times[dom] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), 'kernels': per, 'fused_mlp_all_three': three,
    }


def fp32_bytes_per_sample(D):
    """Algorithmic HBM bytes per sample of the three fp32-mode kernels with term products (tile-major fp32 stash planes: DESIGN 4.4): hidden
    activations h1..h8 (8 D) and the colour-hidden layer g (D / 2) as fp32, their gradients likewise, the encodings (64 + 32) as fp32, ReLU
    gates 1 bit per activation, (rgb, sigma) 16 B, z 4 B, jitter 4 B, position + view direction 2 x 16 B, d point + d view 2 x 16 B; the
    weight gradient also reads the 4-wide output gradients."""
    h, g, dg = 8 * D * 4, (D // 2) * 4, (D // 2) * 4
    enc, gates = (64 + 32) * 4, 9 * D // 8
    fwd = h + g + enc + gates + 16 + 4 + 32 + 4
    dgrad = h + dg + 32 + gates + 16 + 32
    wgrad = 2 * h + g + dg + enc + 16
    return {'mlp_fwd': fwd, 'mlp_dgrad': dgrad, 'mlp_wgrad': wgrad}


def bf16_bytes_per_sample(D):
    """Algorithmic HBM bytes per sample of the three bf16-mode kernels = the planes each one writes / reads exactly once (DESIGN.md
    section 4.2, round 3): hidden activations h1..h8 (8 D) and the colour-hidden layer g (D / 2) as bf16; their gradients likewise, the
    colour gradient with one extra 16-wide group (the output gradients as bf16); bf16 copies of the two encodings (64 + 32); ReLU gates
    1 bit per activation (9 layers x D / 8 bytes); (rgb, sigma) 16 B, z 4 B, jitter 4 B; position + view direction 2 x 16 B (what the
    input-gradient kernel recomputes the chain-rule factors from: the 384 B of fp32 factor planes of round 2 are gone); d point + d
    view 2 x 16 B."""
    h, g, dg = 8 * D * 2, (D // 2) * 2, (D // 2 + 16) * 2
    enc16, gates = (64 + 32) * 2, 9 * D // 8
    fwd = h + g + enc16 + gates + 16 + 4 + 32 + 4
    dgrad = h + dg + 32 + gates + 16 + 32
    wgrad = 2 * h + g + dg + enc16
    return {'mlp_fwd': fwd, 'mlp_dgrad': dgrad, 'mlp_wgrad': wgrad}


def _host_cpu():
    model = ''
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return os.cpu_count() or 1, model


def cpu_baseline_reference(rays=R_PER_GPU, warmup=2, steps=5):
    """The REFERENCE ITSELF on this host's cores (`kind: "reference"`): tools/cpu_reference_baseline.py, in its own process, runs the
    staged, unmodified copy of /root/reference/model (gpurun_stage/reference_cpu/, git-ignored, put there by tools/stage_reference.py /
    __graft_entry__.build() in the authoring container) -- nope_nerf.forward + the reference's loss heads + backward on the headline
    workload, 2 warm-ups + 5 timed steps, median.  None when nothing is staged or the worker fails (the caller falls back to the port)."""
    stage = os.path.join(ROOT, 'gpurun_stage', 'reference_cpu')
    if not os.path.isfile(os.path.join(stage, 'model', 'rendering.py')):
        return None, 'no staged reference (gpurun_stage/reference_cpu/model missing)'
    cmd = [sys.executable, os.path.join(ROOT, 'tools', 'cpu_reference_baseline.py'), '--rays', str(rays), '--samples', str(N_SAMPLES),
           '--hidden', str(HIDDEN), '--warmup', str(warmup), '--steps', str(steps)]
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'PYTHONPATH')}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        if r.returncode != 0 or not lines:
            return None, 'reference worker failed: ' + (r.stderr.strip().splitlines() or ['?'])[-1][:300]
        return json.loads(lines[-1]), None
    except Exception as e:        # noqa: BLE001 -- a baseline must never take the benchmark down
        return None, 'reference worker failed: %r' % (e,)


def cpu_baseline(sample_rays=192, warmup=2, steps=5):
    """`cpu_baseline` of the JSON line: the reference's own code when it is staged (cpu_baseline_reference: kind "reference", the
    whole 1024-ray headline step), otherwise the oracle port on a bounded sample (cpu_baseline_port: kind "port"); the block says
    which ran and, for the port, why."""
    ref, why = cpu_baseline_reference()
    if ref is not None:
        port = cpu_baseline_port(sample_rays, warmup, steps)      # the oracle beside it: the port runs at the reference's speed
        ref['oracle_port_cross_check'] = {k: port[k] for k in ('value', 'unit', 'sample')}
        return ref
    out = cpu_baseline_port(sample_rays, warmup, steps)
    out['fallback_reason'] = why
    return out


def cpu_baseline_port(sample_rays=192, warmup=2, steps=5):
    """The CPU oracle (oracle/nerf_oracle.py, a port of the reference's PyTorch path: the reference itself cannot travel to the GPU
    box) on this host's cores, on a bounded sample of the headline workload: `sample_rays` rays x 192 samples, D=256, forward +
    loss heads + backward; median of `steps` timed steps after `warmup` untimed ones."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as orc
    host_cores, cpu_model = _host_cpu()
    threads = min(host_cores, 32)   # the GEMMs here are (37k x 256) x (256 x 256): more threads only add sync cost
    torch.set_num_threads(threads)
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
    for i in range(warmup + steps):
        ray_idx = torch.randperm(IMG_H * IMG_W, generator=g)[:sample_rays]
        jitter = torch.rand(1, sample_rays, N_SAMPLES, generator=g)
        t0 = time.perf_counter()
        loss, _ = orc.train_step_scope(params, pose_r, pose_t, scales, shifts, 3, K, depth, img, (IMG_H, IMG_W), ray_idx,
                                       jitter, rcfg)
        loss.backward()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts[warmup:]))
    return {'value': round(sample_rays / med, 2), 'unit': 'rays/s', 'cores': threads, 'host_cores': host_cores, 'threads': threads,
            'cpu': cpu_model, 'kind': 'port',
            'sample': f'{sample_rays} rays x {N_SAMPLES} samples, D={HIDDEN}, fwd+loss+bwd, median of {steps} steps after {warmup} '
                      f'warm-ups, torch {torch.__version__} CPU, {threads} threads on a {host_cores}-core host'}


def cpu_config0(warmup=1, steps=3):
    """BASELINE configs[0] -- the reference's own CPU-runnable case -- on the REFERENCE ITSELF when it is staged (`kind: "reference"`):
    tools/cpu_reference_baseline.py --train-step runs the staged, unmodified /root/reference/model package's `Trainer.train_step`
    (model/training.py:67-97) with configs/Tanks/Ignatius.yaml over configs/default.yaml, 32 rays x 64 samples, hidden_dim 128, the
    first-phase per-image losses on, three Adam steps, for a small sweep of thread counts (2 k samples per step do not feed 32 threads) and
    reports the best; otherwise the oracle port (`kind: "port"`, with the reason)."""
    stage = os.path.join(ROOT, 'gpurun_stage', 'reference_cpu')
    why = 'no staged reference (gpurun_stage/reference_cpu/model missing)'
    if os.path.isfile(os.path.join(stage, 'model', 'training.py')) and os.path.isfile(os.path.join(stage, 'configs', 'Tanks', 'Ignatius.yaml')):
        cmd = [sys.executable, os.path.join(ROOT, 'tools', 'cpu_reference_baseline.py'), '--train-step', '--rays', '32', '--samples', '64',
               '--hidden', '128', '--warmup', str(warmup), '--steps', str(steps)]
        env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'PYTHONPATH')}
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
            lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
            if r.returncode == 0 and lines:
                return json.loads(lines[-1])
            why = 'reference worker failed: ' + (r.stderr.strip().splitlines() or ['?'])[-1][:300]
        except Exception as e:        # noqa: BLE001 -- a baseline must never take the benchmark down
            why = 'reference worker failed: %r' % (e,)
    out = cpu_config0_port(warmup, steps)
    out['fallback_reason'] = why
    return out


def cpu_config0_port(warmup=1, steps=3, threads=None):
    """The same case through the oracle port (fallback when no reference is staged): forward + backward of the render slice and the
    per-image terms on 108 x 192 mono-depth maps (a 27 x 48 grid).  threads=None: the thread counts the reference path sweeps (1, 2, 4, 8,
    min(cores, 32)) are each timed and the FASTEST is reported, with its thread count -- a slow baseline flatters every ratio built on it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as orc
    host_cores, cpu_model = _host_cpu()
    if threads is None:
        sweep = sorted({t for t in (1, 2, 4, 8, min(host_cores, 32)) if t <= host_cores})
        runs = [cpu_config0_port(warmup, max(2, steps - 1), t) for t in sweep]
        best = max(runs, key=lambda r: r['value'])
        best['thread_sweep'] = {str(r['threads']): r['value'] for r in runs}
        return best
    torch.set_num_threads(threads)
    R, N, D, dh, dw = 32, 64, 128, 108, 192
    g = torch.Generator().manual_seed(0)
    params = {k: v.requires_grad_(True) for k, v in orc.init_params(D, 1).items()}
    pose_r = (0.01 * torch.randn(N_CAMS, 3, generator=g)).requires_grad_(True)
    pose_t = (0.01 * torch.randn(N_CAMS, 3, generator=g)).requires_grad_(True)
    scales = (1 + 0.05 * torch.randn(N_CAMS, 1, generator=g)).requires_grad_(True)
    shifts = (0.05 * torch.randn(N_CAMS, 1, generator=g)).requires_grad_(True)
    f = 0.7 * IMG_W
    K = torch.diag(torch.tensor([2 * f / IMG_W, -2 * f / IMG_H, -1.0, 1.0])).unsqueeze(0)
    depth, depth_ref = (1 + 2 * torch.rand(1, 1, dh, dw, generator=g) for _ in range(2))
    img, img_ref = (torch.rand(1, 3, IMG_H, IMG_W, generator=g) for _ in range(2))
    rcfg = full_cfg(R, n_samples=N, hidden=D)['rendering']
    rcfg['occ_activation'] = 'softplus'
    ts = []
    for i in range(warmup + steps):
        ray_idx = torch.randperm(IMG_H * IMG_W, generator=g)[:R]
        jitter = torch.rand(1, R, N, generator=g)
        t0 = time.perf_counter()
        loss, _ = orc.train_step_scope(params, pose_r, pose_t, scales, shifts, 3, K, depth, img, (IMG_H, IMG_W), ray_idx, jitter, rcfg)
        aux, _, _ = orc.aux_scope(pose_r, pose_t, scales, shifts, 3, 4, K, depth, depth_ref, img, img_ref)
        (loss + aux).backward()
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts[warmup:]))
    return {'workload': 'BASELINE configs[0]: Tanks/Ignatius settings, 32 rays x 64 samples, D=128, full step incl. per-image losses '
                        '(27x48 grid), CPU', 'ms_per_step': round(med * 1e3, 3), 'value': round(R / med, 1), 'unit': 'rays/s',
            'kind': 'port', 'threads': threads, 'host_cores': host_cores, 'cpu': cpu_model,
            'sample': f'median of {steps} steps after {warmup} warm-ups'}


def _timed_steps(trainer, data, warmup, steps, use_dist):
    def step(i):
        return trainer.train_step(data, it=i, epoch=0, scheduling_start=10000, render_path=None)
    ld = None
    for i in range(warmup):
        ld = step(i)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    from nnr import lib as L
    lib = L.load()
    L.check(lib.nnr_prof_begin(steps), 'nnr_prof_begin')      # two event records per MLP kernel launch: microseconds of host time per step
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]      # one record per step on the step's stream, no sync:
    t0 = time.perf_counter()                                                      # the spread of the timed steps (the region is ~70 ms)
    marks[0].record()
    for i in range(steps):
        ld = step(warmup + i)
        marks[i + 1].record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms4, n4 = (C.c_float * 4)(), (C.c_int32 * 4)()
    L.check(lib.nnr_prof_end(ms4, n4), 'nnr_prof_end')
    in_step = {'mlp_fwd': float(ms4[0]), 'mlp_dgrad': float(ms4[1]), 'mlp_wgrad': float(ms4[2]), 'launches': int(n4[0])}
    per_step = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(steps)], dtype=np.float64)
    in_step['step_ms'] = {'mean': round(float(per_step.mean()), 4), 'std': round(float(per_step.std()), 4), 'min': round(float(per_step.min()), 4),
                          'max': round(float(per_step.max()), 4), 'median': round(float(np.median(per_step)), 4),
                          'how': 'GPU time between event records at the end of consecutive timed steps (this rank)'}
    if use_dist:
        t = torch.tensor([elapsed], device=data['img'].device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    trainer.flush_nan_check()
    return elapsed, float(ld['loss'].detach()), in_step


def extra_config(device, name, rays, n_samples, bf16, steps=10, warmup=3, products=None, aux=False):
    """One more BASELINE configuration measured in the same run on one GPU: full Trainer.train_step + the kernels' roofline.
    products: 'mfma' / 'split3' for this block only (nnr.lib.set_fp32_products)."""
    from nnr import lib as L
    prev = L.set_fp32_products(products) if products else None
    try:
        return _extra_config(device, name, rays, n_samples, bf16, steps, warmup, aux)
    finally:
        if prev:
            L.set_fp32_products(prev)


def _extra_config(device, name, rays, n_samples, bf16, steps, warmup, aux=False):
    from nnr import lib as L
    trainer, net = build_trainer(device, 1, aux, bf16, rays, n_samples)
    data = synthetic_batch(device)
    elapsed, loss, in_step = _timed_steps(trainer, data, warmup, steps, False)
    ms = elapsed / steps * 1e3
    roof = kernel_roofline(net, device, reps=3, bf16=bf16, rays=rays, n_samples=n_samples, in_step=in_step)
    out = {'workload': name, 'rays_per_gpu': rays, 'n_samples': n_samples, 'hidden': HIDDEN,
           'dtype': 'bf16 products / f32 accumulate' if bf16 else 'f32', 'fp32_products': None if bf16 else L.fp32_products(),
           'steps': steps, 'warmup': warmup, 'aux_per_image_losses': bool(aux),
           'ms_per_step': round(ms, 4), 'step_ms': in_step['step_ms'], 'value': round(rays / (ms * 1e-3), 1), 'unit': 'rays/s',
           'final_loss': round(loss, 6),
           'roofline': {k: roof[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'frac_algorithmic', 'what', 'traffic', 'traffic_source', 'timing') if k in roof},
           'kernels_ms': {k: v['ms'] for k, v in roof['kernels'].items()},
           'kernels_isolated_ms': {k: v['isolated_ms'] for k, v in roof['kernels'].items()}, 'fused_mlp_all_three': roof['fused_mlp_all_three']}
    if 'hbm' in roof:
        out['roofline']['hbm'] = roof['hbm']
    if bf16:
        out['tolerance_vs_fp32'] = BF16_TOLERANCE_VS_FP32
    for k in ('algorithmic_tflops', 'frac_of_fp32_matrix_peak'):
        if k in roof:
            out['roofline'][k] = roof[k]
    if aux:     # everything of the step that is not one of the three MLP kernels: the per-image block + the small launches
        mlp = sum(out['kernels_ms'][k] for k in ('mlp_fwd', 'mlp_dgrad', 'mlp_wgrad'))
        out['outside_the_three_mlp_kernels_ms'] = round(ms - mlp, 4)
    del trainer, net, data
    torch.cuda.empty_cache()
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_launch(n, script=None, argv=None, capture=False):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run (one per GPU).
    script / argv: another rank script and its arguments through the SAME launch line (the 8-rank CPU dry run of the data-parallel step,
    tests/dp_dryrun_worker.py); capture: return (returncode, stdout) instead of the return code."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(script or __file__)] + (sys.argv[1:] if argv is None else list(argv))
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '4')
    if capture:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        return r.returncode, r.stdout + ('\n' + r.stderr[-3000:] if r.returncode else '')
    return subprocess.call(cmd, env=env)


def allreduce_probe(device, n_floats, reps=20):
    """The step's one collective on its own: SUM all-reduce of a flat fp32 bucket of the gradient size, microseconds per call
    (max over ranks), and the number of ranks that actually took part (the sum of ones)."""
    sync = torch.cuda.synchronize if torch.device(device).type == 'cuda' else (lambda: None)      # (a CPU group in the dry-run tests)
    buf = torch.ones(n_floats, device=device)
    dist.all_reduce(buf)
    ranks_seen = int(round(float(buf[0].item())))
    buf.fill_(1.0)
    sync()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(buf)
    sync()
    us = (time.perf_counter() - t0) / reps * 1e6
    t = torch.tensor([us], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return ranks_seen, float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rays-per-gpu', type=int, default=R_PER_GPU, help='rays each rank renders per step (BASELINE configs[3]: 4096)')
    ap.add_argument('--total-rays', type=int, default=0,
                    help='STRONG scaling: the step is this many rays in total, every rank renders total / N of them (the default is weak scaling: '
                         '--rays-per-gpu on every rank); e.g. --total-rays 32768 --gpus 8 = BASELINE configs[3]')
    ap.add_argument('--samples', type=int, default=N_SAMPLES, help='samples per ray (headline 192 = 64 + 128; stock default 128)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the extra `configs` blocks (bf16 4096x128, fp32 N=128, CPU configs[0])')
    ap.add_argument('--aux', action='store_true',
                    help='also run the per-image point-cloud / reprojection losses of the first training phase (not the headline metric)')
    ap.add_argument('--bf16', action='store_true',
                    help='bf16-MFMA mode of the MLP kernels for the MAIN measurement (BASELINE configs[2] arithmetic; not the headline metric)')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / rendezvous check without a GPU: every rank joins a gloo group, all-reduces and rank 0 prints the line skeleton')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')

    if args.dry_run:
        if world > 1:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            t = torch.ones(1)
            dist.all_reduce(t)
            seen = int(t.item())
            dist.barrier()
            dist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            print(json.dumps({'metric': 'training rays/sec', 'value': None, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
                              'warmup': args.warmup, 'dry_run': True, 'ranks_seen': seen, 'backend': 'gloo'}), flush=True)
        return

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the HIP render path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    shared = world > n_dev
    if shared and os.environ.get('NNR_ALLOW_SHARED_GPU') != '1':
        raise SystemExit(f"--gpus {world} but only {n_dev} GPU(s) visible (set NNR_ALLOW_SHARED_GPU=1 to share them through gloo)")
    torch.cuda.set_device(local_rank % n_dev)
    device = torch.device('cuda', local_rank % n_dev)
    backend = 'gloo' if shared else 'nccl'      # RCCL refuses two ranks on one device; gloo carries device tensors through the host
    # NNR_BENCH_FORCE_DIST=1: join the process group, all-reduce the gradients and report the `collective` block at world size 1
    # too -- the RCCL leg of this script (init with device_id, the step's all-reduce, the probe) on a one-GPU box (tests/test_gpu_bench_ranks.py)
    use_dist = world > 1 or os.environ.get('NNR_BENCH_FORCE_DIST') == '1'
    if use_dist:
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        if world == 1:
            os.environ['NNR_DP_ALWAYS_REDUCE'] = '1'
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)

    R, N = args.rays_per_gpu, args.samples
    if args.total_rays:
        if args.total_rays % (4 * world):
            raise SystemExit(f"--total-rays {args.total_rays} must be a multiple of 4 x the number of ranks ({world})")
        R = args.total_rays // world
    trainer, net = build_trainer(device, world, args.aux, args.bf16, R, N)
    data = synthetic_batch(device)
    elapsed, loss_val, in_step = _timed_steps(trainer, data, args.warmup, args.steps, use_dist)
    ranks_seen, ar_us = (1, None)
    if use_dist:
        n_grad = sum(p.numel() for m in (net, trainer.pose_param_net, trainer.distortion_net) for p in m.parameters()) + 9
        ranks_seen, ar_us = allreduce_probe(device, n_grad)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        rays = R * world
        from nnr import lib as nnr_lib
        fp32_products = nnr_lib.fp32_products()
        trainer_adam = getattr(trainer, 'adam_arithmetic', 'single')
        headline = (R, N) == (R_PER_GPU, N_SAMPLES) and not args.bf16 and not args.total_rays
        what = ('BASELINE configs[1]: 1024 rays/GPU x 192 samples' if headline else f'{R} rays/GPU x {N} samples')
        out = {
            'metric': 'training rays/sec', 'value': round(rays / (ms * 1e-3), 1), 'unit': 'rays/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms, 4), 'step_ms': in_step['step_ms'], 'higher_is_better': True,
            'scaling': 'strong' if args.total_rays else 'weak', 'vs_baseline': None, 'dtype': 'bf16 products / f32 accumulate' if args.bf16 else 'f32', 'data': 'synthetic',
            'config': {'workload': what + ', D=256, ' + ('bf16 MFMA' if args.bf16 else FP32_HOW[fp32_products]),
                       'workload_notes': '64 coarse + 128 fine pinned as ONE 192-sample stratified pass (the reference has no resampling); 8-layer-256 MLP, pose + '
                                         'distortion learnable; full Trainer.train_step incl. the three Adam updates (arithmetic: ' + trainer_adam + '); per-image '
                                         'losses ' + ('ON (pc + rgb_s)' if args.aux else 'off') + '; ' + ('bf16 MFMA products' if args.bf16 else FP32_NOTE[fp32_products]),
                       'fp32_products': None if args.bf16 else fp32_products, 'adam_arithmetic': trainer_adam,
                       'rays_per_gpu': R, 'total_rays': rays, 'n_samples': N, 'hidden': HIDDEN, 'image': [IMG_H, IMG_W],
                       'parallelism': f'dp{world} (ray-sharded, one flat all-reduce)'},
            'final_loss': round(loss_val, 6),
        }
        if use_dist:
            out['collective'] = {'backend': 'rccl' if backend == 'nccl' else 'gloo (shared GPU dry run)', 'rccl_ranks_seen': ranks_seen,
                                 'allreduce_us': round(ar_us, 1), 'bucket_floats': n_grad}
        out['roofline'] = kernel_roofline(net, device, bf16=args.bf16, rays=R, n_samples=N, in_step=in_step)
        # (one process only: under N ranks the step holds a collective, and a time-bounded loop on rank 0 alone would leave the others waiting)
        clk = None if use_dist else clock_probe(device, lambda: trainer.train_step(data, it=args.warmup + args.steps, epoch=0, scheduling_start=10000,
                                                                                   render_path=None))
        out['roofline']['clock'] = clk
        # the matrix-pipe fraction against what the chip had at the clock it held (the top-level block, or its `mfma` part where HBM is the nearer roof)
        blk = out['roofline'] if out['roofline'].get('unit') == 'TFLOP/s' else out['roofline'].get('mfma')
        if clk is not None and blk is not None:
            at = blk['peak'] * clk['sclk_mhz_median'] / clk['nominal_mhz']
            blk['peak_at_measured_clock'] = round(at, 1)
            blk['frac_at_measured_clock'] = round(blk['achieved'] / at, 4)
        out['box'] = box_probe(device)
        out['cpu_baseline'] = None if (args.no_cpu_baseline or world > 1) else cpu_baseline()
        if world == 1 and not args.no_extra and headline:
            del trainer
            torch.cuda.empty_cache()
            out['configs'] = {
                'bf16_4096x128': extra_config(device, 'BASELINE configs[2]: 4096 rays x 128 samples, bf16 MFMA products with fp32 accumulate',
                                              4096, 128, True),
                'fp32_1024x128': extra_config(device, "the reference's stock default: 1024 rays x 128 samples (configs/default.yaml:37,76), fp32",
                                              1024, 128, False),
                # the headline shape with v_mfma_f32_32x32x2_f32 products in all three kernels (the fp32 path of rounds 1-2)
                'fp32_mfma_1024x192': extra_config(device, 'BASELINE configs[1] with fp32-MFMA products (NNR_FP32_PRODUCTS=mfma)',
                                                   R_PER_GPU, N_SAMPLES, False, products='mfma') if fp32_products != 'mfma' else None,
                # ... and with the six-term bf16 products of rounds 3-5 in all three kernels (NNR_FP32_PRODUCTS=split3)
                'fp32_split3_1024x192': extra_config(device, 'BASELINE configs[1] with six bf16 MFMA terms per product in all three kernels (NNR_FP32_PRODUCTS=split3)',
                                                     R_PER_GPU, N_SAMPLES, False, products='split3') if fp32_products == 'split2' else None,
                # what the reference runs for its first 10 000 epochs (configs/default.yaml:99-100,118): the headline step with the per-image
                # point-cloud + surface re-projection losses ON
                'fp32_1024x192_aux': extra_config(device, 'BASELINE configs[1] in the first training phase: per-image losses (pc + rgb_s) on',
                                                  R_PER_GPU, N_SAMPLES, False, aux=True),
                'cpu_32x64_d128': None if args.no_cpu_baseline else cpu_config0(),
            }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
