"""CPU tests of the host side: the `model` package mirrors the reference's API surface, parameter names and O(R) glue
arithmetic; the render path itself refuses to run without the HIP kernels."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as gu
import nerf_oracle as orc
import oracle_backend


def make_cfg(hidden=128, **rend):
    cfg = {
        'model': {'hidden_dim': hidden, 'pos_enc_levels': 10, 'dir_enc_levels': 4, 'occ_activation': 'softplus'},
        'rendering': {'type': 'nope_nerf', 'n_max_network_queries': 64000, 'white_background': False, 'radius': 4.0,
                      'num_points': 64, 'depth_range': [0.01, 10], 'dist_alpha': False, 'use_ray_dir': True,
                      'normalise_ray': True, 'normal_loss': False, 'sample_option': 'uniform', 'outside_steps': 0},
        'depth': {'type': 'None'},
        'distortion': {'fix_scaleN': True},
    }
    cfg['rendering'].update(rend)
    return cfg


def test_public_surface_and_state_dict_names():
    import model as mdl
    for name in ("CheckpointIO", "nope_nerf", "Trainer", "Renderer", "get_model", "OfficialStaticNerf", "LearnPose",
                 "LearnFocal", "Trainer_pose", "Learn_Distortion"):
        assert hasattr(mdl, name), name                       # reference model/__init__.py:1-10
    for hidden in (128, 256):
        cfg = make_cfg(hidden)
        net = mdl.OfficialStaticNerf(cfg)
        model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device='cpu'), cfg, device='cpu')
        ref = np.load(gu.GOLDEN + f"/weights_d{hidden}.npz")
        sd = net.state_dict()
        assert set(sd) == set(ref.files)                      # reference state_dict keys
        for k in ref.files:
            assert tuple(sd[k].shape) == ref[k].shape, k
        assert all(k.startswith("renderer.model.") for k in model.state_dict())
        assert float(net.fc_density.bias) == pytest.approx(0.1) and float(net.fc_rgb.bias[0]) == pytest.approx(0.02)
        assert sum(isinstance(m, torch.nn.Linear) for m in model.modules()) == 12   # train.py:342-344 walks these
        net.load_state_dict({k: torch.from_numpy(ref[k]) for k in ref.files})      # reference checkpoints load


def test_no_cpu_fallback():
    import model as mdl
    cfg = make_cfg()
    net = mdl.OfficialStaticNerf(cfg)
    model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device='cpu'), cfg, device='cpu')
    p = torch.zeros(1, 8, 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(p, torch.arange(8), torch.eye(4)[None], torch.eye(4)[None], torch.eye(4)[None], 'nope_nerf',
              depth_img=torch.ones(1, 1, 4, 4), img_size=(4, 4))
    with pytest.raises(NotImplementedError):
        model.renderer(p, None, torch.eye(4)[None], torch.eye(4)[None], torch.eye(4)[None], 'phong_renderer')


def test_pixel_and_pose_helpers_match_oracle():
    from model.common import arange_pixels, make_c2w, transform_to_world, origin_to_world, get_ndc_rays_fxfy
    _, scaled = arange_pixels((7, 11))
    assert torch.equal(scaled, orc.pixel_grid(7, 11))
    g = torch.Generator().manual_seed(0)
    r, t = torch.randn(3, generator=g) * 0.1, torch.randn(3, generator=g)
    assert torch.allclose(make_c2w(r, t), orc.pose_c2w(r, t), atol=0, rtol=0)
    assert torch.allclose(make_c2w(torch.zeros(3), t)[:3, :3], torch.eye(3))
    K = torch.diag(torch.tensor([1.4, -2.5, -1.0, 1.0]))[None]
    W = torch.inverse(orc.pose_c2w(r, t))[None]
    S = torch.eye(4)[None]
    pix = torch.rand(1, 9, 2, generator=g) * 2 - 1
    d = torch.rand(1, 9, 1, generator=g) + 0.5
    assert torch.allclose(transform_to_world(pix, d, K, W, S), orc.unproject(pix, d, K, W, S), atol=1e-6)
    assert torch.allclose(origin_to_world(9, K, W, S), orc.camera_origin(9, K, W, S), atol=1e-7)
    o, dd = torch.randn(5, 3, generator=g) + torch.tensor([0, 0, 3.0]), torch.randn(5, 3, generator=g)
    a = get_ndc_rays_fxfy(torch.tensor([1.4, -2.5]), 1.0, o, dd)
    b = orc.ndc_rays(torch.tensor([1.4, -2.5]), 1.0, o, dd)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


@pytest.mark.parametrize("dst,src", [((60, 80), (30, 40)), ((540, 960), (384, 672)), ((75, 100), (756, 1008)),
                                       ((60, 80), (60, 80)), ((33, 17), (7, 50))])
def test_nearest_gather_equals_interpolate(dst, src):
    """Gather-at-source == F.interpolate(nearest) + gather (reference model/network.py:22-24)."""
    from model.network import nearest_source_index
    h, w = dst
    img = torch.rand(1, 1, *src)
    idx = torch.randperm(h * w)[:500]
    ref = F.interpolate(img, dst, mode='nearest').view(-1)[idx]
    ys = nearest_source_index(torch.div(idx, w, rounding_mode='floor'), h, src[0])
    xs = nearest_source_index(idx % w, w, src[1])
    assert torch.equal(img[0, 0][ys, xs], ref)


def test_distortion_semantics():
    from model.distortions import Learn_Distortion
    d = Learn_Distortion(3, True, True, {'distortion': {'fix_scaleN': True}})
    with torch.no_grad():
        d.global_scales[0] = 0.001
        d.global_scales[2] = 5.0
    s0, _ = d(0)
    s2, _ = d(2)
    assert float(s0) == pytest.approx(0.01)                                  # clamped ...
    s0.sum().backward()
    assert float(d.global_scales.grad[0]) == 0.0                             # ... to a constant: no gradient reaches the parameter
    assert float(s2) == 1.0 and not s2.requires_grad                         # last camera pinned
    d.global_scales.grad = None
    s1, _ = d(1)
    s1.sum().backward()
    assert float(d.global_scales.grad[1]) == 1.0


def test_render_output_is_lazy_but_complete():
    from model.rendering import RenderOutput
    out = RenderOutput({'rgb': torch.zeros(1, 4, 3), 'dist_dense': torch.tensor([1., 2., 3., 4.]),
                        'd_gt_dense': torch.tensor([2., 0., 4., float('inf')]),
                        'mask': torch.tensor([True, False, True, False]), 'ndc': True})
    assert 'depth_pred' in out and 'depth_gt' in out and 'nope' not in out
    assert torch.equal(out['depth_pred'], torch.tensor([1., 3.]))
    assert torch.equal(out['depth_gt'], 1 - 1 / torch.tensor([2., 4.]))
    assert out.get('depth_pred') is out['depth_pred'] and out.get('nope') is None


@pytest.mark.parametrize("name", ["tanks_d128", "llff_ndc_d128", "uniform_distalpha_masked_d128", "white_nonorm_d128",
                                  "zero_pose_d128", "noraydir_relu_d128"])
def test_renderer_host_glue_against_reference_golden(name, monkeypatch):
    """model.Renderer / nope_nerf / LearnPose / Learn_Distortion with the kernels swapped for the CPU oracle must
    reproduce the reference's outputs and gradients: pins the O(R) host arithmetic (ray generation through the matrix
    inverses, masks, NDC warp, depth_gt) and the autograd edges to pose and distortion."""
    import model as mdl
    import model.rendering as rendering
    monkeypatch.setattr(rendering.nnr, "render_rays", oracle_backend.render_rays)
    case = gu.load_case(name)
    t = gu.tensors(case)
    rc = gu.render_cfg(case)
    cfg = make_cfg(int(case["cfg.hidden"]), **{k: rc[k] for k in ('num_points', 'dist_alpha', 'sample_option', 'depth_range',
                                                                  'normalise_ray', 'white_background', 'use_ray_dir')})
    cfg['model']['occ_activation'] = rc['occ_activation']
    net = mdl.OfficialStaticNerf(cfg)
    net.load_state_dict(case["weights"])
    model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device='cpu'), cfg, device='cpu')
    pose = mdl.LearnPose(gu.N_CAMS, True, True, cfg)
    dist = mdl.Learn_Distortion(gu.N_CAMS, True, True, cfg)
    with torch.no_grad():
        pose.r.copy_(t["pose_r"]); pose.t.copy_(t["pose_t"])
        dist.global_scales.copy_(t["scales"]); dist.global_shifts.copy_(t["shifts"])
    h, w, cam = int(case["cfg.h"]), int(case["cfg.w"]), int(case["cfg.cam"])
    world_mat = torch.inverse(pose(cam)).unsqueeze(0)
    sc, sh = dist(cam)
    depth_in = t["depth_img"] * sc + sh
    from model.common import arange_pixels
    p = arange_pixels((h, w))[1][:, t["ray_idx"]]
    torch.manual_seed(43)   # the renderer's torch.rand(1,R,N) then equals the fixture's jitter
    out = model(p, t["ray_idx"], t["K"], world_mat, torch.eye(4)[None], 'nope_nerf', it=0, depth_img=depth_in,
                add_noise=t["jitter"] is not None, img_size=(h, w))
    for k in ("rgb", "depth_pred", "depth_gt", "alpha", "z_vals"):
        np.testing.assert_allclose(out[k].detach().numpy(), case["out." + k], rtol=0, atol=2e-6, err_msg=k)
    rgb_gt = t["img"].view(1, 3, h * w).permute(0, 2, 1)[:, t["ray_idx"]]
    from model.losses import Loss
    crit = Loss({'depth_loss_type': 'l1'})
    loss = crit.get_rgb_full_loss(out['rgb'], rgb_gt, 'l1') + 0.04 * crit.get_depth_loss(out['depth_pred'], out['depth_gt'])
    assert float(loss) == pytest.approx(float(case["out.loss"]), abs=1e-6)
    loss.backward()
    got = {"w." + k: v.grad for k, v in net.named_parameters()}
    got.update(pose_r=pose.r.grad, pose_t=pose.t.grad, scales=dist.global_scales.grad, shifts=dist.global_shifts.grad)
    for k, (kind, ref, norm) in gu.golden_grads(case).items():
        gu.compare_grad(k, got[k], kind, ref, norm, 3e-5)


def test_shard_bounds_cover():
    from nnr.parallel import shard_bounds
    for n in (1, 7, 1024, 1025):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_resize_nearest_matches_interpolate():
    from model.imaging import resize_nearest
    a = np.random.default_rng(0).standard_normal((21, 34)).astype(np.float32)
    ref = F.interpolate(torch.from_numpy(a)[None, None], (40, 57), mode='nearest')[0, 0].numpy()
    np.testing.assert_array_equal(resize_nearest(a, (40, 57)), ref)


def test_bench_traffic_lookup_finds_the_committed_pmc_summary():
    """bench.py reports roofline.traffic from profiles/r01/hbm_traffic.json; the kernel-name keys must keep matching."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for k in ("mlp_fwd", "mlp_dgrad", "mlp_wgrad"):
        t, src = bench._hbm_traffic(k, False, (1024, 192))
        assert t is not None and 1e9 < t < 1e10, (k, t)
        assert src.startswith("profiles/") and "offline" in src       # the line says the figure is not measured in the run
    assert bench._hbm_traffic("mlp_fwd", False, (7, 7)) == (None, None)   # no summary at that shape: null, never a stale number


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it re-executes under torch.distributed.run (one rank per GPU).  Here, without
    a GPU: --dry-run joins a gloo group instead of touching the device, all-reduces, and rank 0 prints the one JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["dry_run"] is True
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run"], capture_output=True, text=True, timeout=600, env=env)
    assert r1.returncode == 0 and json.loads(r1.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_scene_training_loop_on_the_cpu_stand_in(monkeypatch, tmp_path):
    """tools/train_scene.py end to end (scene writer -> dataloading -> Trainer -> pose metrics) with the oracle-backed operator."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import scene_writer
    import train_scene
    from model import rendering
    monkeypatch.setattr(rendering.nnr, "render_rays", oracle_backend.render_rays)
    scene_writer.write_scene(str(tmp_path), scene="toy", frames=5, size=(24, 32), seed=4)
    res = train_scene.run(str(tmp_path), "toy", epochs=2, log_every=1, device="cpu", n_rays=16, n_samples=8, hidden=128,
                          sample_rate=10 ** 6)
    assert res["steps"] == 10 and res["views"] == 5 and len(res["curve"]) == 3 and res["novel_views"] is None
    # with held-out views: test-time pose optimisation + full-frame scoring of frames 1 and 4 (sample_rate 3)
    res = train_scene.run(str(tmp_path), "toy", epochs=1, log_every=1, device="cpu", n_rays=16, n_samples=8, hidden=128,
                          sample_rate=3, resident=False, eval_epochs=5)
    nv = res["novel_views"]
    assert res["views"] == 3 and nv["views"] == 2 and nv["psnr"] > 0 and 0 < nv["ssim"] <= 1 and nv["pose_opt_mse_last"] > 0
    assert res["curve"][0]["psnr"] is None and res["curve"][-1]["psnr"] > 0 and res["curve"][-1]["ate"] > 0


def test_pixel_pick_capacity_rule_matches_the_library():
    """nnr/sampling.py decides in Python whether the fast pixel pick applies; the C side must agree for every r."""
    import ctypes as C
    from nnr import lib as L
    from nnr import sampling
    lib = L.load()
    for r in list(range(1, 3000, 7)) + [1401, 1402, 8192, 9943, 9944, 12288, 32768, 51463, 51464, 10 ** 6]:
        cap = sampling._capacity(r)
        assert lib.nnr_randperm_scratch_bytes(r) == (8 + 20 * cap if cap else 0), r
    assert [sampling._capacity(r) for r in (1024, 1401, 1402, 8192, 9943, 9944, 32768, 51463, 51464)] == \
        [4096, 4096, 16384, 16384, 16384, 65536, 65536, 65536, 0]
    assert sampling.supported(540 * 960, 8192) and sampling.supported(540 * 960, 32768) and not sampling.supported(540 * 960, 70000)


def test_ssim_metric_matches_the_reference_function():
    """model.imaging.ssim_gaussian == third_party/pytorch_ssim.ssim of the reference checkout (oracle/gen_golden_ssim.py)."""
    import os
    from model import imaging
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssim.npz"))
    for i in range(3):
        a, b = torch.from_numpy(gold[f"a{i}"]), torch.from_numpy(gold[f"b{i}"])
        assert abs(float(imaging.ssim_gaussian(a, b)) - float(gold[f"v{i}"])) <= 2e-6
        assert abs(float(imaging.ssim_gaussian(a, a)) - 1.0) <= 1e-6


def test_encode_position_matches_the_oracle_encoding():
    from model.official_nerf import encode_position
    x = torch.randn(6, 4, 3, generator=torch.Generator().manual_seed(0))
    for levels in (10, 4):
        assert torch.equal(encode_position(x, levels, True), orc.posenc(x, levels))
    assert encode_position(x, 3, False).shape == (6, 4, 18)


def test_committed_bench_line_follows_the_contract():
    """The bench.py line committed with the round's profiles carries every field the driver / judge read (task contract):
    metric block, `roofline` of the dominant kernel with PMC traffic, `cpu_baseline` of the oracle port."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    line = json.loads(open(os.path.join(root, "profiles", "r02", "z_round_end_bench.json.txt")).read())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"] == "training rays/sec" and line["unit"] == "rays/s" and line["n_gpus"] == 1 and line["higher_is_better"] is True
    assert line["scaling"] == "weak" and line["vs_baseline"] is None and line["dtype"] == "f32" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 1024 / (line["ms_per_step"] * 1e-3)) <= 1e-3 * line["value"]
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3
    assert isinstance(r["traffic"], int) and r["traffic"] > 1e9            # HBM bytes per launch from the PMC passes
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "rays/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c


def test_ctypes_signatures_have_the_arity_of_the_header():
    """Every prototype of include/nnr.h against the argtypes the binding declares: a count mismatch would corrupt the call."""
    import os
    import re
    from nnr import lib as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "nnr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = dict(re.findall(r"\b(?:int|size_t|int64_t|const char\*)\s+(nnr_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S))
    assert set(protos) == set(L.EXPORTS), set(protos) ^ set(L.EXPORTS)
    lib = L.load()
    checked = 0
    for name, params in protos.items():
        n = 0 if params.strip() in ("", "void") else params.count(",") + 1
        argtypes = getattr(lib, name).argtypes
        if argtypes is not None:
            assert len(argtypes) == n, (name, len(argtypes), n)
            checked += 1
    assert checked >= 30


def test_checkpoint_roundtrip_with_numpy_scalar(tmp_path):
    """train.py stores `loss_val_best = np.array(psnr_window).mean()` (a numpy.float64) in model.pt once the PSNR window is full;
    torch >= 2.6 refuses to unpickle that with its default weights_only=True.  Trusted local files: they must load."""
    import model as mdl
    cfg = make_cfg()
    net = mdl.OfficialStaticNerf(cfg)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    io = mdl.CheckpointIO(str(tmp_path), model=net, optimizer=opt)
    best = np.array([21.5, 22.5]).mean()
    assert isinstance(best, np.floating)
    io.save("model.pt", epoch_it=1001, it=14014, loss_val_best=best)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    with torch.no_grad():
        for p in net.parameters():
            p.add_(1.0)
    scalars = io.load("model.pt")
    assert scalars["epoch_it"] == 1001 and scalars["it"] == 14014 and float(scalars["loss_val_best"]) == 22.0
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k]), k
    with pytest.raises(FileExistsError):       # sic: the exception type train.py:64-67 catches for "no checkpoint yet"
        io.load("missing.pt")


def test_build_rejects_hot_kernels_that_use_scratch_memory():
    """csrc/build.py parses hipcc's resource report: an MLP kernel whose register arrays fell back to scratch memory (a loop that did
    not unroll) fails the build instead of shipping 8x slower with every parity test green."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("nnr_build", os.path.join(root, "nope-nerf_amd", "csrc", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    rep = ("x.hip:1:1: remark: Function Name: _ZN3nnr16mlp_dgrad_kernelILi256EEEvNS_12MlpDgradArgsE [-Rpass-analysis=kernel-resource-usage]\n"
           "x.hip:1:1: remark:     ScratchSize [bytes/lane]: %d [-Rpass-analysis=kernel-resource-usage]\n")
    b.check_resources(rep % 0, "ok")
    with pytest.raises(RuntimeError, match="scratch"):
        b.check_resources(rep % 832, "bad")
    bf16 = rep.replace("16mlp_dgrad_kernel", "21mlp_dgrad_bf16_kernel")
    b.check_resources(bf16 % 32, "bf16: a few spilled addresses outside the passes are tolerated")
    with pytest.raises(RuntimeError, match="limit 64"):       # ... but not the 96+ bytes that put reloads (= store-queue drains) in a pass
        b.check_resources(bf16 % 96, "bf16 bad")
    assert "-pragma-unroll-threshold=1048576" in b.FLAGS


def test_committed_round6_bench_line_follows_the_contract():
    """The round-6 line (profiles/r06/z_bench_untraced.json.txt: `python bench.py` on one MI355X, the final binaries): the contract's fields, the
    roofline of the dominant kernel against the nearer of its two roofs with PMC traffic, the reference itself as the CPU baseline."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "profiles", "r06", "z_bench_untraced.json.txt")).read()
    line = json.loads([l for l in text.splitlines() if l.startswith("{")][-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"] == "training rays/sec" and line["unit"] == "rays/s" and line["n_gpus"] == 1 and line["higher_is_better"] is True
    assert line["scaling"] == "weak" and line["vs_baseline"] is None and line["dtype"] == "f32" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"] and "configs[1]" in line["config"]["workload"]
    assert abs(line["value"] - 1024 / (line["ms_per_step"] * 1e-3)) <= 1e-3 * line["value"]
    assert line["ms_per_step"] <= 2.75            # VERDICT r05 item 1's target for this configuration
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == {"hbm": "GB/s", "mfma": "TFLOP/s"}[r["bound"]]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-3 and 0.0 < r["frac"] < 1.0
    assert isinstance(r["traffic"], int) and r["traffic"] > 1e9            # HBM bytes per launch from the PMC passes of the same binaries
    assert "r06" in r["traffic_source"]
    other = r["mfma"] if r["bound"] == "hbm" else r["hbm"]                 # the other roof of the same kernel stands beside it
    assert abs(other["frac"] - other["achieved"] / other["peak"]) <= 1e-3
    assert r["clock"]["sclk_mhz_median"] > 1000 and "peak_at_measured_clock" in (r["mfma"] if r["bound"] == "hbm" else r)
    for k in ("mlp_fwd", "mlp_dgrad", "mlp_wgrad"):
        kk = r["kernels"][k]
        assert kk["terms_per_product"] == 3 and 0 < kk["frac_of_bf16_mfma_peak"] < 1 and 0 < kk["frac_of_hbm_peak"] < 1 and kk["algorithmic_tflops"] > 157.3
    c = line["cpu_baseline"]
    assert c["kind"] == "reference" and c["unit"] == "rays/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert line["configs"]["bf16_4096x128"]["tolerance_vs_fp32"]["grad_rel_l2"] == 0.25
