"""CPU, world_size 2 over gloo: the ray-sharded data-parallel step (model.Trainer + nnr.parallel) reproduces the
single-process step -- same loss, same gradients for MLP, pose and distortion -- with ONE flat all-reduce.
The HIP render operator is swapped for the CPU oracle backend (tests/oracle_backend.py) so that the host logic under test
(shard bounds, jitter window, global loss normalisation, flat-bucket all-reduce) runs without a GPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import golden_util as gu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(case, n_rays, depth_loss_type='l1'):
    import model as mdl
    import model.rendering as rendering
    import oracle_backend
    from test_host_logic import make_cfg
    rendering.nnr.render_rays = oracle_backend.render_rays
    t = gu.tensors(case)
    rc = gu.render_cfg(case)
    cfg = make_cfg(int(case["cfg.hidden"]), **{k: rc[k] for k in ('num_points', 'dist_alpha', 'sample_option', 'depth_range',
                                                                  'normalise_ray', 'white_background', 'use_ray_dir')})
    cfg['model']['occ_activation'] = rc['occ_activation']
    tcfg = {'type': 'nope_nerf', 'n_training_points': n_rays, 'vis_geo': False, 'detach_gt_depth': False, 'pc_ratio': 4,
            'match_method': 'dense', 'shift_first': False, 'detach_ref_img': True, 'scale_pcs': True, 'detach_rgbs_scale': False,
            'vis_reprojection_every': 5000, 'nearest_limit': 0.01, 'annealing_epochs': 2000, 'rgb_weight': [1.0, 1.0],
            'depth_weight': [0.04, 0.0], 'pc_weight': [0.0, 0.0], 'rgb_s_weight': [0.0, 0.0],
            'depth_consistency_weight': [0.0, 0.0], 'weight_dist_2nd_loss': [0.5, 0.5], 'weight_dist_1st_loss': [0.1, 0.1],
            'depth_loss_type': depth_loss_type, 'with_ssim': False, 'with_auto_mask': False}
    net = mdl.OfficialStaticNerf(cfg)
    net.load_state_dict(case["weights"])
    model = mdl.get_model(mdl.Renderer(net, cfg['rendering'], device='cpu'), cfg, device='cpu')
    pose = mdl.LearnPose(gu.N_CAMS, True, True, cfg)
    distn = mdl.Learn_Distortion(gu.N_CAMS, True, True, cfg)
    with torch.no_grad():
        pose.r.copy_(t["pose_r"]); pose.t.copy_(t["pose_t"])
        distn.global_scales.copy_(t["scales"]); distn.global_shifts.copy_(t["shifts"])
    sgd = lambda m: torch.optim.SGD(m.parameters(), lr=0.0)
    tr = mdl.Trainer(model, sgd(model), tcfg, device='cpu', optimizer_pose=sgd(pose), pose_param_net=pose,
                     optimizer_distortion=sgd(distn), distortion_net=distn)
    data = {'img': t["img"], 'img.idx': int(case["cfg.cam"]), 'img.dpt': t["depth_img"][:, 0], 'img.camera_mat': t["K"],
            'img.scale_mat': torch.eye(4)[None]}
    return tr, net, pose, distn, data


def _step(case, n_rays, depth_loss_type='l1'):
    tr, net, pose, distn, data = _build(case, n_rays, depth_loss_type)
    torch.manual_seed(123)                      # same permutation and jitter stream on every rank
    ld = tr.train_step(data, it=0, epoch=0, scheduling_start=10000, render_path=None)
    grads = {k: v.grad.clone() for k, v in net.named_parameters()}
    grads.update(r=pose.r.grad.clone(), t=pose.t.grad.clone(), scale=distn.global_scales.grad.clone(),
                 shift=distn.global_shifts.grad.clone())
    return {k: float(ld[k]) for k in ('loss', 'loss_rgb', 'loss_depth', 'l2_mean', 'loss_dist_1st', 'loss_dist_2nd')}, grads


def _worker(rank, world, port, name, n_rays, q, depth_loss_type='l1'):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        losses, grads = _step(gu.load_case(name), n_rays, depth_loss_type)
        if rank == 0:
            q.put((losses, {k: v.numpy() for k, v in grads.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,n_rays,depth_loss_type", [("tanks_d128", 31, "l1"), ("uniform_distalpha_masked_d128", 48, "l1"),
                                                         ("uniform_distalpha_masked_d128", 47, "invariant")])
def test_two_rank_step_equals_single_process(name, n_rays, depth_loss_type):
    """'invariant' normalises the depths by the median / mean deviation over all valid rays of the step: the ranks all-gather
    the per-ray values (nnr.parallel.gather_rays) and differentiate the full loss with respect to their own rays."""
    ref_losses, ref_grads = _step(gu.load_case(name), n_rays, depth_loss_type)          # world size 1, this process
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, n_rays, q, depth_loss_type)) for r in range(2)]
    for p in procs:
        p.start()
    losses, grads = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k, v in ref_losses.items():
        assert abs(losses[k] - v) <= 2e-6 * max(1.0, abs(v)), (k, losses[k], v)
    for k, g in ref_grads.items():
        err = float((torch.from_numpy(grads[k]) - g).abs().max())
        assert err <= 1e-5 * max(1.0, float(g.abs().max())), (k, err)


def _missing_worker(rank, world, port, q):
    from nnr import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a, b, c = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2)), torch.nn.Parameter(torch.ones(4))
        a.grad = torch.full((3,), 2.0 + rank)
        if rank == 1:
            b.grad = torch.full((2,), 5.0)          # only rank 1 has a gradient for b; nobody has one for c
        ld = {'loss': torch.tensor(1.5)}
        parallel.allreduce_gradients([a, b, c], ld)
        q.put((rank, a.grad.tolist(), None if b.grad is None else b.grad.tolist(), c.grad is None, float(ld['loss'])))
    finally:
        dist.destroy_process_group()


def test_allreduce_keeps_a_gradient_none_only_when_no_rank_has_one():
    """Adam skips a parameter without a gradient (no momentum step, no counter increment): under data parallelism a table that no rank has
    a gradient for -- the distortion scales in a gauge-camera step, model/distortions.py:23-24 -- must stay without one, while a table only
    SOME rank has a gradient for gets the sum everywhere (the bucket layout is the same on every rank either way)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_missing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ga, gb, c_none, loss in got:
        assert ga == [5.0, 5.0, 5.0] and gb == [5.0, 5.0] and c_none and loss == 3.0, got


def _inplace_worker(rank, world, port, q):
    from nnr import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # twelve gradients as views of ONE buffer with alignment gaps and a spare tail -- what nnr.ops._RenderRays.backward hands autograd --
        # plus two free-standing tables (one without a gradient on rank 1, one on no rank) and two logged scalars
        sizes = [7, 3, 16, 1, 5, 9, 2, 4, 8, 6, 10, 11]
        offs, o = [], 0
        for n in sizes:
            offs.append(o)
            o += (n + 3) // 4 * 4
        # gaps and tail zero-filled and the buffer REGISTERED, as _RenderRays.backward does under data parallelism: the in-place path is taken for
        # such a buffer only (an unregistered storage that merely holds many gradients goes through the private bucket: second half of this worker)
        from nnr import ops
        flat = torch.zeros(o + 64)
        ops.register_flat_grads(flat, o)
        params = []
        for i, (n, off) in enumerate(zip(sizes, offs)):
            p = torch.nn.Parameter(torch.zeros(n))
            flat[off:off + n] = float(rank + 1) * (i + 1)
            p.grad = flat[off:off + n]
            params.append(p)
        a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))
        if rank == 0:
            a.grad = torch.full((3,), 4.0)
        ld = {'loss': torch.tensor(1.0 + rank), 'loss_rgb': torch.tensor(0.5)}
        ptr = flat.data_ptr()
        parallel.allreduce_gradients(params + [a, b], ld)
        ok = all(p.grad.data_ptr() == ptr + 4 * off for p, off in zip(params, offs))           # reduced where they lie: no copy
        vals = [float(p.grad[0]) for p in params] + [float(p.grad.min()) for p in params]
        # the logged scalars are copies, not views of the reduced buffer: rewriting the buffer (the next step, zero_grad(set_to_none=False)) leaves them alone
        flat.fill_(-7.0)
        # an UNREGISTERED shared storage (NaN in its gaps, live data behind the views): summed through the private bucket, nothing outside the views touched
        other = torch.full((64,), float("nan"))
        qs = []
        for i in range(9):
            pq = torch.nn.Parameter(torch.zeros(3))
            other[4 * i:4 * i + 3] = float(rank + 1)
            pq.grad = other[4 * i:4 * i + 3]
            qs.append(pq)
        other[40:] = 123.0
        parallel.allreduce_gradients(qs, None)
        ok = ok and all(float(pq.grad[0]) == 3.0 for pq in qs) and bool((other[40:] == 123.0).all()) and bool(torch.isnan(other[3]))
        q.put((rank, ok, vals, a.grad.tolist() if a.grad is not None else None, b.grad is None, float(ld['loss']), float(ld['loss_rgb'])))
    finally:
        dist.destroy_process_group()


def test_allreduce_runs_in_place_on_a_shared_gradient_buffer():
    """The GPU path's layout on CPU tensors: gradients that are views of one allocation are summed in place (their storage pointers do not
    change), everything else rides in the buffer's tail; values, None-ness and the logged scalars as in the private-bucket path."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_inplace_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, vals, ga, b_none, loss, lrgb in got:
        assert ok, "the shared gradients were copied"
        assert vals[:12] == [3.0 * (i + 1) for i in range(12)] and vals[12:] == vals[:12]      # (1 + 2) (i + 1), every element
        assert ga == [4.0, 4.0, 4.0] and b_none and loss == 3.0 and lrgb == 1.0


def test_allreduce_handles_missing_grads():
    """A parameter without a gradient on this rank still occupies its slot in the flat bucket."""
    from nnr import parallel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        a, b = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2))
        a.grad = torch.full((3,), 2.0)
        ld = {'loss': torch.tensor(1.5)}
        parallel.allreduce_gradients([a, b], ld)
        assert torch.equal(a.grad, torch.full((3,), 2.0)) and b.grad is None and float(ld['loss']) == 1.5
    finally:
        dist.destroy_process_group()


def _scene_worker(rank, world, port, path, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "tools"))
    import model.rendering as rendering
    import oracle_backend
    import train_scene
    rendering.nnr.render_rays = oracle_backend.render_rays
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        res = train_scene.run(path, "toy", epochs=3, log_every=1, device="cpu", n_rays=24, n_samples=8, hidden=128, sample_rate=10 ** 6,
                              resident=False)
        if rank == 0:
            q.put(res["curve"])
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_two_rank_training_loop_follows_the_single_process_run(tmp_path):
    """The whole loop (tools/train_scene.py: loader order, pixel picks, jitter, per-image losses, three Adam optimisers) under
    two gloo ranks: identical seeds give both ranks the same views and draws, each renders half of the rays, and after three epochs
    PSNR and pose errors equal the single-process run."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import scene_writer
    scene_writer.write_scene(str(tmp_path), scene="toy", frames=4, size=(24, 32), seed=6)
    ctx = mp.get_context("spawn")
    curves = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_scene_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
        for p in procs:
            p.start()
        curves[world] = q.get(timeout=600)
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    a, b = curves[1][-1], curves[2][-1]
    assert len(curves[1]) == len(curves[2]) == 4
    # Adam divides by sqrt(v) + 1e-8: fp32 summation-order differences of the all-reduce show up at 1e-5 in the poses
    assert abs(a["psnr"] - b["psnr"]) <= 2e-3 and abs(a["ate"] - b["ate"]) <= 1e-4 and abs(a["rpe_rot_deg"] - b["rpe_rot_deg"]) <= 5e-3, (a, b)


# ---- first-phase per-image losses (point cloud + surface re-projection) sharded by source points ---------------------------
def _aux_step(name, monkeypatch_like=None):
    """One Trainer.train_step with pc_weight = rgb_s_weight = 1 on the reference golden's inputs and draws (tests/golden/aux_terms.npz,
    minted from the REFERENCE Trainer.train_step)."""
    import numpy as np
    import model.rendering as rendering
    import oracle_backend
    import test_aux_terms as ta
    rendering.nnr.render_rays = oracle_backend.render_rays
    G = ta.GOLD
    inp = ta._inp(name)
    cam, ref = int(G[f"{name}.cam"]), int(G[f"{name}.ref"])
    tr, pose, distn = ta._trainer(inp, torch.device("cpu"))
    ray_idx, jitter = torch.from_numpy(G[f"{name}.ray_idx"]), torch.from_numpy(G[f"{name}.jitter"])
    real_randperm, real_rand = torch.randperm, torch.rand
    torch.randperm = lambda n, device=None, **kw: torch.cat([ray_idx, torch.zeros(n - ta.R, dtype=torch.int64)])
    torch.rand = lambda *s, device=None, **kw: jitter if tuple(s) == (1, ta.R, ta.N) else real_rand(*s, device=device, **kw)
    try:
        data = {"img": inp["img"], "img.idx": cam, "img.dpt": inp["dpt"], "img.camera_mat": inp["K"], "img.scale_mat": torch.eye(4).unsqueeze(0),
                "img.ref_imgs": inp["ref_img"], "img.ref_dpts": inp["ref_dpt"], "img.ref_idxs": ref}
        ld = tr.train_step(data, it=1, epoch=0, scheduling_start=10000, render_path=None)
    finally:
        torch.randperm, torch.rand = real_randperm, real_rand
    losses = {k: float(ld[k]) for k in ("loss", "loss_pc", "loss_rgb_s", "loss_rgb", "loss_depth")}
    zeros = lambda p: torch.zeros_like(p)
    grads = {"pose_r": pose.r.grad if pose.r.grad is not None else zeros(pose.r), "pose_t": pose.t.grad if pose.t.grad is not None else zeros(pose.t),
             "scales": distn.global_scales.grad if distn.global_scales.grad is not None else zeros(distn.global_scales),      # (the gauge camera's step leaves the scales without a gradient, on every rank)
             "shifts": distn.global_shifts.grad if distn.global_shifts.grad is not None else zeros(distn.global_shifts)}
    return losses, {k: v.detach().numpy().copy() for k, v in grads.items()}


def _aux_worker(rank, world, port, name, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        out = _aux_step(name)
        if rank == 0:
            q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("mid", 2), ("last", 3)])
def test_per_image_losses_sharded_by_source_points_equal_the_reference_golden(name, world):
    """Rank k evaluates the point-cloud / re-projection sums over its shard of the sampling grid (the O(S^2) nearest-neighbour
    search shrinks by the world size), normalisers stay global; after the flat all-reduce losses and pose / distortion
    gradients are the REFERENCE's single-process values (golden minted by oracle/gen_golden_aux.py)."""
    import numpy as np
    import test_aux_terms as ta
    G = ta.GOLD
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_aux_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    losses, grads = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k, v in losses.items():
        np.testing.assert_allclose(v, float(G[f"{name}.out.{k}"]), rtol=0, atol=1e-5, err_msg=k)
    for k, g in grads.items():
        ref_g = G[f"{name}.g.{k}"]
        assert float(np.abs(g - ref_g).max()) / max(1.0, float(np.abs(ref_g).max())) <= 1e-4, k


def test_eight_ranks_through_the_bench_launcher_equal_one_process(tmp_path):
    """What the driver's `bench.py --gpus 8` does that no test with two hand-spawned ranks covers: EIGHT ranks started by
    bench.self_launch -- torch.distributed.run, --master-addr 127.0.0.1, one process per rank -- join one group, shard a step's rays eight ways
    (BASELINE configs[3]'s decomposition: 8 shards of one step, scaled down to what eight CPU processes finish in seconds), run TWO steps and
    all-reduce once per step; the summed gradients and the losses equal a single process's on the same 8 x R rays, and the line rank 0
    prints carries collective.rccl_ranks_seen == 8."""
    import json
    import sys
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    import dp_dryrun_worker as worker
    name, n_rays = "tanks_d128", 8 * 12
    out = str(tmp_path / "dp8.npz")
    rc, stdout = bench.self_launch(8, script=os.path.join(ROOT, "tests", "dp_dryrun_worker.py"), argv=[name, str(n_rays), out], capture=True)
    assert rc == 0, stdout[-3000:]
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]                 # rank 0 prints, nobody else does
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["collective"]["rccl_ranks_seen"] == 8 and line["collective"]["bucket_floats"] > 150000
    got = np.load(out)
    ref = worker.run_steps(name, n_rays)                   # world size 1, this process: the same two steps on all 96 rays
    for i, (losses, grads) in enumerate(ref):
        for k, v in losses.items():
            assert abs(float(got["s%d.loss.%s" % (i, k)]) - v) <= 2e-6 * max(1.0, abs(v)), (i, k)
        for k, g in grads.items():
            err = float(np.abs(got["s%d.grad.%s" % (i, k)] - g).max())
            assert err <= 1e-5 * max(1.0, float(np.abs(g).max())), (i, k, err)
    assert any(np.abs(got["s1.grad.%s" % k] - got["s0.grad.%s" % k]).max() > 0 for k in ("r", "t"))      # the second step drew other rays
