"""Convergence through the REFERENCE code path (BASELINE configs[4]: LLFF-style NDC rays, train to convergence, PSNR + ATE against
the reference): the reference's own Trainer / Renderer / OfficialStaticNerf / LearnPose / Learn_Distortion / Loss and pose metrics
(utils_poses/comp_ate.py:33-73, align_traj.py:26-69) run the training loop of train.py:157-230 on CPU over a synthetic
forward-facing scene (tools/scene_writer.py: 8 views of 48 x 64, LLFF layout) for 100 epochs = 800 Adam steps -- LLFF settings
(configs/LLFF/fern.yaml: sample_option ndc, dist_alpha, depth_range [0, 1]), D = 128, 128 rays x 32 samples, first-phase losses on,
poses from (almost) identity.  Every step's frame, neighbour and pixel permutation are recorded; tests/test_conv_reference.py replays them
through this repository's loop (HIP kernels on the GPU, the oracle stand-in on the CPU) and must track the curve.

TEST INFRASTRUCTURE; authoring container only:   python oracle/gen_golden_conv.py      -> tests/golden/conv_llff.npz"""
import copy
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import gen_golden as gg  # noqa: E402

FRAMES, SIZE, SEED_SCENE = 8, (48, 64), 5
R, N, D, EPOCHS = 128, 32, 128, 100
LOGGED = ("loss", "loss_rgb", "loss_depth", "loss_pc", "loss_rgb_s", "l2_mean")

LOAD_SCENE = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1] + "/nope-nerf_amd"); sys.path.insert(0, sys.argv[1] + "/tools")
import scene_writer, train_scene, dataloading as dl
scene_writer.write_scene(sys.argv[2], scene="toy", frames=%d, size=%r, seed=%d)
cfg = train_scene.scene_cfg(sys.argv[2], "toy", style="llff", n_rays=%d, n_samples=%d, hidden=%d, resident=False, sample_rate=10 ** 6)
_, fields = dl.get_dataloader(cfg, mode="train", shuffle=True)
f = fields["img"]
np.savez(sys.argv[3], imgs=f.imgs, dpt=f.dpt_depth, K=f.K, c2ws=np.asarray(f.c2ws), refs=np.array([f.pick_reference(i) for i in range(f.N_imgs)]))
''' % (FRAMES, SIZE, SEED_SCENE, R, N, D)


def main(threads=8, replay=None, out_name="conv_llff.npz", epochs=EPOCHS, phase=None, seed=None):
    """threads / replay: the chaos envelope (see envelope()) -- the same reference run with another GEMM thread count, fed the frames
    and pixel permutations the golden run drew (`replay` = the golden blob).
    phase = (scheduling_start, annealing_epochs): the TWO-PHASE run -- what train.py's PSNR-plateau scheduler (train.py:309-340) does once
    it fires at epoch `scheduling_start`: model/training.py:187-217 anneals pc / rgb_s / depth weights to their end values over
    `annealing_epochs` epochs and switches the rgb term from L1 to L2 afterwards.
    seed: another DRAW of the run -- the same scene, network and initial poses, but the order of the frames and every step's pixel pick come
    from torch.manual_seed(1000 + seed) (see seeds())."""
    with tempfile.TemporaryDirectory() as tmp:
        # the scene goes through THIS repository's loader (pinned bit-exact against the reference's DataField: tests/test_dataloading.py)
        # in a separate process: `model` / `dataloading` of the reference are imported below under the same names
        npz = os.path.join(tmp, "scene.npz")
        subprocess.run([sys.executable, "-c", LOAD_SCENE, ROOT, tmp, npz], check=True)
        sc = np.load(npz)
        imgs, dpt, K, gt, refs = (torch.from_numpy(sc[k]) for k in ("imgs", "dpt", "K", "c2ws", "refs"))
    ref = gg.import_reference()
    sys.path.insert(0, gg.REF)
    from utils_poses.align_traj import align_ate_c2b_use_a2b
    from utils_poses.comp_ate import compute_ATE, compute_rpe
    from model.common import mse2psnr
    torch.set_num_threads(threads)
    cfg = copy.deepcopy(gg.base_cfg(D))
    cfg["training"].update(n_training_points=R, pc_weight=[1.0, 0.0], rgb_s_weight=[1.0, 0.0], vis_reprojection_every=10 ** 9)
    sched_start = 10 ** 6
    if phase is not None:
        sched_start = int(phase[0])
        cfg["training"]["annealing_epochs"] = int(phase[1])
    cfg["rendering"].update(num_points=N, sample_option="ndc", dist_alpha=True, depth_range=[0.0, 1.0])
    dev = torch.device("cpu")
    np.random.seed(42)
    torch.manual_seed(42)                                            # train.py:22-23
    net = ref.OfficialStaticNerf(cfg)
    base = np.load(os.path.join(gg.OUT, "weights_d128.npz"))
    assert all(np.array_equal(base[k], v.numpy()) for k, v in net.state_dict().items())      # the seed-42 D=128 network
    model = ref.get_model(ref.Renderer(net, cfg["rendering"], device=dev), cfg, device=dev)
    pose, dist = ref.LearnPose(FRAMES, True, True, cfg), ref.Learn_Distortion(FRAMES, True, True, cfg)
    # Not exactly the identity: with all cameras at the identity the surface re-projection maps the sampling grid onto itself, every
    # bilinear lookup sits exactly on a pixel centre and every border point exactly on |xy| = 1 -- floor() and the validity test
    # then hinge on the last bit, and two correct implementations disagree by a whole point (1/192 of the term) at step 0.
    g = torch.Generator().manual_seed(17)
    with torch.no_grad():
        pose.r.copy_(0.01 * torch.randn(FRAMES, 3, generator=g))
        pose.t.copy_(0.01 * torch.randn(FRAMES, 3, generator=g))
    init_r, init_t = pose.r.detach().clone().numpy(), pose.t.detach().clone().numpy()
    adam = lambda m, lr: torch.optim.Adam(m.parameters(), lr=lr)
    tr = ref.Trainer(model, adam(model, 1e-3), cfg["training"], device=dev, optimizer_pose=adam(pose, 5e-4), pose_param_net=pose,
                     optimizer_distortion=adam(dist, 5e-4), distortion_net=dist)
    real_randperm = torch.randperm
    drawn = {}

    def randperm(n, *a, **k):
        if replay is not None:      # the recorded pick of this step (the tail of the permutation is never read)
            drawn["perm"] = torch.cat([torch.from_numpy(replay["ray_idx"][len(order)].astype(np.int64)), torch.zeros(n - R, dtype=torch.int64)])
        else:
            drawn["perm"] = real_randperm(n, *a, **k)
        return drawn["perm"]

    def pose_errors():
        with torch.no_grad():
            learned = torch.stack([pose(i) for i in range(FRAMES)])
        aligned = align_ate_c2b_use_a2b(learned, gt).cpu().numpy()
        rpe_t, rpe_r = compute_rpe(gt.numpy(), aligned)
        return float(compute_ATE(gt.numpy(), aligned)), float(rpe_t * 100), float(np.degrees(rpe_r))

    if seed is not None:
        torch.manual_seed(1000 + int(seed))
    Kt = K.unsqueeze(0) if K.dim() == 2 else K
    eye = torch.eye(4).unsqueeze(0)
    order, picks, losses, curve = [], [], [], [(-1, float("nan")) + pose_errors()]
    torch.randperm = randperm
    try:
        it = 0          # train.py starts at -1 + 1 = 0, where the first step also dumps the re-projection PNGs (it % vis_reprojection_every): start at 1
        for epoch in range(epochs):
            l2 = []
            cams = real_randperm(FRAMES).tolist() if replay is None else [int(c) for c, _ in replay["order"][epoch * FRAMES:(epoch + 1) * FRAMES]]
            for cam in cams:                                        # the shuffled DataLoader of train.py:35
                it += 1
                nb = int(refs[cam])
                data = {"img": imgs[cam:cam + 1], "img.idx": cam, "img.dpt": dpt[cam:cam + 1], "img.camera_mat": Kt,
                        "img.scale_mat": eye, "img.ref_imgs": imgs[nb:nb + 1], "img.ref_dpts": dpt[nb:nb + 1], "img.ref_idxs": nb}
                ld = tr.train_step(data, it, epoch, sched_start, None)
                order.append((cam, nb))
                picks.append(drawn["perm"][:R].numpy().astype(np.int16))
                losses.append([float(ld[k]) for k in LOGGED])
                l2.append(float(ld["l2_mean"]))
            if epoch % 10 == 0 or epoch == epochs - 1:
                curve.append((epoch, float(mse2psnr(np.mean(l2)))) + pose_errors())
                print("epoch %3d  PSNR %.2f dB  ATE %.4f  RPE_t %.3f  RPE_r %.3f deg" % curve[-1])
    finally:
        torch.randperm = real_randperm
    blob = {"order": np.array(order, dtype=np.int16), "ray_idx": np.stack(picks), "losses": np.array(losses, dtype=np.float64),
            "logged": np.array(LOGGED), "curve": np.array(curve, dtype=np.float64),
            "init.pose_r": init_r, "init.pose_t": init_t, "final.pose_r": pose.r.detach().numpy(), "final.pose_t": pose.t.detach().numpy(),
            "final.scales": dist.global_scales.detach().numpy(), "final.shifts": dist.global_shifts.detach().numpy(),
            "cfg": np.array([FRAMES, SIZE[0], SIZE[1], SEED_SCENE, R, N, D, epochs]),
            "phase": np.array([sched_start, cfg["training"]["annealing_epochs"]])}
    if out_name is None:
        return blob
    out = os.path.join(gg.OUT, out_name)
    np.savez_compressed(out, **blob)
    print("wrote", out, os.path.getsize(out), "bytes;", len(order), "steps")
    return blob


def envelope(thread_counts=(1, 2, 3, 4, 5, 6, 7), name="conv_llff", **kw):
    """How far apart do two runs of the REFERENCE ITSELF end?  Training is a chaotic map: another summation order inside the CPU GEMMs
    (another thread count) changes last bits, Adam amplifies them, and after 800 steps the two runs are two samples of the same
    distribution.  Each variant replays the golden run's frames and pixel picks; recorded per variant: final PSNR / ATE / RPE and the
    deviation of the loss curve from the golden run, in the very statistics tests/test_conv_reference.py asserts.  The test's
    statistical tolerances are tied to this spread (tests/golden/conv_llff_envelope.npz)."""
    gold = dict(np.load(os.path.join(gg.OUT, name + ".npz")))
    k = list(gold["logged"]).index("loss")
    smooth = lambda x: np.convolve(x, np.ones(40) / 40, mode="valid")
    rows = []
    for t in thread_counts:
        b = main(threads=t, replay=gold, out_name=None, **kw)
        dev = np.abs(b["losses"] - gold["losses"]) / np.maximum(1.0, np.abs(gold["losses"]))
        curve = float(np.abs(smooth(b["losses"][:, k]) - smooth(gold["losses"][:, k])).max() / smooth(gold["losses"][:, k]).max())
        _, psnr, ate, rpe_t, rpe_r = b["curve"][-1]
        rows.append((t, psnr, ate, rpe_t, rpe_r, float(dev[:20].max()), float(dev[:50].max()), curve))
        print("threads %d: PSNR %.3f ATE %.4f RPE_r %.3f; first-20 dev %.2e, first-50 %.2e, smoothed curve %.2e" % (t, psnr, ate, rpe_r, rows[-1][5], rows[-1][6], curve))
    out = os.path.join(gg.OUT, name + "_envelope.npz")
    np.savez_compressed(out, runs=np.array(rows, dtype=np.float64), golden_final=gold["curve"][-1],
                        columns=np.array(["threads", "psnr", "ate", "rpe_t", "rpe_r", "dev_first20", "dev_first50", "curve_dev"]))
    print("wrote", out)


def seeds(n=8, threads=8):
    """the envelope above varies only the summation order; a statistic needs independent SAMPLES.  n more runs of the
    reference, each with its own frame order and pixel picks (seed 1 .. n), recorded so that tests / tools can replay every one of them on
    the HIP kernels: a PAIRED comparison (same draws, reference arithmetic against ours), in which the batch noise cancels.
    tests/golden/conv_llff_seeds.npz: per seed the draws (order, ray_idx), the first 20 steps' logged losses, the final PSNR / ATE / RPE."""
    out = dict(order=[], ray_idx=[], losses20=[], final=[], psnr_curve=[])
    for s in range(1, n + 1):
        b = main(threads=threads, out_name=None, seed=s)
        out["order"].append(b["order"]); out["ray_idx"].append(b["ray_idx"]); out["losses20"].append(b["losses"][:20])
        out["final"].append(b["curve"][-1][1:]); out["psnr_curve"].append(b["curve"][:, 1])
        print("seed %d: final PSNR %.3f ATE %.4f RPE_r %.3f" % ((s,) + tuple(b["curve"][-1][[1, 2, 4]])), flush=True)
    gold = np.load(os.path.join(gg.OUT, "conv_llff.npz"))
    path = os.path.join(gg.OUT, "conv_llff_seeds.npz")
    np.savez_compressed(path, seeds=np.arange(1, n + 1), order=np.stack(out["order"]), ray_idx=np.stack(out["ray_idx"]),
                        losses20=np.stack(out["losses20"]), final=np.array(out["final"], dtype=np.float64), psnr_curve=np.array(out["psnr_curve"]),
                        columns=np.array(["psnr", "ate", "rpe_t", "rpe_r"]), logged=np.array(LOGGED), cfg=gold["cfg"],
                        **{k: gold[k] for k in ("init.pose_r", "init.pose_t")})
    print("wrote", path, os.path.getsize(path), "bytes")


TWO_PHASE = dict(epochs=110, phase=(40, 20))     # switch at epoch 40, annealed by 60, L2 from 60 on, 50 more epochs = 400 steps beyond

if __name__ == "__main__":
    if "--two-phase" in sys.argv:       # tests/golden/conv_llff_2phase.npz + its reference-vs-reference envelope
        main(out_name="conv_llff_2phase.npz", **TWO_PHASE)
        envelope((1, 3, 5, 7), name="conv_llff_2phase", **TWO_PHASE)
    elif "--envelope" in sys.argv:
        envelope()
    elif "--seeds" in sys.argv:         # tests/golden/conv_llff_seeds.npz: eight independent draws of the reference run
        seeds(int(sys.argv[sys.argv.index("--seeds") + 1]) if len(sys.argv) > sys.argv.index("--seeds") + 1 else 8)
    else:
        main()
