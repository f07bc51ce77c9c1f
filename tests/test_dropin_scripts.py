"""CPU, authoring container only (skipped where the reference checkout is absent): the reference's OWN unmodified train.py and
evaluation/eval_poses.py run end to end against this repository's `model` + `dataloading` + `utils_poses` packages on a synthetic
scene written by tools/scene_writer.py -- the drop-in boundary of SURVEY.md 8(b) exercised by its real callers.  The render
operator is the oracle-backed CPU stand-in (no GPU here); tests/test_gpu_scene_training.py runs the same loop on the HIP kernels."""
import json
import os
import subprocess
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
REF = os.environ.get("NNR_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train.py")), reason="reference checkout not present")


def _run(script, cfg_path, env_extra, *args):
    env = dict(os.environ, PYTHONPATH="", **env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "dropin_runner.py"), os.path.join(REF, script), cfg_path, *args]
    return subprocess.run(cmd, cwd=REF, env=env, capture_output=True, text=True, timeout=600)


def test_reference_train_and_eval_poses_run_on_our_packages(tmp_path):
    import scene_writer
    data = str(tmp_path / "data")
    scene_writer.write_scene(data, scene="toy", frames=6, size=(24, 32), seed=2)
    out_dir = str(tmp_path / "out")
    cfg = {
        "model": {"hidden_dim": 128},
        "dataloading": {"path": data, "scene": ["toy"], "n_workers": 0, "resize_factor": None, "sample_rate": 3, "spherify": False},
        "rendering": {"num_points": 8},
        "pose": {"learn_pose": True},
        "training": {"out_dir": out_dir, "n_training_points": 16, "scheduling_start": 1, "scheduling_epoch": 1, "annealing_epochs": 1,
                     "print_every": 4, "checkpoint_every": 4, "visualize_every": 6, "vis_resolution": [6, 8], "pc_ratio": 2,
                     "auto_scheduler": False},
        "extract_images": {"resolution": [24, 32], "N_novel_imgs": 5},
        "eval_pose": {"opt_pose_epoch": 5, "n_points": 16},
    }
    cfg_path = str(tmp_path / "toy.yaml")
    with open(cfg_path, "w") as fh:
        yaml.safe_dump(cfg, fh)
    scalars = str(tmp_path / "scalars.json")
    r = _run("train.py", cfg_path, {"DROPIN_SCALARS": scalars})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    # 2 epochs x 4 training views (frames 1 and 4 are held out) = 8 steps went through Trainer.train_step; the loop logged, evaluated the poses and checkpointed
    assert "ATE:" in r.stdout and "PSNR:" in r.stdout
    for f in ("model.pt", "model_pose.pt", "model_distortion.pt"):
        assert os.path.isfile(os.path.join(out_dir, f)), f
    assert os.path.isdir(os.path.join(out_dir, "rendering", "0006_vis"))
    tags = {t for t, _, _ in json.load(open(scalars))}
    for t in ("train/loss", "train/loss_rgb", "train/loss_depth", "train/loss_pc", "train/loss_rgb_s", "train/l2_mean",
              "eval/ate_trans", "eval/rpe_rot", "train/psnr", "train/lr_pose", "train/lr_distortion"):
        assert t in tags, (t, sorted(tags))
    # the pose evaluation script reads the checkpoint back through our CheckpointIO / LearnPose / dataloading / utils_poses
    r = _run(os.path.join("evaluation", "eval_poses.py"), cfg_path, {})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.count("&") == 2, last                                    # "rpe_t & rpe_r & ate"
    # novel-view evaluation: test-time pose optimisation of the held-out views (Trainer_pose) + full-image rendering (Eval_Images)
    r = _run(os.path.join("evaluation", "eval.py"), cfg_path, {})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Opt: L2 loss" in r.stdout and "Mean MSE" in r.stdout
    shots = os.listdir(os.path.join(out_dir, "extraction", "eval", "pre"))
    assert "video_out" in shots and len(shots) >= 3, shots
    # novel-view rendering along a B-spline through the learned poses (model.common path helpers + Extract_Images)
    r = _run(os.path.join("vis", "render.py"), cfg_path, {})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    frames = os.listdir(os.path.join(out_dir, "extraction", "extracted_images", "bspline"))
    assert "video_out" in frames and len(frames) >= 3, frames
    # the other two path options of vis/render.py: slerp + linear interpolation, and the NeRF-style spiral ("sprial" upstream)
    for option in ("interp", "sprial"):
        cfg["extract_images"]["traj_option"] = option
        other = str(tmp_path / ("toy_%s.yaml" % option))
        with open(other, "w") as fh:
            yaml.safe_dump(cfg, fh)
        r = _run(os.path.join("vis", "render.py"), other, {})
        assert r.returncode == 0, option + r.stdout[-2000:] + r.stderr[-3000:]
        assert "video_out" in os.listdir(os.path.join(out_dir, "extraction", "extracted_images", option))


def _base_cfg(data, out_dir):
    return {
        "model": {"hidden_dim": 128},
        "dataloading": {"path": data, "scene": ["toy"], "n_workers": 0, "resize_factor": None, "sample_rate": 3, "spherify": False},
        "rendering": {"num_points": 8},
        "pose": {"learn_pose": True},
        "training": {"out_dir": out_dir, "n_training_points": 16, "scheduling_start": 1, "scheduling_epoch": 1, "annealing_epochs": 1,
                     "print_every": 4, "checkpoint_every": 4, "visualize_every": 1000, "vis_resolution": [6, 8], "pc_ratio": 2,
                     "auto_scheduler": False},
        "extract_images": {"resolution": [24, 32], "N_novel_imgs": 5},
        "eval_pose": {"opt_pose_epoch": 5, "n_points": 16},
    }


@pytest.mark.parametrize("name,over", [
    # the PSNR-plateau scheduler of the Tanks configs, network re-initialisation at the switch, per-view distortion logging
    ("auto_reset", {"training": {"auto_scheduler": True, "scheduling_mode": "reset", "length_smooth": 1, "patient": 1,
                                 "log_scale_shift_per_view": True}}),
    # learnable focal length from the ground-truth focal, poses initialised from the (COLMAP-format) ground truth, SSIM term on
    ("focal_gtpose_ssim", {"pose": {"learn_focal": True, "init_pose": True, "init_pose_type": "gt", "init_focal_type": "gt"},
                           "training": {"with_ssim": True}}),
    # LLFF-style rendering: NDC sampling, distance-based alpha, down-sized frames
    ("llff", {"rendering": {"sample_option": "ndc", "dist_alpha": True, "depth_range": [0.0, 1.0]}, "dataloading": {"resize_factor": 2},
              "extract_images": {"resolution": [12, 16]}}),
])
def test_reference_train_script_variants(tmp_path, name, over):
    """Other branches of the reference's train.py / eval_poses.py against our packages (CPU stand-in)."""
    import scene_writer
    from dataloading.configloading import update_recursive
    data = str(tmp_path / "data")
    scene_writer.write_scene(data, scene="toy", frames=6, size=(24, 32), factor=2, seed=2)
    out_dir = str(tmp_path / "out")
    cfg = _base_cfg(data, out_dir)
    update_recursive(cfg, over)
    cfg_path = str(tmp_path / (name + ".yaml"))
    with open(cfg_path, "w") as fh:
        yaml.safe_dump(cfg, fh)
    scalars = str(tmp_path / "scalars.json")
    r = _run("train.py", cfg_path, {"DROPIN_SCALARS": scalars})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    tags = {t for t, _, _ in json.load(open(scalars))}
    assert "train/psnr" in tags and "eval/ate_trans" in tags
    if name == "focal_gtpose_ssim":
        assert "train/focalx" in tags and "train/lr_focal" in tags and os.path.isfile(os.path.join(out_dir, "model_focal.pt"))
    if name == "auto_reset":
        assert any(t.startswith("train/scaleview") for t in tags)
    r = _run(os.path.join("evaluation", "eval_poses.py"), cfg_path, {})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


def test_reference_eval_script_pose_initialisations(tmp_path):
    """evaluation/eval.py with every way it initialises the held-out poses (scale / ATE alignment of the ground-truth poses to the
    learned trajectory, none) and with type_to_eval: train (no test-time optimisation)."""
    import scene_writer
    data = str(tmp_path / "data")
    scene_writer.write_scene(data, scene="toy", frames=7, size=(24, 32), seed=3)
    out_dir = str(tmp_path / "out")
    cfg = _base_cfg(data, out_dir)
    cfg_path = str(tmp_path / "base.yaml")
    with open(cfg_path, "w") as fh:
        yaml.safe_dump(cfg, fh)
    r = _run("train.py", cfg_path, {})
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for tag, eval_pose in (("scale", {"init_method": "scale"}), ("ate", {"init_method": "ate"}), ("none", {"init_method": "none"}),
                           ("train", {"type_to_eval": "train"})):
        cfg["eval_pose"] = dict(cfg["eval_pose"], **eval_pose)
        path = str(tmp_path / (tag + ".yaml"))
        with open(path, "w") as fh:
            yaml.safe_dump(cfg, fh)
        r = _run(os.path.join("evaluation", "eval.py"), path, {})
        assert r.returncode == 0, tag + r.stdout[-2000:] + r.stderr[-3000:]
        assert "Mean MSE" in r.stdout, tag
        cfg["eval_pose"].pop("type_to_eval", None)


def test_checkpoints_cross_load_between_reference_and_ours(tmp_path):
    """A checkpoint written by the REFERENCE CheckpointIO (model + Adam state, poses, distortions) loads into our classes with
    identical tensors, and one written by ours loads back into the reference's -- the migration path for existing runs."""
    ref_dir, our_dir = str(tmp_path / "ref"), str(tmp_path / "ours")
    os.makedirs(ref_dir), os.makedirs(our_dir)
    writer = f'''
import sys, torch
sys.path.insert(0, {os.path.join(ROOT, "oracle")!r})
import gen_golden as gg
ref = gg.import_reference()
cfg = gg.base_cfg(128)
torch.manual_seed(5)
net = ref.OfficialStaticNerf(cfg)
model = ref.get_model(ref.Renderer(net, cfg["rendering"], device=torch.device("cpu")), cfg, device=torch.device("cpu"))
pose, dist = ref.LearnPose(4, True, True, cfg), ref.Learn_Distortion(4, True, True, cfg)
with torch.no_grad():
    pose.r.normal_(); pose.t.normal_(); dist.global_scales.uniform_(0.5, 1.5); dist.global_shifts.normal_()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
sum(p.sum() for p in model.parameters()).backward(); opt.step()
mode = sys.argv[1]
if mode == "save":
    ref.CheckpointIO({ref_dir!r}, model=model, optimizer=opt).save("model.pt", epoch_it=7, it=123, loss_val_best=1.5)
    ref.CheckpointIO({ref_dir!r}, model=pose).save("model_pose.pt", epoch_it=7, it=123)
    ref.CheckpointIO({ref_dir!r}, model=dist).save("model_distortion.pt", epoch_it=7, it=123)
    torch.save({{k: v for k, v in model.state_dict().items()}}, {os.path.join(ref_dir, "expect.pt")!r})
else:
    d = ref.CheckpointIO({our_dir!r}, model=model, optimizer=opt).load("model.pt")
    want = torch.load({os.path.join(our_dir, "expect.pt")!r})
    assert d["it"] == 321 and d["epoch_it"] == 9, d
    assert all(torch.equal(v, want[k]) for k, v in model.state_dict().items())
    ref.CheckpointIO({our_dir!r}, model=pose).load("model_pose.pt")
    assert torch.equal(pose.r, want["__pose_r"])
    print("reference loaded ours")
'''
    env = dict(os.environ, PYTHONPATH="")
    r = subprocess.run([sys.executable, "-c", writer, "save"], cwd=REF, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    import torch
    import model as mdl
    from test_host_logic import make_cfg
    cfg = make_cfg(128)
    net = mdl.OfficialStaticNerf(cfg)
    model = mdl.get_model(mdl.Renderer(net, cfg["rendering"], device="cpu"), cfg, device="cpu")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    pose, dist = mdl.LearnPose(4, True, True, cfg), mdl.Learn_Distortion(4, True, True, cfg)
    d = mdl.CheckpointIO(ref_dir, model=model, optimizer=opt).load("model.pt")
    assert d["it"] == 123 and d["epoch_it"] == 7 and d["loss_val_best"] == 1.5
    want = torch.load(os.path.join(ref_dir, "expect.pt"))
    assert all(torch.equal(v, want[k]) for k, v in model.state_dict().items())
    assert len(opt.state) == len(list(model.parameters()))                       # Adam moments came along
    mdl.CheckpointIO(ref_dir, model=pose).load("model_pose.pt")
    mdl.CheckpointIO(ref_dir, model=dist).load("model_distortion.pt")
    assert float(pose.r.detach().abs().sum()) > 0 and float((dist.global_scales.detach() - 1).abs().sum()) > 0
    # and back: ours -> reference
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.25)
    mdl.CheckpointIO(our_dir, model=model, optimizer=opt).save("model.pt", epoch_it=9, it=321)
    mdl.CheckpointIO(our_dir, model=pose).save("model_pose.pt", epoch_it=9, it=321)
    expect = {k: v.clone() for k, v in model.state_dict().items()}
    expect["__pose_r"] = pose.r.detach().clone()
    torch.save(expect, os.path.join(our_dir, "expect.pt"))
    r = subprocess.run([sys.executable, "-c", writer, "load"], cwd=REF, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "reference loaded ours" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
