"""CPU, authoring container only: the reference's OWN unmodified train.py under a multi-process launcher -- what
`torchrun --nproc-per-node N train.py configs/...yaml` (BASELINE configs[3]) starts: N processes with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in the environment.  train.py never calls init_process_group and asks for the device "cuda"
(/root/reference/train.py:18-60); `import model` joins the group by itself (nnr.parallel.auto_init), the Trainer shards the step's rays,
one flat all-reduce sums the gradients, and rank 0 alone writes checkpoints / visualisations / the config backup.  Two gloo ranks must
land where the single process lands (the bars of tests/test_parallel_gloo.py::test_two_rank_training_loop_follows_the_single_process_run).
The render operator is the oracle-backed CPU stand-in (tests/dropin_runner.py); tests/test_gpu_bench_ranks.py runs the same entry with
two ranks sharing the one GPU of the GPU box."""
import json
import os
import socket
import subprocess
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
REF = os.environ.get("NNR_REFERENCE", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "train.py")), reason="reference checkout not present")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, cfg_path, tmp_path, tag, extra_env=None):
    """`torchrun --nproc-per-node world train.py cfg` by hand (the launcher only sets these variables and starts the processes)."""
    port = _free_port()
    procs, outs = [], []
    for rank in range(world):
        scal = str(tmp_path / ("scalars_%s_%d.json" % (tag, rank)))
        env = dict(os.environ, PYTHONPATH="", DROPIN_SCALARS=scal, OMP_NUM_THREADS="2", **(extra_env or {}))
        if world > 1:
            env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
        else:
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE"):
                env.pop(k, None)
        cmd = [sys.executable, os.path.join(ROOT, "tests", "dropin_runner.py"), os.path.join(REF, "train.py"), cfg_path]
        procs.append(subprocess.Popen(cmd, cwd=REF, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs.append(scal)
    logs = []
    for p in procs:
        out, _ = p.communicate(timeout=900)
        logs.append(out)
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-4000:]
    return [json.load(open(f)) for f in outs], logs


def _cfg(data, out_dir):
    return {
        "model": {"hidden_dim": 128},
        "dataloading": {"path": data, "scene": ["toy"], "n_workers": 0, "resize_factor": None, "sample_rate": 10 ** 6, "spherify": False},
        "rendering": {"num_points": 8},
        "pose": {"learn_pose": True},
        "training": {"out_dir": out_dir, "n_training_points": 24, "scheduling_start": 2, "scheduling_epoch": 1, "annealing_epochs": 1,
                     "print_every": 4, "checkpoint_every": 4, "visualize_every": 8, "vis_resolution": [6, 8], "pc_ratio": 2,
                     "auto_scheduler": False},
    }


def test_unmodified_train_py_under_two_ranks_lands_on_the_single_process_run(tmp_path):
    import scene_writer
    data = str(tmp_path / "data")
    scene_writer.write_scene(data, scene="toy", frames=4, size=(24, 32), seed=6)
    last = {}
    for world in (1, 2):
        out_dir = str(tmp_path / ("out%d" % world))
        cfg_path = str(tmp_path / ("toy%d.yaml" % world))
        with open(cfg_path, "w") as fh:
            yaml.safe_dump(_cfg(data, out_dir), fh)
        scalars, logs = _launch(world, cfg_path, tmp_path, "w%d" % world)
        assert "ATE:" in logs[0] and "PSNR:" in logs[0]
        for f in ("model.pt", "model_pose.pt", "model_distortion.pt"):
            assert os.path.isfile(os.path.join(out_dir, f)), f
        assert os.path.isfile(os.path.join(out_dir, "backup", "config.yaml"))
        by_tag = {}
        for tag, value, step in scalars[0]:
            by_tag.setdefault(tag, []).append((step, value))
        last[world] = {t: v[-1][1] for t, v in by_tag.items()}
        n_steps = len(by_tag["train/loss"])
        if world == 2:
            # every rank logged the same (all-reduced) scalars
            other = {}
            for tag, value, step in scalars[1]:
                other.setdefault(tag, []).append((step, value))
            for t in ("train/loss", "train/loss_rgb", "train/l2_mean", "eval/ate_trans"):
                a, b = [v for _, v in by_tag[t]], [v for _, v in other[t]]
                assert len(a) == len(b) and max(abs(x - y) for x, y in zip(a, b)) <= 1e-6 * max(1.0, max(abs(x) for x in a)), t
        assert n_steps >= 3, n_steps
    a, b = last[1], last[2]
    # the bars of tests/test_parallel_gloo.py:137 (fp32 summation order of the all-reduce through Adam: 1e-5 in the poses)
    assert abs(a["train/psnr"] - b["train/psnr"]) <= 2e-3, (a["train/psnr"], b["train/psnr"])
    assert abs(a["eval/ate_trans"] - b["eval/ate_trans"]) <= 1e-4 and abs(a["eval/rpe_rot"] - b["eval/rpe_rot"]) <= 5e-3, (a, b)
    for t in ("train/loss", "train/loss_rgb", "train/loss_depth", "train/loss_pc", "train/loss_rgb_s"):
        assert abs(a[t] - b[t]) <= 2e-4 * max(1.0, abs(a[t])), (t, a[t], b[t])


def test_only_rank_zero_writes_files(tmp_path, monkeypatch):
    import torch
    import model as mdl
    from nnr import parallel
    net = torch.nn.Linear(2, 2)
    io = mdl.CheckpointIO(str(tmp_path), model=net)
    monkeypatch.setattr(parallel, "rank", lambda: 1)
    io.save("model.pt", it=3)
    io.backup_model_best("model.pt")
    assert os.listdir(str(tmp_path)) == []
    monkeypatch.setattr(parallel, "rank", lambda: 0)
    io.save("model.pt", it=3)
    assert os.listdir(str(tmp_path)) == ["model.pt"]


def test_auto_init_is_a_no_op_without_a_launcher(monkeypatch):
    from nnr import parallel
    monkeypatch.setitem(parallel._auto, "done", False)
    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert parallel.auto_init() is False and parallel.world_size() == 1
