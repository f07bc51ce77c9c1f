"""GPU (-m gpu): the reference's UNMODIFIED train.py on the HIP kernels -- Python surface and kernels together (tensorboard scalars,
.item() syncs, DataLoader batches, checkpoints, the deferred NaN flag, the pack cache).  The script and configs/default.yaml are
STAGED, not committed: tools/gpu_dropin.sh copies them from the reference checkout into gpurun_stage/ (git-ignored, shipped to the
GPU box like the built library) and records a CPU run of the same script with the oracle-backed operator; this test runs the GPU
leg and compares the logged scalars step by step (same draws: DROPIN_CPU_DRAWS).  Skipped where nothing is staged."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, "gpurun_stage")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.isfile(os.path.join(STAGE, "ref", "train.py")) and os.path.isfile(os.path.join(STAGE, "scalars_cpu.json"))),
                                 reason="reference train.py not staged (run tools/gpu_dropin.sh in the authoring container)")]


def test_reference_train_script_on_the_hip_kernels_tracks_its_cpu_run():
    ref = os.path.join(STAGE, "ref")
    out = os.path.join(ROOT, "gpurun_out", "dropin")
    os.makedirs(out, exist_ok=True)
    import shutil
    shutil.rmtree(os.path.join(STAGE, "out_gpu"), ignore_errors=True)        # train.py resumes from a checkpoint it finds in out_dir
    env = dict(os.environ, DROPIN_BACKEND="hip", DROPIN_CPU_DRAWS="1", DROPIN_SCALARS=os.path.join(out, "scalars_gpu.json"), NNR_REFERENCE=ref,
               PYTHONPATH="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_runner.py"), "train.py", os.path.join("..", "dropin_gpu.yaml")],
                       cwd=ref, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "ATE:" in r.stdout and "PSNR:" in r.stdout
    c = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dropin_compare.py"), "compare"], capture_output=True, text=True, timeout=120)
    assert c.returncode == 0, c.stdout[-2000:] + c.stderr[-2000:]
    res = json.loads(c.stdout.strip().splitlines()[-1])
    print("train.py on HIP vs its CPU run: %d scalars over %d steps; loss deviation first 10 steps %.2e, all steps %.2e; final PSNR %.2f / %.2f"
          % (res["scalars_logged"], res["steps"], res["train/loss"]["first_10_max_dev"], res["train/loss"]["max_dev"],
             res["train/psnr"]["gpu_last"], res["train/psnr"]["cpu_last"]))
    assert res["train/loss"]["first_10_max_dev"] <= 1e-3


def test_reference_train_script_under_two_ranks_sharing_the_gpu():
    """`torchrun --nproc-per-node 2 train.py cfg` by hand, HIP kernels: two processes with the launcher's environment, the UNMODIFIED script.
    `import model` binds the device and joins the group (nnr.parallel.auto_init; NNR_DIST_BACKEND=gloo because RCCL refuses two ranks on one
    device), each rank renders half of the step's rays, one flat all-reduce, rank 0 writes.  Against the single-process run of the same
    configuration on the same GPU: the logged losses of the first epoch to 1e-4 (same draws: both ranks and the single process draw the
    step's whole permutation and jitter), PSNR / ATE at the end inside the chaos envelope of 2 dB / 30 %."""
    import socket
    ref = os.path.join(STAGE, "ref")
    out = os.path.join(ROOT, "gpurun_out", "dropin")
    os.makedirs(out, exist_ok=True)
    import shutil
    import yaml
    cfg = yaml.safe_load(open(os.path.join(STAGE, "dropin_gpu.yaml")))
    results = {}
    for world in (1, 2):
        out_dir = os.path.join(STAGE, "out_gpu_w%d" % world)
        shutil.rmtree(out_dir, ignore_errors=True)
        cfg["training"]["out_dir"] = out_dir
        cfg_path = os.path.join(STAGE, "dropin_gpu_w%d.yaml" % world)
        with open(cfg_path, "w") as fh:
            yaml.safe_dump(cfg, fh)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for rank in range(world):
            env = dict(os.environ, DROPIN_BACKEND="hip", DROPIN_SCALARS=os.path.join(out, "scalars_gpu_w%d_r%d.json" % (world, rank)), NNR_REFERENCE=ref,
                       PYTHONPATH="")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE"):
                env.pop(k, None)
            if world > 1:
                env.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                           NNR_DIST_BACKEND="gloo")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dropin_runner.py"), "train.py", cfg_path], cwd=ref, env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        logs = [p.communicate(timeout=900)[0] for p in procs]
        for p, log in zip(procs, logs):
            assert p.returncode == 0, log[-4000:]
        by = {}
        for tag, value, step in json.load(open(os.path.join(out, "scalars_gpu_w%d_r0.json" % world))):
            by.setdefault(tag, []).append(value)
        results[world] = by
        assert os.path.isfile(os.path.join(out_dir, "model.pt"))
    a, b = results[1], results[2]
    n = min(len(a["train/loss"]), len(b["train/loss"]))
    first = max(abs(x - y) / max(1.0, abs(x)) for x, y in zip(a["train/loss"][:3], b["train/loss"][:3]))
    print("train.py, 2 ranks on one GPU vs 1: %d logged steps, first-3 loss deviation %.2e, final PSNR %.2f / %.2f, ATE %.4f / %.4f"
          % (n, first, a["train/psnr"][-1], b["train/psnr"][-1], a["eval/ate_trans"][-1], b["eval/ate_trans"][-1]))
    assert first <= 1e-4, first
    assert abs(a["train/psnr"][-1] - b["train/psnr"][-1]) <= 2.0
    assert abs(a["eval/ate_trans"][-1] - b["eval/ate_trans"][-1]) <= 0.3 * max(a["eval/ate_trans"][-1], 1e-3)
