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
