"""GPU (-m gpu): the REAL multi-rank leg of bench.py, executed on the one GPU of the test box.

The driver launches `bench.py --gpus N` on an 8-GPU node that this repository's tests never see; until now only `--dry-run`
ran the launcher.  Here every line of the multi-rank path runs for real:

* two ranks sharing the GPU (NNR_ALLOW_SHARED_GPU=1: gloo carries the device tensors; RCCL refuses two ranks on one device):
  self-launch under torch.distributed.run, sharded Trainer.train_step, barrier + max-over-ranks timing, the `collective` block,
  rank-0-only printing, `cpu_baseline: null`;
* ONE rank on a real RCCL group (NNR_BENCH_FORCE_DIST=1): `init_process_group('nccl', device_id=...)`, the step's flat gradient
  all-reduce issued on RCCL inside every step, `allreduce_probe` on RCCL.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env, launcher=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra_env)
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]           # rank 0 prints, nobody else does
    return json.loads(lines[0])


COMMON = ["--steps", "3", "--warmup", "1", "--no-extra", "--no-cpu-baseline"]


def test_two_ranks_share_the_gpu_through_the_real_bench_path():
    one = _run(["--gpus", "1"] + COMMON, {})
    two = _run(["--gpus", "2"] + COMMON, {"NNR_ALLOW_SHARED_GPU": "1"})
    assert one["n_gpus"] == 1 and "collective" not in one
    assert two["n_gpus"] == 2 and two["steps"] == 3 and two["warmup"] == 1 and two["scaling"] == "weak"
    assert two["collective"]["rccl_ranks_seen"] == 2
    assert two["collective"]["backend"].startswith("gloo")
    assert two["collective"]["bucket_floats"] > 595844
    assert two["cpu_baseline"] is None
    # the workload per GPU is the one the N = 1 line names; only the parallelism string differs
    assert two["config"]["workload"] == one["config"]["workload"]
    assert two["config"]["rays_per_gpu"] == one["config"]["rays_per_gpu"] == 1024
    assert two["config"]["parallelism"].startswith("dp2")
    # value counts the rays of BOTH ranks.  The two ranks time-share one GPU (each also pays the gloo host round trip of the
    # bucket), so the aggregate is of the order of the single-rank rate -- not 2x as on two GPUs, and not 0.5x as it would be if
    # only one rank's rays were counted against the shared time.
    # (round 6: the single-rank step got 20 % faster, the two-rank step is dominated by gloo's host round trip of the 2.4 MB bucket -- ~20 ms per
    # step -- and did not: measured 0.31 of the single-rank rate; the lower bound only has to exclude "one rank's rays counted", which would be half of that)
    assert 0.2 * one["value"] < two["value"] < 1.5 * one["value"], (one["value"], two["value"])
    assert abs(two["value"] - 2048 / (two["ms_per_step"] * 1e-3)) <= 1e-3 * two["value"]
    # (rank 0's kernels share the chip with rank 1's: the per-kernel durations, and with them the roofline fraction, are those of half a GPU)
    assert two["roofline"]["kernel"] in ("mlp_fwd", "mlp_dgrad", "mlp_wgrad") and 0.05 < two["roofline"]["frac"] < 1.05
    print("bench --gpus 1: %.0f rays/s; --gpus 2 on one GPU: %.0f rays/s, all-reduce %.0f us (gloo)"
          % (one["value"], two["value"], two["collective"]["allreduce_us"]))


def test_one_rank_on_a_real_rccl_group():
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                "--master-port", "29631"]
    line = _run(["--gpus", "1"] + COMMON, {"NNR_BENCH_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, launcher)
    assert line["n_gpus"] == 1
    assert line["collective"]["backend"] == "rccl" and line["collective"]["rccl_ranks_seen"] == 1
    assert 0 < line["collective"]["allreduce_us"] < 5000
    assert line["value"] > 1e4 and line["final_loss"] == line["final_loss"]      # a finite loss after steps that all-reduced on RCCL
    print("one-rank RCCL group: %.0f rays/s, flat all-reduce of %d floats %.0f us"
          % (line["value"], line["collective"]["bucket_floats"], line["collective"]["allreduce_us"]))
    # the distributed plumbing of a rank (process group, the in-place all-reduce of the gradient buffer issued on RCCL in
    # every step) must not cost the step more than 3 % -- compared on the MEDIAN step of 20, which one slow step cannot move
    args = ["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extra", "--no-cpu-baseline"]
    plain = _run(args, {})
    forced = _run(args, {"NNR_BENCH_FORCE_DIST": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}, launcher[:-1] + ["29633"])
    a, b = plain["step_ms"]["median"], forced["step_ms"]["median"]
    print("median step: plain %.3f ms, one rank of an RCCL group %.3f ms (%+.1f %%)" % (a, b, 100 * (b - a) / a))
    assert b <= 1.03 * a + 0.02, (a, b)


def test_strong_scaling_switch_and_config4_rank_shape():
    """`--total-rays T`: the step is T rays in total and each of the N ranks renders T / N of them (strong scaling; the default is weak).
    Two ranks sharing the GPU split a 2048-ray step; then BASELINE configs[3]'s per-rank shape -- 4096 rays x 128 samples, bf16 products --
    through the same two-rank path (8192 rays per step)."""
    strong = _run(["--gpus", "2", "--total-rays", "2048"] + COMMON, {"NNR_ALLOW_SHARED_GPU": "1"})
    assert strong["scaling"] == "strong" and strong["n_gpus"] == 2
    assert strong["config"]["rays_per_gpu"] == 1024 and strong["config"]["total_rays"] == 2048
    assert abs(strong["value"] - 2048 / (strong["ms_per_step"] * 1e-3)) <= 1e-3 * strong["value"]
    c4 = _run(["--gpus", "2", "--rays-per-gpu", "4096", "--samples", "128", "--bf16"] + COMMON, {"NNR_ALLOW_SHARED_GPU": "1"})
    assert c4["scaling"] == "weak" and c4["config"]["rays_per_gpu"] == 4096 and c4["config"]["total_rays"] == 8192
    assert c4["collective"]["rccl_ranks_seen"] == 2 and c4["dtype"].startswith("bf16")
    assert c4["value"] > 1e5 and c4["final_loss"] == c4["final_loss"]
    print("strong scaling, 2048 rays over 2 ranks on one GPU: %.0f rays/s; configs[3] rank shape (2 x 4096 x 128, bf16): %.0f rays/s"
          % (strong["value"], c4["value"]))
