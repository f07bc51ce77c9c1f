"""Rate of the reference's UNMODIFIED train.py on the HIP kernels against the bench rate of the same shape:

    python tools/loop_rate.py [--out profiles/r04/scene_loop.json] [--modes auto auto_noaux host0 host1] [--epochs 40]

Needs the staged script (gpurun_stage/ref/train.py + configs/default.yaml: tools/gpu_dropin.sh, authoring container) and a GPU.  Writes a
16-frame 540 x 960 synthetic scene (tools/scene_writer.py), then runs train.py through tests/dropin_runner.py (our model / dataloading /
utils_poses packages ahead of the reference's on the import path, HIP kernels) with a YAML that sets ONLY what a scene file of the
reference sets (path, scene, out_dir) plus the shape under test (n_training_points 1024, num_points 192: BASELINE configs[1]) and the
length of the run -- no key the reference does not have, except in the `host*` modes, which force the host loader (`dataloading.resident:
False`) to show what the default used to cost.  Modes:
    auto        nothing said about the loader: the scene goes resident by itself (dataloading/dataloading.py); first-phase losses on (default)
    auto_noaux  the same with pc_weight = rgb_s_weight = 0 (the second phase's step: the bench headline's shape)
    host0/host1 the reference's DataLoader, n_workers 0 / 1 (its default), one collated host batch per step
The loop's rate = iterations between epoch-end marks / wall clock (the first 5 epochs are warm-up), everything train.py does included:
three .item() syncs per step, scalar logging, the per-epoch pose evaluation (eval_pose_every 1) and PSNR.  The bench rate beside it =
model.Trainer.train_step on a resident synthetic batch of the same shape, bench.py's own timed-step loop, measured in this process."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, "gpurun_stage")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nope-nerf_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def staged():
    return os.path.isfile(os.path.join(STAGE, "ref", "train.py")) and os.path.isfile(os.path.join(STAGE, "ref", "configs", "default.yaml"))


def run_mode(mode, scene_root, scene, rays, samples, epochs, work):
    import yaml
    aux = not mode.endswith("noaux")
    cfg = {"dataloading": {"path": scene_root, "scene": [scene]},
           "rendering": {"num_points": samples},
           "training": {"out_dir": os.path.join(work, "out_" + mode), "n_training_points": rays, "scheduling_start": epochs, "scheduling_epoch": 0,
                        "vis_reprojection_every": 10 ** 9}}
    if not aux:
        cfg["training"].update(pc_weight=[0.0, 0.0], rgb_s_weight=[0.0, 0.0])
    if mode.startswith("host"):
        cfg["dataloading"].update(resident=False, n_workers=int(mode[4:5]))
    ypath = os.path.join(work, mode + ".yaml")
    with open(ypath, "w") as fh:
        yaml.safe_dump(cfg, fh)
    times = os.path.join(work, mode + "_times.json")
    env = dict(os.environ, DROPIN_BACKEND=os.environ.get("LOOP_BACKEND", "hip"), DROPIN_TIMES=times, NNR_REFERENCE=os.path.join(STAGE, "ref"), PYTHONPATH="")
    t0 = time.perf_counter()
    wrap = os.environ.get("LOOP_TRACE_DIR")      # LOOP_TRACE_DIR=/tmp/x: the child under rocprofv3 --kernel-trace --stats (its rate is then the tracer's, not the loop's)
    pre = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", wrap, "-o", "child_" + mode, "--"] if wrap else []
    r = subprocess.run(pre + [sys.executable, os.path.join(ROOT, "tests", "dropin_runner.py"), "train.py", ypath], cwd=os.path.join(STAGE, "ref"), env=env,
                       capture_output=True, text=True, timeout=900)
    wall = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError("train.py failed in mode %s:\n%s\n%s" % (mode, r.stdout[-1500:], r.stderr[-3000:]))
    marks = json.load(open(times))
    skip = min(5, len(marks) - 2)
    (i0, t0m), (i1, t1m) = marks[skip], marks[-1]
    its = i1 - i0
    loader = "resident" if "views (resident on" in r.stdout else "host"
    return {"mode": mode, "loader": loader, "aux_per_image_losses": aux, "rays": rays, "samples": samples, "epochs": len(marks), "iterations_timed": its,
            "ms_per_iteration": round((t1m - t0m) / its * 1e3, 4), "rays_per_s": round(rays * its / (t1m - t0m), 1),
            "process_wall_s": round(wall, 1), "yaml": cfg}


def bench_rate(rays, samples, aux, steps=40, warmup=10):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    trainer, net = bench.build_trainer(dev, 1, aux, False, rays, samples)
    data = bench.synthetic_batch(dev)
    elapsed, _, in_step = bench._timed_steps(trainer, data, warmup, steps, False)
    del trainer, net, data
    torch.cuda.empty_cache()
    return {"ms_per_step": round(elapsed / steps * 1e3, 4), "rays_per_s": round(rays * steps / elapsed, 1), "aux_per_image_losses": aux,
            "step_ms": in_step["step_ms"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "scene_loop.json"))
    ap.add_argument("--modes", nargs="+", default=["auto_noaux", "auto", "host0", "host1"])
    ap.add_argument("--epochs", type=int, default=40)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=192)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, nargs=2, default=[540, 960])
    ap.add_argument("--work", default="/tmp/loop_rate")
    a = ap.parse_args()
    if not staged():
        raise SystemExit("the reference's train.py is not staged (tools/gpu_dropin.sh, authoring container)")
    import scene_writer
    os.makedirs(a.work, exist_ok=True)
    scene = "loop%dx%d" % tuple(a.size)
    if not os.path.isdir(os.path.join(a.work, "scenes", scene)):
        scene_writer.write_scene(os.path.join(a.work, "scenes"), scene=scene, frames=a.frames, size=tuple(a.size), seed=0)
    out = {"what": __doc__.split("\n")[0], "scene": "%d frames of %d x %d (tools/scene_writer.py), sample_rate 8 holds out every 8th" % (a.frames, *a.size),
           "bench": {}, "loop": []}
    for aux in sorted({not m.endswith("noaux") for m in a.modes}):
        out["bench"]["aux" if aux else "noaux"] = bench_rate(a.rays, a.samples, aux)
    for m in a.modes:
        epochs = a.epochs if not m.startswith("host") else max(8, a.epochs // 4)      # the host loader is 10-50x slower per step
        res = run_mode(m, os.path.join(a.work, "scenes"), scene, a.rays, a.samples, epochs, a.work)
        ref = out["bench"]["aux" if res["aux_per_image_losses"] else "noaux"]
        res["fraction_of_bench_rate"] = round(res["rays_per_s"] / ref["rays_per_s"], 4)
        out["loop"].append(res)
        print(m, res["loader"], res["rays_per_s"], "rays/s =", res["fraction_of_bench_rate"], "of the bench rate", ref["rays_per_s"], flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({m["mode"]: m["fraction_of_bench_rate"] for m in out["loop"]}))


if __name__ == "__main__":
    main()
