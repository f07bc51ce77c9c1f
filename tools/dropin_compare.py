"""Helper of tools/gpu_dropin.sh: `prepare` writes the synthetic scene and the two config files (same content, different out_dir)
under gpurun_stage/; `compare` puts the scalars train.py logged on the GPU (HIP kernels) beside the ones it logged on the CPU
(oracle-backed operator) -- same script, same draws -- and writes gpurun_out/dropin/compare.json."""
import json
import os
import sys

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, "gpurun_stage")
ITERS_PER_EPOCH = 6          # 8 frames, sample_rate 4 holds out 2


def prepare():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scene_writer
    scene_writer.write_scene(os.path.join(STAGE, "scene"), scene="toy", frames=8, size=(48, 64), seed=7)
    for tag in ("cpu", "gpu"):
        cfg = {
            "model": {"hidden_dim": 128},
            # paths relative to the staged reference directory the script runs in (the stage travels to the GPU box as a whole)
            "dataloading": {"path": "../scene", "scene": ["toy"], "n_workers": 0, "resize_factor": None, "sample_rate": 4, "spherify": False},
            "rendering": {"num_points": 32},
            "pose": {"learn_pose": True},
            "training": {"out_dir": "../out_%s" % tag, "n_training_points": 256, "scheduling_start": 9, "scheduling_epoch": 1,
                         "annealing_epochs": 1, "print_every": 1, "checkpoint_every": 1000, "visualize_every": 1000, "eval_pose_every": 3,
                         "vis_resolution": [6, 8], "pc_ratio": 2, "auto_scheduler": False},
            "extract_images": {"resolution": [48, 64], "N_novel_imgs": 5},
        }
        with open(os.path.join(STAGE, "dropin_%s.yaml" % tag), "w") as fh:
            yaml.safe_dump(cfg, fh)


def compare():
    cpu = json.load(open(os.path.join(STAGE, "scalars_cpu.json")))
    gpu = json.load(open(os.path.join(ROOT, "gpurun_out", "dropin", "scalars_gpu.json")))
    by = lambda rows: {(t, s): v for t, v, s in rows}
    a, b = by(cpu), by(gpu)
    keys = sorted(set(a) & set(b), key=lambda k: (k[1], k[0]))
    assert keys and set(a) == set(b), (len(a), len(b))
    out = {"scalars_logged": len(keys), "steps": max(s for _, s in keys) + 1, "tags": sorted({t for t, _ in keys})}
    for tag in ("train/loss", "train/loss_rgb", "train/loss_depth", "train/loss_pc", "train/loss_rgb_s", "train/l2_mean"):
        rows = [(s, a[(t, s)], b[(t, s)]) for t, s in keys if t == tag]
        if not rows:
            continue
        dev = [abs(x - y) / max(1.0, abs(x)) for _, x, y in rows]
        out[tag] = {"n": len(rows), "first_10_max_dev": max(dev[:10]), "max_dev": max(dev), "cpu_last": rows[-1][1], "gpu_last": rows[-1][2]}
    for tag in ("eval/ate_trans", "eval/rpe_rot", "train/psnr"):
        rows = [(s, a[(t, s)], b[(t, s)]) for t, s in keys if t == tag]
        if rows:
            out[tag] = {"cpu_last": rows[-1][1], "gpu_last": rows[-1][2], "steps": [s for s, _, _ in rows]}
    os.makedirs(os.path.join(ROOT, "gpurun_out", "dropin"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "dropin", "compare.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))
    worst = out["train/loss"]["first_10_max_dev"]
    assert worst <= 1e-3, "the first ten steps of train.py on the HIP kernels deviate from the CPU run by %.2e" % worst


if __name__ == "__main__":
    {"prepare": prepare, "compare": compare}[sys.argv[1]]()
