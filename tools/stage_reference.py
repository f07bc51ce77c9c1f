"""Stage the reference's own Python sources for the GPU box:  python tools/stage_reference.py

The reference is pure Python and lives at /root/reference in the authoring container only.  Its sources never enter this
repository's history; they are COPIED, unmodified, into the git-ignored `gpurun_stage/reference_cpu/` (which gpurun ships to the GPU box like
the built libnnr.so) so that bench.py's `cpu_baseline` can time the REFERENCE ITSELF on the box's host cores (`kind: "reference"`, tools/cpu_reference_baseline.py;
  BASELINE.md section 3) -- it falls back to the oracle port, and says so, when nothing is staged;
(tools/gpu_dropin.sh stages train.py the same way for tests/test_gpu_dropin.py.)

Called by __graft_entry__.build() whenever /root/reference is present."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = os.path.join(ROOT, "gpurun_stage", "reference_cpu")      # (its own directory: gpurun_stage/ref/ is the drop-in test's cwd)


def stage(ref=None):
    ref = ref or os.environ.get("NNR_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "model")):
        return None
    os.makedirs(os.path.join(STAGE, "model"), exist_ok=True)
    os.makedirs(os.path.join(STAGE, "configs"), exist_ok=True)
    for name in sorted(os.listdir(os.path.join(ref, "model"))):
        if name.endswith(".py"):
            shutil.copyfile(os.path.join(ref, "model", name), os.path.join(STAGE, "model", name))
    shutil.copyfile(os.path.join(ref, "configs", "default.yaml"), os.path.join(STAGE, "configs", "default.yaml"))
    # BASELINE configs[0] names configs/Tanks/Ignatius.yaml: bench.py's `cpu_32x64_d128` block runs the reference's Trainer.train_step on it
    os.makedirs(os.path.join(STAGE, "configs", "Tanks"), exist_ok=True)
    shutil.copyfile(os.path.join(ref, "configs", "Tanks", "Ignatius.yaml"), os.path.join(STAGE, "configs", "Tanks", "Ignatius.yaml"))
    return STAGE


if __name__ == "__main__":
    out = stage(sys.argv[1] if len(sys.argv) > 1 else None)
    print("staged reference sources in", out) if out else print("no reference checkout: nothing staged")
