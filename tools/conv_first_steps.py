"""Per-step deviation of the first steps of the HIP convergence replay from the reference run (tests/golden/conv_llff.npz), per logged
loss part, for switches of the first-phase path:   python tools/conv_first_steps.py [--no-fuse-pair] [--steps 24]
(NNR_PC_SEARCH=brute in the environment selects the exhaustive nearest-neighbour search)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "nope-nerf_amd")):
    sys.path.insert(0, p)
import numpy as np
import pytest
import torch

import test_conv_reference as T

if __name__ == "__main__":
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 24
    if "--no-fuse-pair" in sys.argv:
        import train_scene
        real = train_scene.scene_cfg

        def cfg(*a, **k):
            c = real(*a, **k)
            c["training"]["fuse_pair"] = False
            return c
        train_scene.scene_cfg = cfg
    mp = pytest.MonkeyPatch()
    with tempfile.TemporaryDirectory() as d:
        import pathlib
        losses, *_ = T._replay(pathlib.Path(d), torch.device("cuda"), mp, steps)
    mp.undo()
    ref = T.GOLD["losses"][:steps]
    dev = np.abs(losses - ref) / np.maximum(1.0, np.abs(ref))
    print("columns:", " ".join(T.LOGGED))
    for s in range(steps):
        print("step %2d  " % s + " ".join("%.1e" % v for v in dev[s]))
    print("max over the first 20 steps: %.3e" % dev[:20].max())
