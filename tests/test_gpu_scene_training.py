"""GPU (-m gpu): end-to-end on the HIP kernels -- a synthetic scene written in the reference's on-disk layout
(tools/scene_writer.py), read by `dataloading` (HBM-resident loader), trained jointly with poses and depth distortion by
`model.Trainer` (first training phase: rgb + depth + point-cloud + surface-reprojection losses), scored with `utils_poses`.
The photometric error must fall and the learned trajectory must move from the identity initialisation towards the ground truth."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("nope-nerf_amd", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("style,epochs,ate_gain", [("tanks", 200, 0.75), ("llff", 100, None)])
def test_scene_trains_and_poses_improve(tmp_path, style, epochs, ate_gain):
    """Measured on this scene (r01): tanks style ATE 0.124 -> 0.070 / PSNR 11.7 -> 21.7 dB after 200 epochs; the NDC style needs
    the larger frames of tools/gpu_scene.sh to move the poses within a test-sized budget, so only its photometric error is held."""
    import scene_writer
    import train_scene
    scene_writer.write_scene(str(tmp_path), scene="toy", frames=8, size=(60, 80), seed=1)
    res = train_scene.run(str(tmp_path), "toy", style=style, epochs=epochs, log_every=25, n_rays=512, n_samples=64, hidden=128,
                          sample_rate=10 ** 6)
    first, last = res["curve"][1], res["curve"][-1]
    assert res["steps"] == epochs * 8 and res["loader"] == "resident"
    assert last["psnr"] > first["psnr"] + 5.0 and last["psnr"] > 17.5, res["curve"]
    assert all(torch.isfinite(torch.tensor([c["ate"], c["rpe_rot_deg"]])).all() for c in res["curve"][1:])
    if ate_gain is not None:        # identity initialisation (ATE undefined there) -> towards the true trajectory
        assert last["ate"] < ate_gain * first["ate"], res["curve"]
    else:
        assert last["ate"] < 1.05 * first["ate"], res["curve"]


def test_host_and_resident_loaders_train_identically(tmp_path):
    """Same seed, same sampler: the loader that keeps the scene in HBM must feed the very same steps as the host DataLoader."""
    import scene_writer
    import train_scene
    scene_writer.write_scene(str(tmp_path), scene="toy", frames=6, size=(48, 64), seed=3)
    out = {}
    for resident in (True, False):
        out[resident] = train_scene.run(str(tmp_path), "toy", epochs=3, log_every=1, n_rays=128, n_samples=32, hidden=128,
                                        sample_rate=10 ** 6, resident=resident)
    a, b = out[True]["curve"][-1], out[False]["curve"][-1]
    assert abs(a["psnr"] - b["psnr"]) < 1e-3 and abs(a["ate"] - b["ate"]) < 1e-5, (a, b)


def test_bf16_training_converges_like_fp32(tmp_path):
    """BASELINE configs[2] arithmetic end to end: the same scene, seed and schedule trained with bf16 MFMA products must land where
    the fp32 run lands -- no more than 0.5 dB below its PSNR, no more than 10 % above its ATE (one-sided: training is chaotic, two
    runs that differ in rounding end a few tenths of a dB apart either way; r02 measured fp32 22.11 dB / ATE 0.0660, bf16 22.66 dB /
    0.0579).  The bf16 gradients differ from fp32 by 5-15 % in relative L2 per step (tests/test_gpu_parity.py), unbiased rounding
    noise well below the noise of the 512-ray batches."""
    import scene_writer
    import train_scene
    scene_writer.write_scene(str(tmp_path), scene="toy", frames=8, size=(60, 80), seed=1)
    res = {}
    for dt in ("fp32", "bf16"):
        res[dt] = train_scene.run(str(tmp_path), "toy", style="tanks", epochs=200, log_every=50, n_rays=512, n_samples=64, hidden=256,
                                  sample_rate=10 ** 6, mfma_dtype=dt)
    a, b = res["fp32"]["curve"][-1], res["bf16"]["curve"][-1]
    print("fp32: PSNR %.2f dB ATE %.4f | bf16: PSNR %.2f dB ATE %.4f" % (a["psnr"], a["ate"], b["psnr"], b["ate"]))
    assert a["psnr"] > 17.5 and b["psnr"] >= a["psnr"] - 0.5, (a, b)
    assert b["ate"] <= 1.10 * a["ate"], (a, b)
