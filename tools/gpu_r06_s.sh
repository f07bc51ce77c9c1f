# round 6, GPU call s: long training runs under the two-term products (weights grow over thousands of steps): joint pose + field training on a
# synthetic scene, per-image losses on / off, two-term against six-term products -- no NaN, same PSNR / ATE
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
python tools/scene_writer.py /tmp/scenes --frames 16 --size 120 160 > /dev/null 2>&1
for kind in split2 split3; do for aux in "" "--no-aux"; do
  echo "== NNR_FP32_PRODUCTS=$kind $aux"
  NNR_FP32_PRODUCTS=$kind timeout 900 python tools/train_scene.py /tmp/scenes synthetic --epochs 400 --samples 192 $aux --log-every 100 2>&1 | grep -E "epoch|PSNR|ATE|nan|NaN|Error|rays/s" | tail -6
done; done | tee gpurun_out/r06/s_long_training_split2_vs_split3.txt
