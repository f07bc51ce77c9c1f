# bf16 fwd / dgrad ablations at BASELINE configs[2] (profiling variants: results NOT valid)
mkdir -p gpurun_out/r2c
export PYTHONUNBUFFERED=1
R=$PWD
for v in product nostash stashl2 noside vmcnt8 nosidestash; do
  if [ "$v" != product ]; then export NNR_LIB=$R/nope-nerf_amd/nnr/libnnr_$v.so; else unset NNR_LIB; fi
  timeout 200 python tools/time_kernels.py 4096 128 bf16 5 2>/dev/null | tail -1
done | tee gpurun_out/r2c/bf16_ablate.txt
unset NNR_LIB
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scene_training.py -m gpu -q --timeout 250 -k "bf16" 2>&1 | tail -12 | tee gpurun_out/r2c/pytest_bf16.txt
