mkdir -p gpurun_out/r2q
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k bf16 2>&1 | tail -1 > gpurun_out/r2q/tests.txt
for v in "" burst8 burst2; do
  if [ -n "$v" ]; then export NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_$v.so; fi
  timeout 200 python tools/time_kernels.py 4096 128 bf16 20 2>&1 | tail -1 >> gpurun_out/r2q/time.txt
done
cat gpurun_out/r2q/tests.txt gpurun_out/r2q/time.txt
