mkdir -p gpurun_out/r2o
export PYTHONUNBUFFERED=1
for v in "" noside nodma nosync nosidestash; do
  if [ -n "$v" ]; then export NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_$v.so; fi
  timeout 200 python tools/time_kernels.py 4096 128 bf16 20 2>&1 | tail -1 >> gpurun_out/r2o/time.txt
done
cat gpurun_out/r2o/time.txt
