mkdir -p gpurun_out/r2d
export PYTHONUNBUFFERED=1
R=$PWD
for v in product freesched nomask; do
  if [ "$v" != product ]; then export NNR_LIB=$R/nope-nerf_amd/nnr/libnnr_$v.so; else unset NNR_LIB; fi
  timeout 200 python tools/time_kernels.py 4096 128 bf16 5 2>/dev/null | tail -1
done | tee gpurun_out/r2d/bf16_ablate2.txt
