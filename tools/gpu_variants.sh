# usage: bash tools/gpu_variants.sh v1 v2 ...  -- time the MLP kernels (bf16, 4096 x 128) with the product library and with the
# profiling variants built by `python nope-nerf_amd/csrc/build.py --variant <name> <DEFINES...>` (results of variants are NOT valid)
mkdir -p gpurun_out/variants
export PYTHONUNBUFFERED=1
for v in product "$@"; do
  if [ "$v" != product ]; then export NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_$v.so; else unset NNR_LIB; fi
  echo -n "$v: "; timeout 200 python tools/time_kernels.py ${SHAPE:-4096 128 bf16} 20 2>&1 | tail -1
done | tee gpurun_out/variants/time.txt
