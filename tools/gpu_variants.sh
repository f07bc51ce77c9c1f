# usage: bash tools/gpu_variants.sh v1 v2 ...   (times profile_kernels.py with each libnnr_<v>.so, "" = product)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
for v in product "$@"; do
  echo "== variant=$v"
  if [ "$v" != product ]; then export NNR_LIB=$R/nope-nerf_amd/nnr/libnnr_$v.so; else unset NNR_LIB; fi
  timeout 300 python tools/profile_kernels.py 5 2>&1 | grep kernels
done | tee gpurun_out/variants.txt
