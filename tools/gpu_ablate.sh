set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
for v in "" nostash nomask nostashmask noflush nodmasync; do
  echo "== variant=${v:-product}"
  if [ -n "$v" ]; then export NNR_LIB=$R/nope-nerf_amd/nnr/libnnr_$v.so; else unset NNR_LIB; fi
  timeout 300 python tools/profile_kernels.py 5
done > gpurun_out/ablate.txt 2>&1
grep -E "variant|kernels" gpurun_out/ablate.txt
