# usage: bash tools/gpu_knob.sh ENVVAR v1 v2 ... : kernel timings with ENVVAR set to each value
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
K=$1; shift
for v in "$@"; do
  echo "== $K=$v"
  env $K=$v timeout 300 python tools/profile_kernels.py 5 2>&1 | grep kernels
done | tee gpurun_out/knob.txt
