# usage: bash tools/gpu_timelines.sh v1 v2 ...   (libnnr_<v>.so built with -DNNR_TIMELINE [+ ablations])
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
for v in "$@"; do
  echo "== variant=$v"
  NNR_LIB=$R/nope-nerf_amd/nnr/libnnr_$v.so timeout 300 python tools/timeline.py 2>&1 | grep -E "nnr_timeline|rror|wgrad|slowest|class B|wave types"
done | tee gpurun_out/timelines.txt
