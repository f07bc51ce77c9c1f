mkdir -p gpurun_out/r2e
export PYTHONUNBUFFERED=1
NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_tl.so timeout 300 python tools/timeline.py 4096 128 bf16 2>&1 | grep -E "nnr_timeline|rror|mlp_fwd" | tee gpurun_out/r2e/timeline_bf16.txt
