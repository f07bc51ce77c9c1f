# usage: bash tools/gpu_pass.sh <tag> [tests] [bench] [bf16] [fp32] [tl]      -- results under gpurun_out/<tag>/
tag=$1; shift
mkdir -p gpurun_out/$tag
export PYTHONUNBUFFERED=1
for what in "$@"; do
case $what in
tests)  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/$tag/pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/$tag/pytest.txt
        grep -E "passed|failed|^FAILED|^ERROR|bf16 |fp32:|normal term|exit" gpurun_out/$tag/pytest.txt | tail -25 ;;
bench)  timeout 900 python bench.py > gpurun_out/$tag/bench.txt 2> gpurun_out/$tag/bench.err; echo "bench exit $?"; tail -c 4500 gpurun_out/$tag/bench.txt ;;
bf16)   timeout 200 python tools/time_kernels.py 4096 128 bf16 5 2>/dev/null | tail -1 | tee gpurun_out/$tag/time_bf16.txt ;;
fp32)   timeout 200 python tools/time_kernels.py 1024 192 fp32 5 2>/dev/null | tail -1 | tee gpurun_out/$tag/time_fp32.txt ;;
tl)     NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_tl.so timeout 300 python tools/timeline.py 4096 128 bf16 2>&1 | grep -E "nnr_timeline|rror" | tee gpurun_out/$tag/timeline_bf16.txt
        NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_tl.so timeout 300 python tools/timeline.py 1024 192 2>&1 | grep -E "nnr_timeline_fwd|nnr_timeline_dgrad|rror" | tee gpurun_out/$tag/timeline_fp32.txt ;;
esac
done
