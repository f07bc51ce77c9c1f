mkdir -p gpurun_out/r2m
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k bf16 2>&1 | tail -2 > gpurun_out/r2m/tests.txt
timeout 200 python tools/time_kernels.py 4096 128 bf16 20 2>&1 | tail -1 > gpurun_out/r2m/time.txt
timeout 200 python tools/time_kernels.py 1024 192 bf16 20 2>&1 | tail -1 >> gpurun_out/r2m/time.txt
NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_timeline.so timeout 300 python tools/timeline.py 4096 128 bf16 2>&1 | tail -3 > gpurun_out/r2m/timeline.txt
cat gpurun_out/r2m/tests.txt gpurun_out/r2m/time.txt gpurun_out/r2m/timeline.txt
