mkdir -p gpurun_out/r2k
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16" 2>&1 | tail -25 > gpurun_out/r2k/tests_bf16.txt
timeout 200 python tools/time_kernels.py 4096 128 bf16 20 2>&1 | tail -2 > gpurun_out/r2k/time.txt
timeout 200 python tools/time_kernels.py 1024 192 bf16 20 2>&1 | tail -1 >> gpurun_out/r2k/time.txt
cat gpurun_out/r2k/tests_bf16.txt gpurun_out/r2k/time.txt
