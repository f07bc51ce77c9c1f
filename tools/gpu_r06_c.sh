# round 6, GPU call c: the whole GPU suite with the two-term fp16 products as the default of the fp32 mode
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
NNR_PARITY_LOG=gpurun_out/r06/c_parity_rel_l2.txt timeout 3000 python -m pytest tests -q -m gpu --maxfail=25 --deselect tests/test_gpu_perf_guard.py 2>&1 | tail -60 > gpurun_out/r06/c_gpu_tests.txt
tail -40 gpurun_out/r06/c_gpu_tests.txt
