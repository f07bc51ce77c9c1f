# round 6, GPU call g: the whole GPU suite (perf guards included) with the final fp16-term kernels, then smoke()
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
NNR_PARITY_LOG=gpurun_out/r06/g_parity_rel_l2.txt timeout 3000 python -m pytest tests -q -m gpu --maxfail=25 -s 2>&1 | grep -v "Warning\|warnings.warn\|amdgpu.ids\|^$" | tail -150 > gpurun_out/r06/g_gpu_tests.txt
tail -25 gpurun_out/r06/g_gpu_tests.txt
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/r06/g_smoke.txt
