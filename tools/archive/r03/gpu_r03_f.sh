# round 3, call F: bf16 kernels after the LDS-staged gates, recomputed chain-rule factors and the batched pass start: parity first,
# then timings; Adam arithmetic diagnostics (third round)
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layer_local.py tests/test_gpu_bench_shape_parity.py tests/test_gpu_determinism.py tests/test_gpu_perf_guard.py -m gpu -q -k "bf16 or layer or determin or perf" > gpurun_out/r03/f_bf16_tests.txt 2>&1; echo "pytest exit $?"; tail -12 gpurun_out/r03/f_bf16_tests.txt
for i in 1 2; do timeout 200 python tools/time_kernels.py 4096 128 bf16 20 2>&1 | tail -1; done | tee gpurun_out/r03/f_bf16_time.txt
timeout 600 python tools/adam_variants.py 2>&1 | grep -v Warn > gpurun_out/r03/f_adam_variants.txt; head -40 gpurun_out/r03/f_adam_variants.txt
