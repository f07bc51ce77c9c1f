# round 3, call G: bf16 kernels after the scalar-base addressing and LDS-parked positions; Adam bitwise test; timings
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layer_local.py tests/test_gpu_bench_shape_parity.py tests/test_gpu_determinism.py tests/test_gpu_perf_guard.py tests/test_gpu_optim.py -m gpu -q -k "bf16 or layer or determin or perf or adam or inference" > gpurun_out/r03/g_tests.txt 2>&1; echo "pytest exit $?"; tail -6 gpurun_out/r03/g_tests.txt
for i in 1 2; do timeout 200 python tools/time_kernels.py 4096 128 bf16 20 2>&1 | tail -1; done | tee gpurun_out/r03/g_bf16_time.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03/g_bench.json.txt 2> gpurun_out/r03/g_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/g_bench.json.txt').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['roofline']['kernels'].items()})
c = d['configs']['bf16_4096x128']
print('bf16', c['value'], c['ms_per_step'], c['kernels_ms'])
PY
