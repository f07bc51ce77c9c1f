# round 3, call D: the tests that failed in call C, the Adam arithmetic experiment, launches by Python line, bench
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_camera.py tests/test_gpu_sampling.py tests/test_gpu_determinism.py tests/test_gpu_dp.py tests/test_train_steps.py -m gpu -q > gpurun_out/r03/d_gpu_tests.txt 2>&1; echo "pytest exit $?"
tail -6 gpurun_out/r03/d_gpu_tests.txt
timeout 600 python tools/adam_variants.py > gpurun_out/r03/d_adam_variants.txt 2>&1; echo "adam exit $?"; cat gpurun_out/r03/d_adam_variants.txt | tail -40
timeout 300 python tools/step_launches.py > gpurun_out/r03/d_step_launches_fp32.txt 2>&1
timeout 600 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r03/d_bench_headline.json.txt 2> gpurun_out/r03/d_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/d_bench_headline.json.txt').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['roofline']['kernels'].items()})
PY
