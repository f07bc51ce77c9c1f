# round 3, call A: the new tests (multi-rank bench leg, end-to-end parity at the benchmark shapes), which Python line launches each of the
# step's small kernels, and a bench line with the reference-backed cpu_baseline
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_bench_shape_parity.py -x -q -s > gpurun_out/r03/a_new_tests.txt 2>&1; echo "pytest exit $?"
tail -15 gpurun_out/r03/a_new_tests.txt
timeout 300 python tools/step_launches.py > gpurun_out/r03/a_step_launches_fp32.txt 2>&1; echo "launches exit $?"
timeout 300 python tools/step_launches.py --bf16 4096 128 > gpurun_out/r03/a_step_launches_bf16.txt 2>&1
timeout 900 python bench.py > gpurun_out/r03/a_bench.json.txt 2> gpurun_out/r03/a_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/a_bench.json.txt').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['cpu_baseline'])[:600])
print({k: v.get('value') for k, v in d['configs'].items() if v})
PY
