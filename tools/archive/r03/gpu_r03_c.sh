# round 3, call C: whole GPU suite (no -x) with the one-launch Adam, the fused front end (512-thread backward) and the overwrite
# semantics; launches by Python line; headline bench
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r03/c_gpu_tests.txt 2>&1; echo "pytest exit $?"
tail -12 gpurun_out/r03/c_gpu_tests.txt
timeout 300 python tools/step_launches.py > gpurun_out/r03/c_step_launches_fp32.txt 2>&1; echo "launches exit $?"
timeout 600 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r03/c_bench_headline.json.txt 2> gpurun_out/r03/c_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/c_bench_headline.json.txt').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['roofline']['kernels'].items()})
PY
