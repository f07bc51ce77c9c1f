# round 3, call B: full GPU suite after the determinism rewrites, the overwrite semantics of the weight-gradient stage and the fused
# front end; step breakdown of the fp32 headline step; launches by Python line
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03/b_gpu_tests.txt 2>&1; echo "pytest exit $?"
tail -8 gpurun_out/r03/b_gpu_tests.txt
timeout 300 python tools/step_launches.py > gpurun_out/r03/b_step_launches_fp32.txt 2>&1; echo "launches exit $?"
timeout 600 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r03/b_bench_headline.json.txt 2> gpurun_out/r03/b_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/b_bench_headline.json.txt').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernels'])
PY
