# round 3, call K: reduce / un-merge kernels reworked -- weight-gradient parity (layer-local, golden, determinism, shard additivity), then timing
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layer_local.py tests/test_gpu_determinism.py tests/test_train_steps.py tests/test_gpu_bench_shape_parity.py -m gpu -q > gpurun_out/r03/k_tests.txt 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r03/k_tests.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03/k_bench.json.txt 2> gpurun_out/r03/k_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/k_bench.json.txt').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['roofline']['kernels'].items()})
c = d['configs']['bf16_4096x128']
print('bf16', c['value'], c['ms_per_step'], c['kernels_ms'])
PY
cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o k -- python $OLDPWD/bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1; cd $OLDPWD
f=$(find /tmp/kst -name "*kernel_stats.csv" | head -1); grep -E "reduce|unmerge|merge_kernel|step_rays" $f | cut -c1-160
