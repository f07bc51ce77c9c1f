# round 6, GPU call j: fp16-term weight gradient as the default: the tests that see it, isolated kernel times, a short bench with the clock probe
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
O=gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_layer_local.py tests/test_gpu_split3.py tests/test_gpu_golden.py tests/test_gpu_perf_guard.py -q -m gpu --maxfail=5 2>&1 | tail -12
timeout 300 python tools/time_kernels.py 1024 192 f32 5 2>&1 | tail -2 | tee $O/j_time_kernels.txt
timeout 600 python bench.py --no-extra --no-cpu-baseline > $O/j_bench.txt 2>&1; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06/j_bench.txt') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); r=d['roofline']; print('bench', d['value'], 'rays/s', d['ms_per_step'], 'ms', {k:v['ms'] for k,v in r['kernels'].items()}); print(r.get('clock'), r.get('peak_at_measured_clock'), r.get('frac'), r.get('frac_at_measured_clock'), d.get('box'))
else: print(open('gpurun_out/r06/j_bench.txt').read()[-1500:])
PY
