# round 6, GPU call i: fp16-term weight gradient with the residual at 2^11: layer-local + parity + determinism tests, plan-weight sweep, short bench
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
O=gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py tests/test_gpu_layer_local.py -q -m gpu --maxfail=5 2>&1 | tail -12
for w in 280 300 320 340 360; do echo "== split2 weight $w"; NNR_WGRAD_SPLIT2_WEIGHT=$w timeout 300 python tools/time_kernels.py 1024 192 f32 5 2>&1 | tail -1 | sed 's/.*in_sequence_ms/seq/'; done | tee $O/i_wgrad_f16_weight_sweep.txt
timeout 600 python bench.py --no-extra --no-cpu-baseline > $O/i_bench.txt 2>&1; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06/i_bench.txt') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('bench', d['value'], 'rays/s', d['ms_per_step'], 'ms', {k:v['ms'] for k,v in d['roofline']['kernels'].items()})
else: print(open('gpurun_out/r06/i_bench.txt').read()[-1500:])
PY
