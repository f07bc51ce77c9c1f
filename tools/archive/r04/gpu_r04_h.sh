mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_layer_local.py tests/test_gpu_parity.py tests/test_gpu_determinism.py tests/test_gpu_dp.py tests/test_gpu_bench_shape_parity.py tests/test_train_steps.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -15 > gpurun_out/r04/h_gpu_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/h_gpu_tests.txt | tail -2; grep -n "^FAILED\|Error\|assert" gpurun_out/r04/h_gpu_tests.txt | head -10
export NNR_FP32_PRODUCTS=split3
timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1 > gpurun_out/r04/h_time_kernels.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04/h_time_kernels.txt').read())
print('isolated', {k: d['ms'][k] for k in ('mlp_fwd','mlp_dgrad','mlp_wgrad','mlp_fwd_infer')}, 'in-sequence', {k: d['in_sequence_ms'][k] for k in ('mlp_fwd','mlp_dgrad','mlp_wgrad')})
PY
timeout 600 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r04/h_bench_headline.json.txt 2> gpurun_out/r04/h_bench.err; echo "bench exit $?"
python - <<'PY'
import json
for l in open('gpurun_out/r04/h_bench_headline.json.txt'):
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d['step_ms']['median'], {k: v['ms'] for k, v in d['roofline']['kernels'].items() if k.startswith('mlp')}, d['roofline']['frac'], d['roofline']['kernel'])
PY
