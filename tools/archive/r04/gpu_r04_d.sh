#!/bin/bash
# round 4, call D: the gradient planes' stores as true global non-temporal stores (scalar base): correctness, timing in sequence, bench headline;
# host cost of a training step with and without the per-image block (where the unmodified train.py's loop loses its 25 % in the first phase)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_layer_local.py tests/test_gpu_parity.py tests/test_gpu_determinism.py tests/test_gpu_dp.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -15 > gpurun_out/r04/d_gpu_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/d_gpu_tests.txt | tail -3; grep -n "^FAILED\|Error\|assert" gpurun_out/r04/d_gpu_tests.txt | head -20
export NNR_FP32_PRODUCTS=split3
timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1 > gpurun_out/r04/d_time_kernels.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04/d_time_kernels.txt').read())
print('isolated', {k: d['ms'][k] for k in ('mlp_fwd','mlp_dgrad','mlp_wgrad')}, 'in-sequence', {k: d['in_sequence_ms'][k] for k in ('mlp_fwd','mlp_dgrad','mlp_wgrad')})
PY
timeout 600 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r04/d_bench_headline.json.txt 2> gpurun_out/r04/d_bench.err; echo "bench exit $?"
python - <<'PY'
import json
for l in open('gpurun_out/r04/d_bench_headline.json.txt'):
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d.get('step_ms'), {k: v['ms'] for k, v in d['roofline']['kernels'].items()}, d['roofline']['frac'])
PY
timeout 300 python tools/cpu_step_cost.py > gpurun_out/r04/d_host_step_cost.txt 2>&1; head -3 gpurun_out/r04/d_host_step_cost.txt | grep "host time"
timeout 300 python tools/cpu_step_cost.py --aux > gpurun_out/r04/d_host_step_cost_aux.txt 2>&1; grep "host time" gpurun_out/r04/d_host_step_cost_aux.txt; grep -A34 "cumulative" gpurun_out/r04/d_host_step_cost_aux.txt | cut -c1-170 | head -40
