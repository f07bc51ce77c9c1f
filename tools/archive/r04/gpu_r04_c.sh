#!/bin/bash
# round 4, call C: the tile-major gradient planes of the three-term mode (dgrad writes, wgrad reads) -- layer-local / parity / determinism tests first,
# then timing (in sequence), the plan-weight sweep of the weight gradient, the loop rate with the early host copies of the logged scalars
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_layer_local.py tests/test_gpu_parity.py tests/test_gpu_split3.py tests/test_gpu_determinism.py tests/test_gpu_dp.py tests/test_gpu_bench_shape_parity.py tests/test_train_steps.py tests/test_gpu_loop_rate.py -q -m gpu -s -x 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -60 > gpurun_out/r04/c_gpu_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/c_gpu_tests.txt | tail -3; grep -n "^FAILED\|Error\|train.py (\|vs fp64\|assert" gpurun_out/r04/c_gpu_tests.txt | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/c_smoke.txt 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r04/c_smoke.txt
export NNR_FP32_PRODUCTS=split3
{
for w in 440 520 600 680; do
  echo -n "NNR_WGRAD_SPLIT_WEIGHT=$w: "; NNR_WGRAD_SPLIT_WEIGHT=$w timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1
done
} > gpurun_out/r04/c_wgrad_weight_sweep.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04/c_wgrad_weight_sweep.txt'):
    n, j = l.split(': ', 1)
    try:
        d = json.loads(j); print(n, 'isolated', {k: d['ms'][k] for k in ('mlp_fwd','mlp_dgrad','mlp_wgrad')}, 'in-sequence', {k: d['in_sequence_ms'][k] for k in ('mlp_fwd','mlp_dgrad','mlp_wgrad')})
    except Exception as e:
        print(n, j[:300])
PY
timeout 600 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r04/c_bench_headline.json.txt 2> gpurun_out/r04/c_bench.err; echo "bench exit $?"
python - <<'PY'
import json
for l in open('gpurun_out/r04/c_bench_headline.json.txt'):
    if l.startswith('{'):
        d = json.loads(l); print(d['value'], d['ms_per_step'], d.get('step_ms'), {k: v['ms'] for k, v in d['roofline']['kernels'].items()}, d['roofline']['frac'])
PY
