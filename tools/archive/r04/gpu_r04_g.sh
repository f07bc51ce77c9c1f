#!/bin/bash
# round 4, call G: the weight gradient's class-A workgroups with the split shared by the four waves (wgrad_group_split): correctness (layer-local,
# parity, determinism, data-parallel identities), then timing against the private split (NNR_WGRAD_NO_COOP=1) with a plan-weight sweep
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_gpu_layer_local.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -15 > gpurun_out/r04/g_layer_local.txt
echo "layer-local exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/g_layer_local.txt | tail -2; grep -n "^FAILED\|Error\|assert" gpurun_out/r04/g_layer_local.txt | head -10
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py tests/test_gpu_dp.py tests/test_gpu_bench_shape_parity.py tests/test_gpu_split3.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -15 > gpurun_out/r04/g_gpu_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/g_gpu_tests.txt | tail -2; grep -n "^FAILED\|Error\|assert" gpurun_out/r04/g_gpu_tests.txt | head -10
export NNR_FP32_PRODUCTS=split3
{
echo -n "private split (NNR_WGRAD_NO_COOP=1), weight 520: "; NNR_WGRAD_NO_COOP=1 timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1
for w in 360 400 440 480 520; do
  echo -n "shared split, NNR_WGRAD_SPLIT_WEIGHT=$w: "; NNR_WGRAD_SPLIT_WEIGHT=$w timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1
done
} > gpurun_out/r04/g_wgrad_coop_sweep.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04/g_wgrad_coop_sweep.txt'):
    n, j = l.rsplit(': {"lib"', 1)
    try:
        d = json.loads('{"lib"' + j); print(n, 'isolated', d['ms']['mlp_wgrad'], 'in-sequence', {k: d['in_sequence_ms'][k] for k in ('mlp_fwd','mlp_dgrad','mlp_wgrad')})
    except Exception as e:
        print(n, j[:300])
PY
