#!/bin/bash
# round 4, call U: weight-gradient micro-steps of the shared split (packed subtracts forced, ...) against the library before (libnnr_w0.so), same box;
# layer-local + parity tests first (the product library only)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_layer_local.py tests/test_gpu_split3.py tests/test_gpu_bench_shape_parity.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -30 > gpurun_out/r04/u${TAG}_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/u${TAG}_tests.txt | tail -2; grep -n "^FAILED\|Error" gpurun_out/r04/u${TAG}_tests.txt | head
export NNR_FP32_PRODUCTS=split3
SHAPE="1024 192 f32" bash tools/gpu_variants.sh ${VARIANTS:-w0} product ${VARIANTS:-w0} > gpurun_out/r04/u${TAG}_wgrad_in_sequence.txt 2>&1
python - "$TAG" <<'PY'
import json, sys
for l in open('gpurun_out/r04/u%s_wgrad_in_sequence.txt' % sys.argv[1]):
    n, j = l.split(': ', 1)
    try:
        d = json.loads(j); print('%-10s isolated %s | in sequence %s' % (n, {k: round(v, 4) for k, v in d['ms'].items() if 'mlp' in k}, {k: round(v, 4) for k, v in d['in_sequence_ms'].items() if 'mlp' in k}))
    except Exception as e:
        print(n, j[:300])
PY
