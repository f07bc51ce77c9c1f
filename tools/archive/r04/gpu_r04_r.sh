#!/bin/bash
# round 4, call R: plan-weight sweep of the weight gradient now that BOTH operands are tile-major (the optimum was 0.48 with row-major activations)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1 NNR_FP32_PRODUCTS=split3
{
for w in ${WEIGHTS:-440 480 520 560 600}; do
  echo -n "NNR_WGRAD_SPLIT_WEIGHT=$w: "; NNR_WGRAD_SPLIT_WEIGHT=$w timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1
done
} > gpurun_out/r04/r${TAG}_wgrad_weight_sweep_tile_x.txt 2>&1
python - "$TAG" <<'PY'
import json, sys
for l in open('gpurun_out/r04/r%s_wgrad_weight_sweep_tile_x.txt' % sys.argv[1]):
    n, j = l.rsplit(': {"lib"', 1)
    try:
        d = json.loads('{"lib"' + j); print(n, 'isolated', d['ms']['mlp_wgrad'], 'in-sequence', {k: d['in_sequence_ms'][k] for k in ('mlp_fwd','mlp_dgrad','mlp_wgrad')})
    except Exception as e:
        print(n, j[:300])
PY
