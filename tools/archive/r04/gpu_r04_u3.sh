#!/bin/bash
# round 4, call U3: what the weight gradient would cost if its operands arrived as terms -- ablation builds of the shared split (results NOT valid):
# wcoop1 = no split arithmetic (fetches, exchange writes and the barrier stay), wcoop2 = no fetches and no exchange writes either
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1 NNR_FP32_PRODUCTS=split3
SHAPE="1024 192 f32" bash tools/gpu_variants.sh wcoop1 wcoop2 product wcoop1 wcoop2 > gpurun_out/r04/u3_wgrad_ablations_in_sequence.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04/u3_wgrad_ablations_in_sequence.txt'):
    n, j = l.split(': ', 1)
    try:
        d = json.loads(j); print('%-10s weight gradient isolated %.4f in sequence %.4f' % (n, d['ms']['mlp_wgrad'], d['in_sequence_ms']['mlp_wgrad']))
    except Exception as e:
        print(n, j[:300])
PY
