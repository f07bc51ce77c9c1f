#!/bin/bash
# round 4, call P: where the training forward's 0.3 ms over the inference forward go -- compile-time ablations of the D = 256 three-term training
# forward only (everything else is the product library), timed IN SEQUENCE (results of such libraries are NOT valid)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1 NNR_FP32_PRODUCTS=split3
SHAPE="1024 192 f32" bash tools/gpu_variants.sh f_nostash f_nomask f_nostash_nomask f_stashl2 f_nodma > gpurun_out/r04/p_forward_ablations_in_sequence.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04/p_forward_ablations_in_sequence.txt'):
    n, j = l.split(': ', 1)
    try:
        d = json.loads(j); print('%-18s forward isolated %.4f in sequence %.4f | inference forward %.4f | input gradient in sequence %.4f' % (n, d['ms']['mlp_fwd'], d['in_sequence_ms']['mlp_fwd'], d['ms']['mlp_fwd_infer'], d['in_sequence_ms']['mlp_dgrad']))
    except Exception as e:
        print(n, j[:200])
PY
