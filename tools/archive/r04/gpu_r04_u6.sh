#!/bin/bash
# round 4, call U6: sweep of the narrow tiles' plan weights (NNR_WGRAD_W42 / _W41 / _W14, NNR_WGRAD_SPLIT_WEIGHT) around the timeline-derived values
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1 NNR_FP32_PRODUCTS=split3
{
for combo in "1035 1145 1145 440" "857 1051 986 440" "857 1051 986 400" "900 1110 1050 440" "900 1110 1050 420" "940 1145 1100 440" "940 1145 1100 420" "980 1145 1145 440" "1035 1145 1145 440" "900 1110 1050 400"; do
  set -- $combo
  echo -n "W42=$1 W41=$2 W14=$3 split=$4: "; NNR_WGRAD_W42=$1 NNR_WGRAD_W41=$2 NNR_WGRAD_W14=$3 NNR_WGRAD_SPLIT_WEIGHT=$4 timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1
done
} > gpurun_out/r04/u6_wgrad_plan_weight_sweep.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04/u6_wgrad_plan_weight_sweep.txt'):
    n, j = l.rsplit(': {"lib"', 1)
    try:
        d = json.loads('{"lib"' + j); print(n, 'isolated', d['ms']['mlp_wgrad'], 'in-sequence', d['in_sequence_ms']['mlp_wgrad'])
    except Exception as e:
        print(n, j[:300])
PY
