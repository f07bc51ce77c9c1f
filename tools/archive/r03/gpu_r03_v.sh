#!/bin/bash
# round 3, call V: balance of the weight-gradient plan between the three-term tiles (4 x 4, 4 x 2) and the fp32 narrow tiles
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_split3.py tests/test_gpu_layer_local.py tests/test_gpu_determinism.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -2
for w in ${WEIGHTS:-"520 650" "520 500" "520 800" "460 650" "580 650"}; do set -- $w; echo -n "split weights $1 $2: "; NNR_WGRAD_SPLIT_WEIGHT=$1 NNR_WGRAD_SPLIT_WEIGHT2=$2 timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms']['mlp_wgrad'])"; done > gpurun_out/r03/v_split_weight_sweep2.txt 2>&1
cat gpurun_out/r03/v_split_weight_sweep2.txt
