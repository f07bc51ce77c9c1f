#!/bin/bash
# round 3, call V: balance of the weight-gradient plan between the three-term 4 x 4 tiles and the fp32 narrow tiles
mkdir -p gpurun_out/r03
for w in ${WEIGHTS:-380 440 500 560 620}; do echo -n "split weight $w: "; NNR_WGRAD_SPLIT_WEIGHT=$w timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms'])"; done > gpurun_out/r03/v_split_weight_sweep.txt 2>&1
cat gpurun_out/r03/v_split_weight_sweep.txt
timeout 600 python -m pytest tests/test_gpu_split3.py tests/test_gpu_layer_local.py tests/test_gpu_determinism.py -q -m gpu -x 2>&1 | tail -2
