#!/bin/bash
# round 3, call P: the three-term kernels: correctness, then timing (product + experiment libraries)
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_split3.py -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r03/p_split3_tests.txt
echo "pytest exit $?"; grep -n "passed\|failed\|three-term\|vs fp64\|Error\|assert" gpurun_out/r03/p_split3_tests.txt | head -20
export NNR_FP32_PRODUCTS=split3
SHAPE="1024 192 f32" bash tools/gpu_variants.sh "$@" > gpurun_out/r03/p_split3_time.txt 2>&1
cat gpurun_out/r03/p_split3_time.txt
