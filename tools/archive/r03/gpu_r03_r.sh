#!/bin/bash
# round 3, call R: the whole GPU suite with the three-term products as the fp32 mode
mkdir -p gpurun_out/r03
export NNR_FP32_PRODUCTS=split3
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_conv_reference.py 2>&1 | tail -30 > gpurun_out/r03/r_gpu_suite_split3.txt
echo "pytest exit $?"; tail -12 gpurun_out/r03/r_gpu_suite_split3.txt
