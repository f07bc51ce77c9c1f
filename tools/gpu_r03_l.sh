#!/bin/bash
# round 3, call L: with_ssim in the fused per-image block
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_aux_terms.py tests/test_gpu_dp.py tests/test_gpu_determinism.py -q -m gpu -x 2>&1 | tail -25 > gpurun_out/r03/l_ssim_tests.txt
echo "pytest exit $?"; tail -5 gpurun_out/r03/l_ssim_tests.txt
