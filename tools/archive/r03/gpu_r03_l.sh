#!/bin/bash
# round 3, call L: with_ssim and a learnable focal in the fused per-image block
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_aux_terms.py tests/test_eval_focal.py tests/test_gpu_dp.py tests/test_gpu_determinism.py tests/test_loss_switches.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r03/l_ssim_focal_tests.txt
echo "pytest exit $?"; grep -n "passed\|failed" gpurun_out/r03/l_ssim_focal_tests.txt
