#!/bin/bash
# round 4, call B: the re-calibrated parity bars, the new GPU tests (loop rate of the unmodified train.py, strong scaling, perf guards), the loop-rate
# table, the HIP convergence spread under both product modes, and the stash variants timed IN SEQUENCE (results of variants are NOT valid)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
export NNR_PARITY_LOG=$PWD/gpurun_out/r04/b_parity_rel_l2.txt
rm -f $NNR_PARITY_LOG
NNR_FP64_YARDSTICK_REPORT_ONLY=1 timeout 1500 python -m pytest tests/test_gpu_split3.py tests/test_gpu_parity.py tests/test_gpu_camera.py tests/test_gpu_bench_shape_parity.py tests/test_gpu_perf_guard.py tests/test_gpu_loop_rate.py tests/test_gpu_bench_ranks.py tests/test_gpu_dropin.py tests/test_gpu_optim.py -q -m gpu -s 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -120 > gpurun_out/r04/b_gpu_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/b_gpu_tests.txt | tail -3; grep -n "^FAILED\|Error\|perf guard\|train.py (\|vs fp64\|strong scaling" gpurun_out/r04/b_gpu_tests.txt | head -40
unset NNR_PARITY_LOG
timeout 900 python tools/loop_rate.py --out gpurun_out/r04/b_scene_loop.json > gpurun_out/r04/b_scene_loop.log 2>&1; echo "loop_rate exit $?"; tail -6 gpurun_out/r04/b_scene_loop.log
timeout 600 python tools/conv_envelope_hip.py 2>&1 | grep -v "amdgpu.ids\|Loaded image\|test set\|train :" > gpurun_out/r04/b_conv_envelope_hip.txt; echo "envelope exit $?"; tail -3 gpurun_out/r04/b_conv_envelope_hip.txt
export NNR_FP32_PRODUCTS=split3
SHAPE="1024 192 f32" bash tools/gpu_variants.sh "$@" > gpurun_out/r04/b_stash_variants_in_sequence.txt 2>&1
cat gpurun_out/r04/b_stash_variants_in_sequence.txt
