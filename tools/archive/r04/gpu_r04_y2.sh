#!/bin/bash
# round 4, call Y2: the nearest-neighbour search with seeded ranges (NNR_PC_SEED 0 / 1 / 2) -- parity tests of the per-image block, the search alone
# on white-noise and smooth depth, the changed timing tests (perf guard, loop rate)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_pointcloud.py tests/test_aux_terms.py tests/test_gpu_determinism.py tests/test_gpu_dp_procs.py -q -m gpu 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -15 > gpurun_out/r04/y2_pc_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/y2_pc_tests.txt | tail -2; grep -n "^FAILED\|Error" gpurun_out/r04/y2_pc_tests.txt | head
for S in 20736 32400; do for mode in white smooth; do for seed in 0 1 2; do
  echo -n "NNR_PC_SEED=$seed: "; NNR_PC_SEED=$seed timeout 120 python tools/time_pc_nearest.py $S $mode 2>&1 | tail -1
done; done; done > gpurun_out/r04/y2_pc_nearest_seeds.txt 2>&1
sed 's/nnr_pc_nearest //; s/ per call (fill + search + decode)//' gpurun_out/r04/y2_pc_nearest_seeds.txt
timeout 900 python -m pytest tests/test_gpu_perf_guard.py tests/test_gpu_loop_rate.py -q -m gpu -s 2>&1 | grep "perf guard\|train.py (\|passed\|failed" > gpurun_out/r04/y2_timing_tests.txt; cat gpurun_out/r04/y2_timing_tests.txt | cut -c1-220
