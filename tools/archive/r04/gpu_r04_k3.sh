#!/bin/bash
# round 4, call K3: nearest-neighbour search with four destination points per trip -- bit-exactness tests, then the knob sweep again
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r04
timeout 300 python -m pytest tests/test_pointcloud.py tests/test_aux_terms.py tests/test_gpu_dp.py -q -m gpu 2>&1 | grep "passed\|failed"
{
for per in 4 2; do for wgs in 512 1024 2048 4096; do
  echo -n "NNR_PC_PER=$per NNR_PC_WGS=$wgs: "; NNR_PC_PER=$per NNR_PC_WGS=$wgs timeout 120 python tools/time_pc_nearest.py 32400 2>&1 | tail -1
done; done
} > gpurun_out/r04/k3_pc_nearest_four_points_per_trip.txt 2>&1
cat gpurun_out/r04/k3_pc_nearest_four_points_per_trip.txt
