#!/bin/bash
# round 4, call K: the nearest-neighbour search of the per-image block -- sources per lane x number of destination ranges
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r04
{
for per in 4 2; do for wgs in 32 64 128 256 512 1024 2048; do
  echo -n "NNR_PC_PER=$per NNR_PC_WGS=$wgs: "; NNR_PC_PER=$per NNR_PC_WGS=$wgs timeout 120 python tools/time_pc_nearest.py 32400 2>&1 | tail -1
done; done
} > gpurun_out/r04/k_pc_nearest_sweep.txt 2>&1
cat gpurun_out/r04/k_pc_nearest_sweep.txt
