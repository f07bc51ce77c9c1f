#!/bin/bash
# round 4, call X: nearest-neighbour search with ONE source per lane and the packed arithmetic over pairs of destination points (NNR_PC_PER=1):
# sweep of the number of workgroups and of the points per trip (NNR_PC_GROUPS x 4) against the product (two sources per lane, 2048 workgroups)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
{
for S in 20736 32400; do for mode in smooth; do
  echo -n "two per lane, 2048 (product): "; NNR_PC_PER=2 NNR_PC_WGS=2048 timeout 120 python tools/time_pc_nearest.py $S $mode 2>&1 | tail -1
  for g in 1 2 4; do for w in 2048 4096 8192 16384; do echo -n "one per lane, NNR_PC_GROUPS=$g NNR_PC_WGS=$w: "; NNR_PC_PER=1 NNR_PC_GROUPS=$g NNR_PC_WGS=$w timeout 120 python tools/time_pc_nearest.py $S $mode 2>&1 | tail -1; done; done
done; done
} > gpurun_out/r04/x${TAG}_pc_nearest_one_per_lane.txt 2>&1
cat gpurun_out/r04/x${TAG}_pc_nearest_one_per_lane.txt | sed 's/nnr_pc_nearest //; s/ per call (fill + search + decode)//'
