#!/bin/bash
# round 4, call N: pruned (opt-in) against brute-force nearest-neighbour search on white-noise and on smooth depth maps; exactness; first-phase step
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_pc_pruned.py tests/test_pointcloud.py -q -m gpu 2>&1 | grep "passed\|failed"
{
for kind in noise smooth; do
  echo -n "brute force: "; timeout 120 python tools/time_pc_nearest.py 32400 $kind 2>&1 | tail -1
  for parts in 8 32; do echo -n "pruned, NNR_PC_PARTS=$parts: "; NNR_PC_PRUNED=1 NNR_PC_PARTS=$parts timeout 120 python tools/time_pc_nearest.py 32400 $kind 2>&1 | tail -1; done
done
} > gpurun_out/r04/n_pc_nearest_pruned_smooth.txt 2>&1
cat gpurun_out/r04/n_pc_nearest_pruned_smooth.txt
timeout 300 python bench.py --aux --no-extra --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('first-phase step', d['value'], d['ms_per_step'], d['step_ms']['median'])"
