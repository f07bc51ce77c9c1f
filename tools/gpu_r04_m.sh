#!/bin/bash
# round 4, call M: the pruned exact nearest-neighbour search against the brute-force one: exactness tests, timing, the first-phase step
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_pc_pruned.py tests/test_pointcloud.py tests/test_aux_terms.py tests/test_gpu_dp.py -q -m gpu 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -12
{
echo -n "brute force: "; NNR_PC_BRUTE=1 timeout 120 python tools/time_pc_nearest.py 32400 2>&1 | tail -1
for parts in 1 2 4 8 16 32; do echo -n "pruned, NNR_PC_PARTS=$parts: "; NNR_PC_PARTS=$parts timeout 120 python tools/time_pc_nearest.py 32400 2>&1 | tail -1; done
echo -n "pruned, 16 128 points: "; timeout 120 python tools/time_pc_nearest.py 16128 2>&1 | tail -1
echo -n "brute, 16 128 points: "; NNR_PC_BRUTE=1 timeout 120 python tools/time_pc_nearest.py 16128 2>&1 | tail -1
} > gpurun_out/r04/m_pc_nearest_pruned.txt 2>&1
cat gpurun_out/r04/m_pc_nearest_pruned.txt
timeout 300 python bench.py --aux --no-extra --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('first-phase step', d['value'], d['ms_per_step'], d['step_ms']['median'])"
