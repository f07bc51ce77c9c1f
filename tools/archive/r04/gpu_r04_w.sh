#!/bin/bash
# round 4, call W: nearest-neighbour search with the destination points through the scalar cache (NNR_PC_SCALAR=1; at the time of the run the default, with NNR_PC_LDS=1 for the LDS tile) against the LDS tile:
# parity tests of the per-image block, then the search alone at the trainer's cloud sizes, then the first-phase bench line both ways
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_pointcloud.py tests/test_aux_terms.py tests/test_gpu_aux_step.py -q -m gpu 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -15 > gpurun_out/r04/w_pc_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/w_pc_tests.txt | tail -2; grep -n "^FAILED\|Error" gpurun_out/r04/w_pc_tests.txt | head
{
for S in 20736 32400; do for mode in white smooth; do
  echo -n "scalar cache: "; timeout 120 python tools/time_pc_nearest.py $S $mode 2>&1 | tail -1
  echo -n "LDS tile:     "; NNR_PC_LDS=1 timeout 120 python tools/time_pc_nearest.py $S $mode 2>&1 | tail -1
  for w in 512 1024 4096; do echo -n "scalar cache, NNR_PC_WGS=$w: "; NNR_PC_WGS=$w timeout 120 python tools/time_pc_nearest.py $S $mode 2>&1 | tail -1; done
  echo -n "scalar cache, NNR_PC_PER=4 NNR_PC_WGS=1024: "; NNR_PC_PER=4 NNR_PC_WGS=1024 timeout 120 python tools/time_pc_nearest.py $S $mode 2>&1 | tail -1
done; done
} > gpurun_out/r04/w_pc_nearest_scalar_vs_lds.txt 2>&1
cat gpurun_out/r04/w_pc_nearest_scalar_vs_lds.txt
for v in scalar lds; do
  if [ "$v" = lds ]; then export NNR_PC_LDS=1; else unset NNR_PC_LDS; fi
  timeout 600 python bench.py --aux --steps 100 --warmup 20 --no-extra --no-cpu-baseline > gpurun_out/r04/w_bench_aux_$v.json.txt 2> gpurun_out/r04/w_bench_aux_$v.err; echo "bench $v exit $?"
done
python - <<'PY'
import json
for v in ('scalar', 'lds'):
    for l in open('gpurun_out/r04/w_bench_aux_%s.json.txt' % v):
        if l.startswith('{'):
            d = json.loads(l); print(v, d['value'], d['ms_per_step'])
PY
