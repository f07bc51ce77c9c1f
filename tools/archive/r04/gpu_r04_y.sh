#!/bin/bash
# round 4, call Y: the one-source-per-lane nearest-neighbour search as the default -- parity tests of the per-image block (reference goldens incl.
# duplicated points, determinism), the search alone, the first-phase bench line against the two-per-lane kernel (NNR_PC_PER=2)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_pointcloud.py tests/test_aux_terms.py tests/test_gpu_determinism.py tests/test_gpu_perf_guard.py -q -m gpu 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -15 > gpurun_out/r04/y_pc_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/y_pc_tests.txt | tail -2; grep -n "^FAILED\|Error" gpurun_out/r04/y_pc_tests.txt | head
for S in 20736 32400 1000 77; do timeout 120 python tools/time_pc_nearest.py $S smooth 2>&1 | tail -1; NNR_PC_PER=2 timeout 120 python tools/time_pc_nearest.py $S smooth 2>&1 | tail -1; done > gpurun_out/r04/y_pc_nearest_default.txt 2>&1
cat gpurun_out/r04/y_pc_nearest_default.txt
for v in one two; do
  if [ "$v" = two ]; then export NNR_PC_PER=2; else unset NNR_PC_PER; fi
  timeout 600 python bench.py --aux --steps 100 --warmup 20 --no-extra --no-cpu-baseline > gpurun_out/r04/y_bench_aux_$v.json.txt 2> gpurun_out/r04/y_bench_aux_$v.err; echo "bench $v exit $?"
done
python - <<'PY'
import json
for v in ('one', 'two'):
    for l in open('gpurun_out/r04/y_bench_aux_%s.json.txt' % v):
        if l.startswith('{'):
            d = json.loads(l); print(v, d['value'], d['ms_per_step'])
PY
