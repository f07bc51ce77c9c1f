#!/bin/bash
# round 3, call T: gate words parked in LDS (input-gradient kernel, three-term mode): parity, determinism, timing
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_gpu_split3.py tests/test_gpu_parity.py tests/test_gpu_determinism.py tests/test_gpu_layer_local.py tests/test_gpu_bench_shape_parity.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r03/t_tests.txt
echo "pytest exit $?"; tail -3 gpurun_out/r03/t_tests.txt
SHAPE="1024 192 f32" bash tools/gpu_variants.sh > gpurun_out/r03/t_time.txt 2>&1; cat gpurun_out/r03/t_time.txt
timeout 600 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r03/t_bench.json.txt 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r03/t_bench.json.txt'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})
PY
