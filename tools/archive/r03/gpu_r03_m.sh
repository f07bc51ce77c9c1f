#!/bin/bash
# round 3, call M: first run of the three-term (six bf16 MFMA) fp32 mode: correctness vs fp64, then the headline step
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_split3.py -q -m gpu -x -s 2>&1 | tail -40 > gpurun_out/r03/m_split3_tests.txt
echo "pytest exit $?"; grep -n "passed\|failed\|three-term\|vs fp64\|Error\|assert" gpurun_out/r03/m_split3_tests.txt | head -20
NNR_FP32_PRODUCTS=split3 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03/m_bench_split3.json.txt 2> gpurun_out/r03/m_bench_split3.err
echo "bench exit $?"
python - <<'PY'
import json
for l in open('gpurun_out/r03/m_bench_split3.json.txt'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('kernels_ms') or d.get('kernel_ms') or {k:v for k,v in d.items() if 'kernel' in k})
PY
