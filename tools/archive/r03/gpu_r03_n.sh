#!/bin/bash
# round 3, call N: the three-term mode against the fp32-MFMA mode (with experiment libraries: fewer terms), the fp64 test, the headline step
mkdir -p gpurun_out/r03
tools/ubench/build/mfma_rounding > gpurun_out/r03/n_mfma_rounding.txt 2>&1
{
for lib in "" terms3 terms1; do
  for cfg in "256 64 256" "1024 192 256"; do
    if [ -n "$lib" ]; then export NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_$lib.so; else unset NNR_LIB; fi
    timeout 300 python tools/split3_debug.py $cfg 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|torch.cuda.synchronize" | tail -5
  done
done
} > gpurun_out/r03/n_split3_debug.txt 2>&1
unset NNR_LIB
cat gpurun_out/r03/n_split3_debug.txt
timeout 900 python -m pytest tests/test_gpu_split3.py -q -m gpu -s 2>&1 | tail -40 > gpurun_out/r03/n_split3_tests.txt
echo "pytest exit $?"; grep -n "passed\|failed\|three-term\|vs fp64\|Error\|assert" gpurun_out/r03/n_split3_tests.txt | head -20
NNR_FP32_PRODUCTS=split3 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03/n_bench_split3.json.txt 2> gpurun_out/r03/n_bench_split3.err
echo "bench exit $?"
python - <<'PY'
import json
for l in open('gpurun_out/r03/n_bench_split3.json.txt'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], {k:v for k,v in d.items() if 'kernel' in k})
PY
