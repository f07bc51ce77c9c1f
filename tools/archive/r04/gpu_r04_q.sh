#!/bin/bash
# round 4, call Q: tile-major ACTIVATION planes (the forward's stash as whole-block non-temporal stores, the weight gradient reading both operands
# tile-major) -- the parity tests that read the planes, then the kernels in sequence and the bench line against the row-major-X library of the
# commit before (libnnr_rowx.so, same box)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_layer_local.py tests/test_gpu_split3.py tests/test_gpu_parity.py tests/test_gpu_bench_shape_parity.py tests/test_gpu_determinism.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -30 > gpurun_out/r04/q_parity_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/q_parity_tests.txt | tail -2; grep -n "^FAILED\|Error" gpurun_out/r04/q_parity_tests.txt | head
export NNR_FP32_PRODUCTS=split3
SHAPE="1024 192 f32" bash tools/gpu_variants.sh rowx f_nostash > gpurun_out/r04/q_tile_x_in_sequence.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04/q_tile_x_in_sequence.txt'):
    n, j = l.split(': ', 1)
    try:
        d = json.loads(j); print('%-10s isolated %s | in sequence %s' % (n, {k: round(v, 4) for k, v in d['ms'].items()}, {k: round(v, 4) for k, v in d['in_sequence_ms'].items()}))
    except Exception as e:
        print(n, j[:300])
PY
unset NNR_FP32_PRODUCTS
for v in product rowx; do
  if [ "$v" != product ]; then export NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_$v.so; else unset NNR_LIB; fi
  timeout 600 python bench.py --steps 200 --warmup 30 --no-extra --no-cpu-baseline > gpurun_out/r04/q_bench_$v.json.txt 2> gpurun_out/r04/q_bench_$v.err; echo "bench $v exit $?"
done
python - <<'PY'
import json
for v in ('product', 'rowx'):
    for l in open('gpurun_out/r04/q_bench_%s.json.txt' % v):
        if l.startswith('{'):
            d = json.loads(l)
            print(v, d['value'], d['ms_per_step'], {k: x['ms'] for k, x in d['roofline']['kernels'].items()}, 'frac', d['roofline']['frac'])
PY
