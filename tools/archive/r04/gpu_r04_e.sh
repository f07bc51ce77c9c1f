#!/bin/bash
# round 4, call E: A/B of the gradient-plane layouts with the weight gradient fetching two steps ahead -- product (tile-major, non-temporal), `tileplain`
# (tile-major, ordinary stores), `rowd` (row-major as in round 3); all three are VALID libraries.  Then what train_step returns, and the loop rate.
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_layer_local.py tests/test_gpu_parity.py tests/test_gpu_determinism.py tests/test_gpu_dp.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -15 > gpurun_out/r04/e_gpu_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/e_gpu_tests.txt | tail -3; grep -n "^FAILED\|Error\|assert" gpurun_out/r04/e_gpu_tests.txt | head -20
NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_rowd.so timeout 600 python -m pytest tests/test_gpu_layer_local.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep "passed\|failed" | tail -2
export NNR_FP32_PRODUCTS=split3
SHAPE="1024 192 f32" bash tools/gpu_variants.sh rowd tileplain > gpurun_out/r04/e_dplane_layouts.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04/e_dplane_layouts.txt'):
    n, j = l.split(': ', 1)
    try:
        d = json.loads(j); print(n, 'isolated', {k: d['ms'][k] for k in ('mlp_fwd','mlp_dgrad','mlp_wgrad')}, 'in-sequence', {k: d['in_sequence_ms'][k] for k in ('mlp_fwd','mlp_dgrad','mlp_wgrad')})
    except Exception as e:
        print(n, j[:300])
PY
for v in product rowd tileplain; do
  if [ "$v" != product ]; then export NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_$v.so; else unset NNR_LIB; fi
  timeout 600 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r04/e_bench_$v.json.txt 2> gpurun_out/r04/e_bench_$v.err
  python - $v <<'PY'
import json, sys
for l in open('gpurun_out/r04/e_bench_%s.json.txt' % sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l); print(sys.argv[1], d['value'], d['ms_per_step'], d['step_ms']['median'], {k: v['ms'] for k, v in d['roofline']['kernels'].items() if k.startswith('mlp')}, d['roofline']['frac'])
PY
done
unset NNR_LIB
timeout 300 python tools/loss_dict_shapes.py 2>&1 | grep -v "amdgpu.ids\|Warning" > gpurun_out/r04/e_loss_dict_shapes.txt; cut -c1-400 gpurun_out/r04/e_loss_dict_shapes.txt
timeout 900 python -m pytest tests/test_gpu_loop_rate.py -q -m gpu -s 2>&1 | grep "train.py (\|passed\|failed"
