#!/bin/bash
# round 4, call U5: the narrow tiles' plan weights from the per-wave timeline -- timeline of the product kernel with the new plan (NNR_TIMELINE
# library), kernel times against the old weights (NNR_WGRAD_W42=1035 NNR_WGRAD_W41=1145 NNR_WGRAD_W14=1145) and a small sweep; parity tests
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_layer_local.py tests/test_gpu_split3.py tests/test_gpu_bench_shape_parity.py tests/test_gpu_determinism.py -q -m gpu -x 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -8 > gpurun_out/r04/u5_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/u5_tests.txt | tail -2
export NNR_FP32_PRODUCTS=split3
for T in wtl $EXTRA_TL; do
NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_$T.so timeout 300 python tools/timeline.py 2>&1 | grep "wgrad waves\|slowest\|class B/A" > gpurun_out/r04/u5_wgrad_timeline_$T.txt
echo "== $T"; cut -c1-200 gpurun_out/r04/u5_wgrad_timeline_$T.txt
done
{
echo -n "old weights (1035 / 1145 / 1145): "; NNR_WGRAD_W42=1035 NNR_WGRAD_W41=1145 NNR_WGRAD_W14=1145 timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1
echo -n "new weights (857 / 1051 / 986): "; timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1
for sw in 400 420 460 480; do echo -n "new weights, NNR_WGRAD_SPLIT_WEIGHT=$sw: "; NNR_WGRAD_SPLIT_WEIGHT=$sw timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1; done
echo -n "old weights (1035 / 1145 / 1145): "; NNR_WGRAD_W42=1035 NNR_WGRAD_W41=1145 NNR_WGRAD_W14=1145 timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1
echo -n "new weights (857 / 1051 / 986): "; timeout 200 python tools/time_kernels.py 1024 192 f32 20 2>&1 | tail -1
} > gpurun_out/r04/u5_wgrad_plan_weights.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04/u5_wgrad_plan_weights.txt'):
    n, j = l.rsplit(': {"lib"', 1)
    try:
        d = json.loads('{"lib"' + j); print(n, 'isolated', d['ms']['mlp_wgrad'], 'in-sequence', d['in_sequence_ms']['mlp_wgrad'])
    except Exception as e:
        print(n, j[:300])
PY
