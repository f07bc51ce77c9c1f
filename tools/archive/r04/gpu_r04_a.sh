#!/bin/bash
# round 4, call A: state of the tree at round start on this box -- the GPU suite (with the relative-L2 figures of every gradient tensor logged),
# smoke, the bench line, the fp64 bisect, and the stash-store timing variants (results of variants are NOT valid)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
export NNR_PARITY_LOG=$PWD/gpurun_out/r04/a_parity_rel_l2.txt
rm -f $NNR_PARITY_LOG
NNR_FP64_YARDSTICK_REPORT_ONLY=1 timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -150 > gpurun_out/r04/a_gpu_suite.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/a_gpu_suite.txt | tail -3; grep -n "^FAILED\|Error" gpurun_out/r04/a_gpu_suite.txt | head -20
unset NNR_PARITY_LOG
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/a_smoke.txt 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r04/a_smoke.txt
timeout 900 python bench.py > gpurun_out/r04/a_bench.json.txt 2> gpurun_out/r04/a_bench.err; echo "bench exit $?"
python - <<'PY'
import json
for l in open('gpurun_out/r04/a_bench.json.txt'):
    if l.startswith('{'):
        d = json.loads(l)
        print(d['value'], d['ms_per_step'], d.get('step_ms'), {k: v['ms'] for k, v in d['roofline']['kernels'].items()})
        for k, v in (d.get('configs') or {}).items():
            if v: print(' ', k, v.get('value'), v.get('ms_per_step'), v.get('kernels_ms'), v.get('thread_sweep_ms_per_step'))
        print(' cpu_baseline', {k: d['cpu_baseline'].get(k) for k in ('value', 'kind', 'cores')} if d.get('cpu_baseline') else None)
PY
timeout 600 python tools/fp64_bisect.py 256 256 64 333 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > gpurun_out/r04/a_fp64_bisect_d256.txt; echo "bisect exit $?"
timeout 600 python tools/fp64_bisect.py 128 256 64 205 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > gpurun_out/r04/a_fp64_bisect_d128.txt
head -40 gpurun_out/r04/a_fp64_bisect_d256.txt | cut -c1-140
export NNR_FP32_PRODUCTS=split3
SHAPE="1024 192 f32" bash tools/gpu_variants.sh "$@" > gpurun_out/r04/a_stash_variants.txt 2>&1
cat gpurun_out/r04/a_stash_variants.txt
