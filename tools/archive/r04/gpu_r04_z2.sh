#!/bin/bash
# round 4, last call: the whole GPU suite, smoke and the untraced bench line on the final binaries (after the z_* profile set: early scalars in one
# copy, nearest-neighbour search with four points per trip)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -30 > gpurun_out/r04/z2_gpu_suite.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/z2_gpu_suite.txt | tail -2; grep -n "^FAILED" gpurun_out/r04/z2_gpu_suite.txt | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/z2_smoke.txt 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r04/z2_smoke.txt
timeout 900 python bench.py > gpurun_out/r04/z2_final_bench.json.txt 2> gpurun_out/r04/z2_bench.err; echo "bench exit $?"
python - <<'PY'
import json
for l in open('gpurun_out/r04/z2_final_bench.json.txt'):
    if l.startswith('{'):
        d = json.loads(l)
        print(d['value'], d['ms_per_step'], d.get('step_ms'), {k: v['ms'] for k, v in d['roofline']['kernels'].items()}, 'frac', d['roofline']['frac'], d['roofline']['kernel'], 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'])
        for k, v in (d.get('configs') or {}).items():
            if v: print(' ', k, v.get('value'), v.get('ms_per_step'), (v.get('roofline') or {}).get('frac'))
PY
