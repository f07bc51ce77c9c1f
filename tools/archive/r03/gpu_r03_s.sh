#!/bin/bash
# round 3, call S: the whole GPU suite, smoke and the default bench with the three-term products as the fp32 default
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/r03/s_gpu_suite.txt
echo "pytest exit $?"; tail -6 gpurun_out/r03/s_gpu_suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/s_smoke.txt 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/r03/s_smoke.txt
timeout 900 python bench.py > gpurun_out/r03/s_bench.json.txt 2> gpurun_out/r03/s_bench.err; echo "bench exit $?"
python - <<'PY'
import json
for l in open('gpurun_out/r03/s_bench.json.txt'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})
        print({k:(v and (v.get('value'), v.get('ms_per_step'))) for k,v in d['configs'].items()}); print(d['cpu_baseline'])
PY
