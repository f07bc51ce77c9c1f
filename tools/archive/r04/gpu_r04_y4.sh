#!/bin/bash
# round 4, call Y4: depth_gather_bwd with four entries per LDS read (camera tests), the first-phase step by kernel, the remaining schedule variants
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_camera.py tests/test_aux_terms.py tests/test_gpu_dropin.py -q -m gpu 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -15 > gpurun_out/r04/y4_tests.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/y4_tests.txt | tail -2; grep -n "^FAILED\|Error" gpurun_out/r04/y4_tests.txt | head
R=$PWD
name=fp32_1024x192_aux; args="--aux --no-extra --no-cpu-baseline"; key="mlp_fwd_kernel<256, true"
mkdir -p gpurun_out/step_$name; rm -rf /tmp/step_$name
( cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/step_$name -o st -- python $R/bench.py $args --steps 40 --warmup 10 > $R/gpurun_out/step_$name/bench.txt 2>/dev/null )
t=$(find /tmp/step_$name -name "*kernel_trace.csv" | head -1)
python - "$t" "$key" "$name" "$R/gpurun_out/step_$name/bench.txt" > gpurun_out/r04/y4_${name}_step_kernel_breakdown.txt <<'PY'
import csv, sys, collections, json
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']][:50]
a, b = idx[-21], idx[-1]
seg = rows[a:b]
wall = int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
print('20 training steps, %s (rocprofv3 --kernel-trace; the tracer slows the HOST, so wall time here is not the step time):' % sys.argv[3])
print('GPU busy %.3f ms/step in %.1f launches/step (traced wall %.3f ms/step; untraced step time: the bench line)' % (busy / 20e6, len(seg) / 20, wall / 20e6))
per = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r['Kernel_Name'][:70]
    per[k][0] += 1; per[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
big = ('mlp_fwd', 'mlp_dgrad', 'wgrad_kernel', 'wgrad_b_kernel')
small = sum(t for k, (n, t) in per.items() if not any(x in k for x in big))
print('small launches (everything but the three main MLP kernels): %.1f us/step in %.1f launches' % (small / 20e3, sum(n for k, (n, t) in per.items() if not any(x in k for x in big)) / 20))
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%6.1f us/step  x%.1f  %s' % (t / 20e3, n / 20, k))
d = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
print('bench line of the traced run: %.3f ms/step' % d['ms_per_step'])
PY
head -12 gpurun_out/r04/y4_${name}_step_kernel_breakdown.txt | cut -c1-110
timeout 600 python bench.py --aux --steps 100 --warmup 20 --no-extra --no-cpu-baseline > gpurun_out/r04/y4_bench_aux.json.txt 2> gpurun_out/r04/y4_bench_aux.err; python -c "
import json
for l in open('gpurun_out/r04/y4_bench_aux.json.txt'):
    if l.startswith('{'):
        d = json.loads(l); print('aux bench', d['value'], d['ms_per_step'])"
if [ -n "$VARIANTS" ]; then
export NNR_FP32_PRODUCTS=split3
SHAPE="1024 192 f32" bash tools/gpu_variants.sh $VARIANTS product $VARIANTS > gpurun_out/r04/y4_schedule_variants_in_sequence.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04/y4_schedule_variants_in_sequence.txt'):
    n, j = l.split(': ', 1)
    try:
        d = json.loads(j); print('%-10s in sequence %s' % (n, {k: round(v, 4) for k, v in d['in_sequence_ms'].items() if 'mlp' in k}))
    except Exception as e:
        print(n, j[:300])
PY
fi
