#!/bin/bash
# round 4, round-end call: the whole GPU suite (measured parity figures logged), smoke, the round-end profile set (bench under --kernel-trace --stats,
# PMC passes FETCH_SIZE / WRITE_SIZE / SQ at both shapes), step breakdowns of the headline, first-phase and bf16 steps, the untraced bench line
T=${1:-z}
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
export NNR_PARITY_LOG=$PWD/gpurun_out/r04/${T}_parity_rel_l2.txt
rm -f $NNR_PARITY_LOG
timeout 2400 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "amdgpu.ids\|UserWarning\|Consider using\|warnings.warn" | tail -120 > gpurun_out/r04/${T}_gpu_suite.txt
echo "pytest exit ${PIPESTATUS[0]}"; grep "passed\|failed" gpurun_out/r04/${T}_gpu_suite.txt | tail -2; grep -n "^FAILED" gpurun_out/r04/${T}_gpu_suite.txt | head
unset NNR_PARITY_LOG
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04/${T}_smoke.txt 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r04/${T}_smoke.txt
bash tools/gpu_prof_round.sh > gpurun_out/prof_round.log 2>&1; echo "prof exit $?"
R=$PWD
for cfg in "fp32_1024x192|--no-extra --no-cpu-baseline|mlp_fwd_kernel<256, true" "fp32_1024x192_aux|--aux --no-extra --no-cpu-baseline|mlp_fwd_kernel<256, true" "bf16_4096x128|--bf16 --rays-per-gpu 4096 --samples 128 --no-extra --no-cpu-baseline|mlp_fwd_bf16_kernel<256, true"; do
  name=${cfg%%|*}; rest=${cfg#*|}; args=${rest%%|*}; key=${rest#*|}
  mkdir -p gpurun_out/step_$name; rm -rf /tmp/step_$name
  ( cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/step_$name -o st -- python $R/bench.py $args --steps 40 --warmup 10 > $R/gpurun_out/step_$name/bench.txt 2>/dev/null )
  t=$(find /tmp/step_$name -name "*kernel_trace.csv" | head -1)
  python - "$t" "$key" "$name" "$R/gpurun_out/step_$name/bench.txt" > gpurun_out/r04/${T}_${name}_step_kernel_breakdown.txt <<'PY'
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
  head -3 gpurun_out/r04/${T}_${name}_step_kernel_breakdown.txt
done
timeout 900 python bench.py > gpurun_out/r04/${T}_round_end_bench_untraced.json.txt 2> gpurun_out/r04/${T}_bench.err; echo "bench exit $?"
python - $T <<'PY'
import json, sys
for l in open('gpurun_out/r04/%s_round_end_bench_untraced.json.txt' % sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print(d['value'], d['ms_per_step'], d.get('step_ms'), {k: v['ms'] for k, v in d['roofline']['kernels'].items()}, 'frac', d['roofline']['frac'], d['roofline']['kernel'])
        for k, v in (d.get('configs') or {}).items():
            if v: print(' ', k, v.get('value'), v.get('ms_per_step'), v.get('kernels_ms'), (v.get('roofline') or {}).get('frac'))
PY
