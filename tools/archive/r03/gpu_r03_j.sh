# round 3, call J: the round-end profile set (kernel-trace stats of bench.py, PMC passes at both shapes) + step breakdowns
export PYTHONUNBUFFERED=1
bash tools/gpu_prof_round.sh > gpurun_out/prof_round.log 2>&1; echo "prof exit $?"
R=$PWD
for cfg in "fp32_1024x192|--no-extra --no-cpu-baseline|mlp_fwd_kernel<256, true" "bf16_4096x128|--bf16 --rays-per-gpu 4096 --samples 128 --no-extra --no-cpu-baseline|mlp_fwd_bf16_kernel<256, true"; do
  name=${cfg%%|*}; rest=${cfg#*|}; args=${rest%%|*}; key=${rest#*|}
  mkdir -p gpurun_out/step_$name; rm -rf /tmp/step_$name
  ( cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/step_$name -o st -- python $R/bench.py $args --steps 40 --warmup 10 > $R/gpurun_out/step_$name/bench.txt 2>/dev/null )
  t=$(find /tmp/step_$name -name "*kernel_trace.csv" | head -1)
  python - "$t" "$key" "$name" "$R/gpurun_out/step_$name/bench.txt" > gpurun_out/step_$name/breakdown.txt <<'PY'
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
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%6.1f us/step  x%.1f  %s' % (t / 20e3, n / 20, k))
d = json.loads(open(sys.argv[4]).read().strip().splitlines()[-1])
print('bench line of the traced run: %.3f ms/step' % d['ms_per_step'])
PY
  head -4 gpurun_out/step_$name/breakdown.txt
done
mkdir -p gpurun_out/r03; timeout 900 python bench.py > gpurun_out/r03/j_bench_untraced.json.txt 2> gpurun_out/r03/j_bench.err; echo "bench exit $?"
