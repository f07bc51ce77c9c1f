# Per-step GPU time of a training step by kernel (rocprofv3 --kernel-trace over bench.py, bf16 4096 x 128): is the GPU ever idle, and what do
# the small launches cost?  -> profiles/r02/l_bf16_4096x128_step_kernel_breakdown.txt (gpu_r2s.sh was the fp32 1024 x 192 twin)
mkdir -p gpurun_out/step_breakdown
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/step_breakdown -o st -- python $R/bench.py --bf16 --rays-per-gpu 4096 --samples 128 --no-extra --no-cpu-baseline --steps 40 --warmup 10 > $R/gpurun_out/step_breakdown/bench.txt 2>/dev/null
cd $R
f=$(find /tmp/step_breakdown -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/step_breakdown/stats.csv
t=$(find /tmp/step_breakdown -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 20 steps: find the last 21 launches of mlp_fwd_bf16 train
idx = [i for i, r in enumerate(rows) if 'mlp_fwd_bf16_kernel<256, true' in r['Kernel_Name']]
idx = idx[:50]          # the 10 + 40 steps of the training loop (later launches belong to the kernel-roofline block)
a, b = idx[-21], idx[-1]
seg = rows[a:b]
wall = int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
print('20 training steps, bf16 4096 x 128 (rocprofv3 --kernel-trace; the tracer slows the HOST, so wall time here is not the step time):')
print('GPU busy %.3f ms/step in %.1f launches/step (traced wall %.3f ms/step; untraced step time: the bench line)' % (busy / 20e6, len(seg) / 20, wall / 20e6))
per = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r['Kernel_Name'][:70]
    per[k][0] += 1; per[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
    print('%6.1f us/step  x%.1f  %s' % (t / 20e3, n / 20, k))
PY
python - <<'PY'
import json
d = json.loads(open('gpurun_out/step_breakdown/bench.txt').read().strip().splitlines()[-1])
print('bench line of the traced run: %.3f ms/step' % d['ms_per_step'])
PY
