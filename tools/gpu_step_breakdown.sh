# Per-step GPU time of a training step by kernel (rocprofv3 --kernel-trace over bench.py): is the GPU ever idle, how many launches, what do
# the small ones cost?     bash tools/gpu_step_breakdown.sh <tag> [bench.py arguments]      -> gpurun_out/step_breakdown/<tag>.txt
# e.g.  ... fp32_1024x192            (the headline step)        ... fp32_1024x192_aux --aux        ... bf16_4096x128 --bf16 --rays-per-gpu 4096 --samples 128
tag=${1:-fp32_1024x192}; shift
mkdir -p gpurun_out/step_breakdown
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/step_breakdown_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/step_breakdown_$tag -o st -- python $R/bench.py "$@" --no-extra --no-cpu-baseline --steps 40 --warmup 10 > $R/gpurun_out/step_breakdown/$tag.bench.txt 2>/dev/null
cd $R
t=$(find /tmp/step_breakdown_$tag -name "*kernel_trace.csv" | head -1)
python - "$t" "$tag" > gpurun_out/step_breakdown/$tag.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the training forward marks a step: the first 50 launches of it are the 10 + 40 steps of the loop (later ones belong to the kernel-roofline block)
idx = [i for i, r in enumerate(rows) if ('mlp_fwd_kernel<256, true' in r['Kernel_Name'] or 'mlp_fwd_f16_kernel<256, true' in r['Kernel_Name'] or 'mlp_fwd_bf16_kernel<256, true' in r['Kernel_Name'])][:50]
a, b = idx[-21], idx[-1]
seg = rows[a:b]
wall = int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
main = ('mlp_fwd_kernel', 'mlp_dgrad_kernel', 'mlp_fwd_f16_kernel', 'mlp_dgrad_f16_kernel', 'wgrad_kernel', 'mlp_fwd_bf16_kernel', 'mlp_dgrad_bf16_kernel', 'wgrad_b_kernel')
small = [r for r in seg if not any(m + '<' in r['Kernel_Name'] or ('::' + m + '(') in r['Kernel_Name'] for m in main)]
print('20 training steps, %s (rocprofv3 --kernel-trace; the tracer slows the HOST, so wall time here is not the step time):' % sys.argv[2])
print('GPU busy %.3f ms/step in %.1f launches/step (traced wall %.3f ms/step; untraced step time: the bench line)' % (busy / 20e6, len(seg) / 20, wall / 20e6))
print('small launches (everything but the three main MLP kernels): %.1f us/step in %.1f launches' % (sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in small) / 20e3, len(small) / 20))
per = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r['Kernel_Name'][:90]
    per[k][0] += 1; per[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:60]:
    print('%6.1f us/step  x%.1f  %s' % (t / 20e3, n / 20, k))
PY
python - "$tag" >> gpurun_out/step_breakdown/$tag.txt <<'PY'
import json, sys
d = json.loads(open('gpurun_out/step_breakdown/%s.bench.txt' % sys.argv[1]).read().strip().splitlines()[-1])
print('bench line of the traced run: %.3f ms/step' % d['ms_per_step'])
PY
cat gpurun_out/step_breakdown/$tag.txt
