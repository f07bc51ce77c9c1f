# round 6, GPU call q: class-B bundles in the weight-gradient plan: tests that see the plan, kernel times with / without, FETCH_SIZE with / without
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
O=$PWD/gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py tests/test_gpu_layer_local.py -q -m gpu --maxfail=5 2>&1 | tail -4
for e in "" "NNR_WGRAD_NO_BUNDLES=1"; do echo "== $e"; env $e timeout 300 python tools/time_kernels.py 1024 192 f32 5 2>&1 | tail -1 | sed 's/.*in_sequence_ms/seq/'; env $e timeout 300 python tools/time_kernels.py 1024 192 f32 5 2>&1 | tail -1 | sed 's/.*in_sequence_ms/seq/'; done | tee $O/q_wgrad_bundles_ab.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for e in "X=1" "NNR_WGRAD_NO_BUNDLES=1"; do
  rm -rf /tmp/pq; env $e timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d /tmp/pq -o pq -- python $R/tools/profile_kernels.py 2 > /dev/null 2>&1
  echo "== $e" >> $O/q_wgrad_bundles_ab.txt
  python - <<'PY' >> $O/q_wgrad_bundles_ab.txt
import csv, glob, collections
f = glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True)[0]
a = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'wgrad_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE': a[r['Kernel_Name']].append(float(r['Counter_Value']))
for k, v in a.items(): print(k[:60], 'fetch %.3f GB per launch (FETCH_SIZE x 2 x 1024)' % (2 * 1024 * sum(v) / len(v) / 1e9))
PY
done
cat $O/q_wgrad_bundles_ab.txt
cd $R; timeout 600 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d['value'], d['ms_per_step'], d['step_ms']['median'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
