#!/bin/bash
# round 4, call L: what a data-parallel rank pays beyond the single-GPU step (virtual 8-rank job on one GPU), by kernel
export PYTHONUNBUFFERED=1 HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/r04
R=$PWD
for w in 2 8; do timeout 300 python tools/dp_overhead.py --world $w --steps 60 2>&1 | tail -1; done > gpurun_out/r04/l_dp_overhead.txt
cat gpurun_out/r04/l_dp_overhead.txt
( cd /tmp && TMPDIR=/tmp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dpo -o dpo -- python $R/tools/dp_overhead.py --world 8 --steps 40 --skip-single > /dev/null 2>&1 )
f=$(find /tmp/dpo -name "*kernel_stats.csv" | head -1)
python - "$f" > gpurun_out/r04/l_dp_virtual_rank_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
print('virtual rank 0 of 8 (48 steps incl. warm-up): per-step kernel time, everything but the three MLP kernels')
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs'])):
    n, t = int(r['Calls']), float(r['TotalDurationNs'])
    if any(k in r['Name'] for k in ('mlp_fwd', 'mlp_dgrad', 'wgrad_kernel')): continue
    tot += t
    print('%7.1f us/step x%.1f  %s' % (t / 48e3, n / 48, r['Name'][:90]))
print('total %.1f us/step' % (tot / 48e3))
PY
head -30 gpurun_out/r04/l_dp_virtual_rank_kernels.txt | cut -c1-150
