# round 6, GPU call d (iteration loop): bisect of the fp16-term kernels against fp64, goldens, kernel timings, a short headline bench
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
O=gpurun_out/r06
T=${1:-d}
NNR_BISECT_KINDS=split2 timeout 300 python tools/fp64_bisect.py 128 64 64 333 > $O/${T}_bisect_d128.txt 2>&1
NNR_BISECT_KINDS=split2 timeout 300 python tools/fp64_bisect.py 256 64 64 333 > $O/${T}_bisect_d256.txt 2>&1
for f in $O/${T}_bisect_d128.txt $O/${T}_bisect_d256.txt; do grep -E "^h8|^RGB|^d pre1|^d point|^d view|dW layers0.0|dW layers1.6|dW rgb_layers" $f | cut -c1-110; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python tools/time_kernels.py 1024 192 f32 5 > $O/${T}_time_kernels.txt 2>&1; cat $O/${T}_time_kernels.txt | tail -1
timeout 600 python bench.py --no-extra --no-cpu-baseline > $O/${T}_bench.txt 2>&1; python - $T <<'PY'
import json,sys
l=[x for x in open('gpurun_out/r06/'+(sys.argv[1] if len(sys.argv)>1 else 'd')+'_bench.txt') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('bench', d['value'], 'rays/s', d['ms_per_step'], 'ms', {k:v['ms'] for k,v in d['roofline']['kernels'].items()})
else: print("no bench line")
PY
