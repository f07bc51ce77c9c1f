# round 3, call H: the whole GPU suite; bf16 timings; shader-clock timeline of the bf16 kernels (libnnr_tl.so); bench
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r03/h_gpu_tests.txt 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/r03/h_gpu_tests.txt
for i in 1 2; do timeout 200 python tools/time_kernels.py 4096 128 bf16 20 2>&1 | tail -1; done | tee gpurun_out/r03/h_bf16_time.txt
NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_tl.so timeout 300 python tools/timeline.py 4096 128 bf16 2>&1 | grep -E "nnr_timeline|rror" | tee gpurun_out/r03/h_bf16_timeline.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03/h_bench.json.txt 2> gpurun_out/r03/h_bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03/h_bench.json.txt').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['roofline']['kernels'].items()})
c = d['configs']['bf16_4096x128']
print('bf16', c['value'], c['ms_per_step'], c['kernels_ms'])
PY
