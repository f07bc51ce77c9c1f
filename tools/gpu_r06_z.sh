# round 6, GPU call z: the final binaries -- full GPU suite, smoke, the round's profile set (tools/gpu_prof_round.sh), an untraced default bench
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
export NNR_PARITY_LOG=$PWD/gpurun_out/r06/z_parity_rel_l2.txt; rm -f $NNR_PARITY_LOG
timeout 2400 python -m pytest tests -q -m gpu -rs 2>&1 | tail -25 > gpurun_out/r06/z_gpu_suite.txt; tail -3 gpurun_out/r06/z_gpu_suite.txt
unset NNR_PARITY_LOG
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/z_smoke.txt 2>&1; tail -2 gpurun_out/r06/z_smoke.txt
bash tools/gpu_prof_round.sh > gpurun_out/r06/z_prof_round.log 2>&1
timeout 900 python bench.py > gpurun_out/r06/z_bench_untraced.json.txt 2> gpurun_out/r06/z_bench_untraced.err; tail -c 400 gpurun_out/r06/z_bench_untraced.json.txt
bash tools/gpu_step_breakdown.sh fp32_1024x192 > gpurun_out/r06/z_step_breakdown.log 2>&1; bash tools/gpu_step_breakdown.sh fp32_1024x192_aux --aux >> gpurun_out/r06/z_step_breakdown.log 2>&1; head -12 gpurun_out/step_breakdown/fp32_1024x192.txt
