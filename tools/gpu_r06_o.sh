# round 6, GPU call o: full GPU suite + smoke on the fp16-term weight gradient / double front end, then the round's profile set
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r06/o_gpu_suite.txt; tail -5 gpurun_out/r06/o_gpu_suite.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/o_smoke.txt 2>&1; tail -2 gpurun_out/r06/o_smoke.txt
bash tools/gpu_prof_round.sh > gpurun_out/r06/o_prof_round.log 2>&1
tail -c 600 gpurun_out/bench.txt
