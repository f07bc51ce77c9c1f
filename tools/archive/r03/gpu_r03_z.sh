#!/bin/bash
# round 3, final call: the whole GPU suite, smoke, the round-end profile set and the untraced bench of the final binaries
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -30 > gpurun_out/r03/z3_gpu_suite.txt
echo "pytest exit $?"; grep "passed\|failed" gpurun_out/r03/z3_gpu_suite.txt | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03/z3_smoke.txt 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/r03/z3_smoke.txt
bash tools/gpu_r03_j.sh
cp gpurun_out/r03/j_bench_untraced.json.txt gpurun_out/r03/z3_round_end_bench_untraced.json.txt
