# round 2, first GPU pass: parity suite, the bench line with the extra config blocks, the self-launched 2-rank dry run
set -x
mkdir -p gpurun_out/r2a
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -x > gpurun_out/r2a/pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r2a/pytest.txt
grep -E "passed|failed|Error|error" gpurun_out/r2a/pytest.txt | tail -8
timeout 600 python bench.py > gpurun_out/r2a/bench.txt 2> gpurun_out/r2a/bench.err; echo "bench exit $?"
tail -c 3000 gpurun_out/r2a/bench.txt
NNR_ALLOW_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r2a/bench_g2.txt 2> gpurun_out/r2a/bench_g2.err; echo "bench g2 exit $?"
tail -c 1500 gpurun_out/r2a/bench_g2.txt; tail -5 gpurun_out/r2a/bench_g2.err
timeout 300 python tools/hbm_bw.py > gpurun_out/r2a/hbm_bw.txt 2>&1; tail -5 gpurun_out/r2a/hbm_bw.txt
timeout 120 ./tools/ubench/hbm_stream > gpurun_out/r2a/hbm_stream.txt 2>&1; cat gpurun_out/r2a/hbm_stream.txt
