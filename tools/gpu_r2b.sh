# round 2, GPU pass b: the new bf16 weight-gradient kernel + tile-major planes -- parity first, then timing
set -x
mkdir -p gpurun_out/r2b
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -x -k "bf16" > gpurun_out/r2b/pytest_bf16.txt 2>&1; echo "bf16 pytest exit $?" >> gpurun_out/r2b/pytest_bf16.txt
tail -25 gpurun_out/r2b/pytest_bf16.txt
timeout 300 python bench.py --bf16 --rays-per-gpu 4096 --samples 128 --no-extra --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r2b/bench_bf16.txt 2> gpurun_out/r2b/bench_bf16.err; echo "bench exit $?"
tail -c 1800 gpurun_out/r2b/bench_bf16.txt; tail -3 gpurun_out/r2b/bench_bf16.err
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r2b/pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/r2b/pytest.txt
grep -E "passed|failed|Error|error|bf16|normal|fp32:" gpurun_out/r2b/pytest.txt | tail -30
