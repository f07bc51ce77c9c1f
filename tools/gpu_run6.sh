set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.txt
timeout 300 python tools/profile_kernels.py 5 > gpurun_out/kernels.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench.txt 2>&1; echo "bench exit $?" >> gpurun_out/bench.txt
grep -E "passed|failed|Error|error" gpurun_out/pytest.txt | tail -8; grep kernels gpurun_out/kernels.txt; tail -2 gpurun_out/bench.txt | cut -c1-700
