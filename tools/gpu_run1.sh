set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python tests/gpu_diag.py > gpurun_out/diag.txt 2>&1; echo "diag exit $?" >> gpurun_out/diag.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.txt 2>&1; echo "bench exit $?" >> gpurun_out/bench.txt
tail -5 gpurun_out/diag.txt; tail -15 gpurun_out/pytest.txt; tail -3 gpurun_out/bench.txt
