set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.txt
grep -E "passed|failed|Error|error" gpurun_out/pytest.txt | tail -8
bash tools/gpu_timelines.sh timeline
bash tools/gpu_variants.sh
