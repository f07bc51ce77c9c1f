set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_torchrun.txt 2>&1; echo "exit $?" >> gpurun_out/bench_torchrun.txt
grep -E "passed|failed|Error|error" gpurun_out/pytest.txt | tail -8; tail -3 gpurun_out/bench_torchrun.txt | cut -c1-600
