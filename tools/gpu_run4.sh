set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.txt 2>&1; echo "bench exit $?" >> gpurun_out/bench.txt
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/bench -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
cd $P; mkdir -p $R/gpurun_out/prof; find . -name "*.csv" -size -8M -exec cp --parents {} $R/gpurun_out/prof/ \;
cd $R; grep -E "passed|failed|Error" gpurun_out/pytest.txt | tail -12; tail -2 gpurun_out/bench.txt
