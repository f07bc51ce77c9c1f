set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
P=/tmp/profaux; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o aux -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --aux > $R/gpurun_out/prof_aux.log 2>&1
mkdir -p $R/gpurun_out/profaux; find $P -name "*.csv" -size -8M -exec cp {} $R/gpurun_out/profaux/ \;
