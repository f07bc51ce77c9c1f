mkdir -p gpurun_out/r2h
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2h/prof -o wg -- python $R/tools/time_kernels.py 4096 128 bf16 8 > /dev/null 2>&1
cd $R; f=$(find gpurun_out/r2h/prof -name "*kernel_stats.csv" | head -1); cut -d, -f1-4,6-7 $f | head -14
