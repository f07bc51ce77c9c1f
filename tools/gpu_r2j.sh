mkdir -p gpurun_out/r2j
export PYTHONUNBUFFERED=1
R=$PWD
python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/r2j/tests.txt
timeout 200 python tools/time_kernels.py 4096 128 bf16 20 2>&1 | grep -i "wgrad\|step" > gpurun_out/r2j/time.txt
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2j/prof -o wg -- python $R/tools/time_kernels.py 4096 128 bf16 8 > /dev/null 2>&1
cd $R; f=$(find gpurun_out/r2j/prof -name "*kernel_stats.csv" | head -1); cut -d, -f1-4,6-7 $f | head -8 > gpurun_out/r2j/stats.txt
cat gpurun_out/r2j/tests.txt gpurun_out/r2j/time.txt gpurun_out/r2j/stats.txt
