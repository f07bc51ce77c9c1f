set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/bench -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
timeout 300 rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/pmc1 -o pmc1 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $P/pmc2 -o pmc2 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $P/pmc3 -o pmc3 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $P/pmc4 -o pmc4 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc4.log 2>&1
cd $P; find . -type f | head -50; du -sh .
mkdir -p $R/gpurun_out/prof
find . -name "*.csv" -size -8M -exec cp --parents {} $R/gpurun_out/prof/ \;
du -sh $R/gpurun_out
