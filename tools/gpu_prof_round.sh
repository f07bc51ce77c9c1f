# Round profile set (r02 script, unchanged commands; r05: + the 1024 x 128 shape's passes): bench.py under --kernel-trace --stats, then separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ/clock) over
# tools/profile_kernels.py for the headline shape (fp32 1024 x 192) and for BASELINE configs[2] (bf16 4096 x 128).
set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof; mkdir -p $P
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/bench -o bench -- python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/bench.txt 2> $R/gpurun_out/prof_bench.log
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/headline -o headline -- python $R/bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $R/gpurun_out/bench_headline_only.txt 2> $R/gpurun_out/prof_headline.log
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/pmc1 -o pmc1 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $P/pmc3 -o pmc3 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $P/pmc4 -o pmc4 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc4.log 2>&1
N=bf16_4096x128
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/${N}_stats -o ${N}_stats -- python $R/tools/profile_kernels.py 8 4096 128 bf16 > $R/gpurun_out/${N}_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $P/${N}_fetch -o ${N}_fetch -- python $R/tools/profile_kernels.py 2 4096 128 bf16 > $R/gpurun_out/${N}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $P/${N}_write -o ${N}_write -- python $R/tools/profile_kernels.py 2 4096 128 bf16 > $R/gpurun_out/${N}_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/${N}_sq -o ${N}_sq -- python $R/tools/profile_kernels.py 2 4096 128 bf16 > $R/gpurun_out/${N}_sq.log 2>&1
N=fp32_1024x128
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/${N}_stats -o ${N}_stats -- python $R/tools/profile_kernels.py 8 1024 128 > $R/gpurun_out/${N}_stats.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $P/${N}_fetch -o ${N}_fetch -- python $R/tools/profile_kernels.py 2 1024 128 > $R/gpurun_out/${N}_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $P/${N}_write -o ${N}_write -- python $R/tools/profile_kernels.py 2 1024 128 > $R/gpurun_out/${N}_write.log 2>&1
cd $P; mkdir -p $R/gpurun_out/prof
find . -name "*.csv" -size -8M -exec cp --parents {} $R/gpurun_out/prof/ \;
du -sh $R/gpurun_out; tail -c 1500 $R/gpurun_out/bench.txt
