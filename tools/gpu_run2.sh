set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
timeout 600 python tests/gpu_diag.py > gpurun_out/diag.txt 2>&1; echo "diag exit $?" >> gpurun_out/diag.txt
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.txt 2>&1; echo "bench exit $?" >> gpurun_out/bench.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
timeout 300 rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc1 -o pmc1 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum -d $R/gpurun_out/pmc2 -o pmc2 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc3 -o pmc3 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc4 -o pmc4 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc4.log 2>&1
cd $R
ls -R gpurun_out | head -50
tail -3 gpurun_out/diag.txt; tail -8 gpurun_out/pytest.txt; tail -2 gpurun_out/bench.txt
