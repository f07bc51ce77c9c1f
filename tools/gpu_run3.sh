set -x
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
R=$PWD
timeout 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/pytest.txt 2>&1; echo "pytest exit $?" >> gpurun_out/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.txt 2>&1; echo "bench exit $?" >> gpurun_out/bench.txt
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/pmc1 -o pmc1 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/pmc1.log 2>&1
cd $P; mkdir -p $R/gpurun_out/prof; find . -name "*.csv" -size -8M -exec cp --parents {} $R/gpurun_out/prof/ \;
cd $R; tail -6 gpurun_out/pytest.txt; tail -2 gpurun_out/bench.txt
