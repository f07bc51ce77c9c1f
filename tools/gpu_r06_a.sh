# round 6, GPU call a: the two-term fp16 products -- micro test (gate i), first numbers of the kernels against the fp32-MFMA / six-term kernels,
# the reference goldens in the new mode, kernel timings of both modes
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
O=gpurun_out/r06
timeout 120 gpurun_stage/split2_f16 > $O/a_split2_micro.txt 2>&1; echo "micro rc $?"
for shape in "64 64 128" "256 64 256" "1024 192 256"; do timeout 300 python tools/split2_debug.py $shape; done > $O/a_split2_debug.txt 2>&1; echo "debug rc $?"
tail -12 $O/a_split2_debug.txt
NNR_FP32_PRODUCTS=split2 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 > $O/a_split2_parity.txt; cat $O/a_split2_parity.txt
for k in split3 split2; do NNR_FP32_PRODUCTS=$k timeout 300 python tools/time_kernels.py 1024 192 f32 5; done > $O/a_time_kernels.txt 2>&1; cat $O/a_time_kernels.txt
