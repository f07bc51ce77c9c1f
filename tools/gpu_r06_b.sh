mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
O=gpurun_out/r06
NNR_BISECT_KINDS=split2,split3 timeout 300 python tools/fp64_bisect.py 128 64 64 333 > $O/b_bisect_d128.txt 2>&1
NNR_BISECT_KINDS=split2,split3 timeout 300 python tools/fp64_bisect.py 256 64 64 333 > $O/b_bisect_d256.txt 2>&1
cut -c1-140 $O/b_bisect_d128.txt | head -60; cut -c1-140 $O/b_bisect_d256.txt | head -60
