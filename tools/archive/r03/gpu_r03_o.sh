#!/bin/bash
# round 3, call O: where the three-term forward / input-gradient kernels spend their time (profiling variants, results NOT valid)
mkdir -p gpurun_out/r03
export NNR_FP32_PRODUCTS=split3
SHAPE="1024 192 f32" bash tools/gpu_variants.sh noside nosplit nosync nostash nodma terms1 terms3 > gpurun_out/r03/o_split3_variants.txt 2>&1
cat gpurun_out/r03/o_split3_variants.txt
