# round 3, call I: paired epilogue units -- parity (layer-local tests check every ReLU gate), then the A/B against libnnr_nopairs.so;
# two-phase convergence replays (fp32 + bf16)
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layer_local.py tests/test_gpu_bench_shape_parity.py tests/test_gpu_determinism.py -m gpu -q -k "bf16 or layer or determin or inference" > gpurun_out/r03/i_bf16_tests.txt 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r03/i_bf16_tests.txt
for rep in 1 2; do for v in product nopairs; do
  if [ "$v" != product ]; then export NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_$v.so; else unset NNR_LIB; fi
  echo -n "$v: "; timeout 200 python tools/time_kernels.py 4096 128 bf16 20 2>&1 | tail -1
done; done | tee gpurun_out/r03/i_pairs_ab.txt
unset NNR_LIB
timeout 900 python -m pytest tests/test_conv_reference.py -m gpu -q -s > gpurun_out/r03/i_conv_tests.txt 2>&1; echo "conv pytest exit $?"; grep -E "two-phase replay|HIP vs reference|passed|failed|Error" gpurun_out/r03/i_conv_tests.txt | cut -c1-420
