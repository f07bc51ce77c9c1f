# round 3, call E: Adam arithmetic (second round of variants); bf16 kernels: product (batched pass start in the forward) vs mask-load /
# factor-load ablations of the input-gradient kernel; bf16 parity tests of the changed forward; HIP-side convergence envelope
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1
timeout 600 python tools/adam_variants.py 2>&1 | grep -v Warn > gpurun_out/r03/e_adam_variants.txt; cat gpurun_out/r03/e_adam_variants.txt | tail -34
bash tools/gpu_variants.sh nomaskload noload > gpurun_out/r03/e_bf16_variants.txt 2>&1; cat gpurun_out/r03/e_bf16_variants.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layer_local.py tests/test_gpu_bench_shape_parity.py -m gpu -q -k "bf16 or layer" > gpurun_out/r03/e_bf16_tests.txt 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r03/e_bf16_tests.txt
timeout 900 python tools/conv_envelope_hip.py 2>&1 | grep -v Warn > gpurun_out/r03/e_conv_envelope_hip.txt; tail -8 gpurun_out/r03/e_conv_envelope_hip.txt
