# round 6, GPU call t: the 128 x 64 encoding tiles of the weight gradient as private two-term jobs: correctness, then the plan weight
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_gpu_layer_local.py tests/test_gpu_parity.py tests/test_gpu_determinism.py -q -m gpu --maxfail=5 2>&1 | tail -6
for w in 0 250 350 450 600; do echo "== NNR_WGRAD_ENC2_WEIGHT=$w"; NNR_WGRAD_ENC2_WEIGHT=$w timeout 300 python tools/time_kernels.py 1024 192 f32 5 2>&1 | tail -1 | sed 's/.*in_sequence_ms/seq/'; done | tee gpurun_out/r06/t_wgrad_enc2_weight_sweep.txt
