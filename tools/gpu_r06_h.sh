# round 6, GPU call h: the fp16-term weight gradient (correctness, plan-weight sweep) and the compositing backward in double (stage bisect over 12 seeds, A/B)
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
O=gpurun_out/r06
NNR_BISECT_KINDS=split2 timeout 300 python tools/fp64_bisect.py 256 64 64 333 > $O/h_bisect_d256.txt 2>&1
grep -E "^d pre1|^d point|dW layers0.0|dW layers0.2|dW layers1.0|dW layers1.6|dW fc_feature|dW rgb_layers" $O/h_bisect_d256.txt | cut -c1-110
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_determinism.py tests/test_gpu_layer_local.py -x -q -m gpu 2>&1 | tail -4
for w in 240 280 320 360 400; do echo "== split2 weight $w"; NNR_WGRAD_SPLIT2_WEIGHT=$w timeout 300 python tools/time_kernels.py 1024 192 f32 5 2>&1 | tail -1 | sed 's/.*in_sequence_ms/seq/'; done | tee $O/h_wgrad_f16_weight_sweep.txt
echo "== six-term workgroup jobs (NNR_WGRAD_BF16_TERMS=1)"; NNR_WGRAD_BF16_TERMS=1 timeout 300 python tools/time_kernels.py 1024 192 f32 5 2>&1 | tail -1 | sed 's/.*in_sequence_ms/seq/' | tee -a $O/h_wgrad_f16_weight_sweep.txt
timeout 900 python tools/fp64_bisect_seeds.py 256 256 64 12 > $O/h_bisect_seeds_double_composite.txt 2>&1; tail -62 $O/h_bisect_seeds_double_composite.txt | cut -c1-120 | grep -E "^d |^dW|^db|^RGB|^alpha|^h8|D=" | head -70
NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_comp32.so timeout 900 python tools/fp64_bisect_seeds.py 256 256 64 12 > $O/h_bisect_seeds_fp32_composite.txt 2>&1; grep -E "^d sigma_raw|^d pre8|^d pre1|^d point|^d pts_o|^d pts_d|dW layers0.6|dW layers0.0|db layers0.6" $O/h_bisect_seeds_fp32_composite.txt | cut -c1-120
