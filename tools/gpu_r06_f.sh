# round 6, GPU call f: ablation builds of the fp16-term kernels (results NOT valid): where does the time go?
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
for v in "" $EXTRA_VARIANTS; do
  if [ -z "$v" ]; then echo "== product"; timeout 300 python tools/time_kernels.py 1024 192 f32 5 2>&1 | tail -1
  else echo "== $v"; NNR_LIB=$PWD/nope-nerf_amd/nnr/libnnr_$v.so timeout 300 python tools/time_kernels.py 1024 192 f32 5 2>&1 | tail -1; fi
done | tee gpurun_out/r06/${1:-f}_ablations.txt
