# cost of a data-parallel rank beyond the single-GPU step, without and with the per-image block, for W = 2, 4, 8 virtual ranks
mkdir -p gpurun_out/dp
export PYTHONUNBUFFERED=1
for w in 2 4 8; do
  timeout 200 python tools/dp_overhead.py --world $w --steps 40 2>/dev/null | tail -1
  timeout 200 python tools/dp_overhead.py --world $w --steps 40 --aux 2>/dev/null | tail -1
done | tee gpurun_out/dp/overhead.txt
