export PYTHONUNBUFFERED=1
{
for per in 4 2; do for wgs in 2048 4096 8192; do
  echo -n "NNR_PC_PER=$per NNR_PC_WGS=$wgs: "; NNR_PC_PER=$per NNR_PC_WGS=$wgs timeout 120 python tools/time_pc_nearest.py 32400 2>&1 | tail -1
done; done
echo -n "S=16128 PER=2 WGS=2048: "; NNR_PC_PER=2 NNR_PC_WGS=2048 timeout 120 python tools/time_pc_nearest.py 16128 2>&1 | tail -1
echo -n "S=16128 PER=4 WGS=1024: "; NNR_PC_PER=4 NNR_PC_WGS=1024 timeout 120 python tools/time_pc_nearest.py 16128 2>&1 | tail -1
} >> gpurun_out/r04/k_pc_nearest_sweep.txt 2>&1
tail -8 gpurun_out/r04/k_pc_nearest_sweep.txt
