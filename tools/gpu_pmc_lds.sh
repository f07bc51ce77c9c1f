# LDS / issue counters of the bf16 kernels at 4096 x 128 (two passes: SQ counters are limited per pass)
mkdir -p gpurun_out/pmclds
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1)); P=/tmp/pmclds_$i; mkdir -p $P
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $set -d $P -o pmc -- python $R/tools/profile_kernels.py 2 4096 128 bf16 > $R/gpurun_out/pmclds/run$i.log 2>&1
  f=$(find $P -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY' | tee -a $R/gpurun_out/pmclds/summary.txt
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name']
    if 'mlp_' not in k and 'wgrad_b_k' not in k: continue
    agg[k.replace('void nnr::','')[:36]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, ' '.join('%s=%.4g'%(n,sum(x)/len(x)) for n,x in sorted(v.items())))
PY
done
