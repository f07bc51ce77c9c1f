# usage: bash tools/gpu_pmc_variants.sh v1 v2 ...  -- SQ counter pass (incl. GRBM_GUI_ACTIVE = shader clock ticks) per variant
mkdir -p gpurun_out/pmcv
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in product "$@"; do
  if [ "$v" != product ]; then export NNR_LIB=$R/nope-nerf_amd/nnr/libnnr_$v.so; else unset NNR_LIB; fi
  P=/tmp/pmcv_$v; mkdir -p $P
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_WR -d $P -o pmc -- python $R/tools/profile_kernels.py 2 $PROFILE_SHAPE > $R/gpurun_out/pmcv/$v.log 2>&1
  f=$(find $P -name "*counter_collection.csv" | head -1)
  python - "$f" "$v" <<'PY' | tee $R/gpurun_out/pmcv/$v.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for r in rows:
    k=r['Kernel_Name']
    if 'mlp_' not in k and 'wgrad' not in k: continue
    k=k.replace('void nnr::','')[:34]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    d=sum(dur[k])/len(dur[k])
    c={n:sum(x)/len(x) for n,x in v.items()}
    print(sys.argv[2],k,'us=%.0f'%d,' '.join('%s=%.4g'%(n,x) for n,x in sorted(c.items())), 'MHz=%.0f'%(c.get('GRBM_GUI_ACTIVE',0)/d if d else 0))
PY
done
