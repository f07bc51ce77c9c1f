# round 6, GPU call e: SQ counter passes over the three MLP kernels at the headline shape (where do the cycles of the fp16-term kernels go?)
mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
R=$PWD
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $P/pmc1 -o pmc1 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/r06/e_pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $P/pmc2 -o pmc2 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/r06/e_pmc2.log 2>&1
cd $R
python - <<'PY'
import csv,collections,glob
for tag in ('pmc1','pmc2'):
    f=glob.glob('/tmp/prof/%s/**/*counter_collection.csv'%tag, recursive=True)
    if not f: print('no file for',tag); continue
    a=collections.defaultdict(lambda: collections.defaultdict(list))
    dur=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name']
        if 'mlp_' in k or 'wgrad_kernel' in k:
            a[k][r['Counter_Name']].append(float(r['Counter_Value'])); dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    out=open('gpurun_out/r06/e_%s_summary.txt'%tag,'w')
    for k,d in a.items():
        line='%-60s us %.0f '%(k.replace('void ','')[:60], sum(dur[k])/len(dur[k]))+' '.join('%s=%.4g'%(c,sum(v)/len(v)) for c,v in sorted(d.items()))
        print(line); out.write(line+'\n')
PY
