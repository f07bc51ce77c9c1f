#!/bin/bash
# round 3, call Q2: HBM traffic of the three-term kernels (is the weight stream still L2-resident next to the stash?)
mkdir -p gpurun_out/r03
export PYTHONUNBUFFERED=1 NNR_FP32_PRODUCTS=split3
R=$PWD
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof; mkdir -p $P
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $P/q4 -o q4 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/r03/q4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $P/q5 -o q5 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/r03/q5.log 2>&1
timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum -d $P/q6 -o q6 -- python $R/tools/profile_kernels.py 2 > $R/gpurun_out/r03/q6.log 2>&1
cd $P
for q in q4 q5 q6; do f=$(find $q -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $R/gpurun_out/r03/${q}_split3_counters.csv; done
python3 - <<PY
import csv, collections
for q in ("q4","q5","q6"):
    try: rows=list(csv.DictReader(open("$R/gpurun_out/r03/%s_split3_counters.csv"%q)))
    except Exception as e: print(q, e); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k=r["Kernel_Name"]
        name="fwd_train" if "mlp_fwd_kernel<256, true" in k else "fwd_infer" if "mlp_fwd_kernel<256, false" in k else "dgrad" if "dgrad" in k else "wgrad" if "wgrad_kernel" in k else None
        if name: agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name,c in agg.items(): print(q,name,{k:round(sum(v)/len(v)) for k,v in c.items()})
PY
