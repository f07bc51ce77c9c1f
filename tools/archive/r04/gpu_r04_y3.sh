#!/bin/bash
# round 4, call Y3: seeded nearest-neighbour search with one / two sources per lane (NNR_PC_SRC) over the number of workgroups; the forward /
# input-gradient schedule variants (stash store counted as one instruction, side-unit cost 6 / 8) timed in sequence (variant results are valid:
# only the placement of operations in the MFMA gaps differs)
mkdir -p gpurun_out/r04
export PYTHONUNBUFFERED=1
for S in 20736 32400; do for nsrc in 1 2; do for w in 1024 2048 4096 8192; do
  echo -n "NNR_PC_SRC=$nsrc NNR_PC_WGS=$w: "; NNR_PC_SRC=$nsrc NNR_PC_WGS=$w timeout 120 python tools/time_pc_nearest.py $S smooth 2>&1 | tail -1
done; done; done > gpurun_out/r04/y3_pc_nearest_src.txt 2>&1
sed 's/nnr_pc_nearest //; s/ per call (fill + search + decode)//' gpurun_out/r04/y3_pc_nearest_src.txt
export NNR_FP32_PRODUCTS=split3
SHAPE="1024 192 f32" bash tools/gpu_variants.sh $VARIANTS product $VARIANTS > gpurun_out/r04/y3_schedule_variants_in_sequence.txt 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r04/y3_schedule_variants_in_sequence.txt'):
    n, j = l.split(': ', 1)
    try:
        d = json.loads(j); print('%-10s isolated %s | in sequence %s' % (n, {k: round(v, 4) for k, v in d['ms'].items() if 'mlp' in k}, {k: round(v, 4) for k, v in d['in_sequence_ms'].items() if 'mlp' in k}))
    except Exception as e:
        print(n, j[:300])
PY
