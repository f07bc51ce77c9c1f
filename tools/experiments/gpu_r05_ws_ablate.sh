#!/bin/bash
# profiling builds of the wave-specialised forward (csrc/build.py --ws-variant), timed against the product kernel: python tools/ws_check.py --timing-only
mkdir -p gpurun_out
for v in product nobar nodmawait nodma nointake noterms nodrain hidle midle hidle_nodrain hidle_nobar; do
  if [ $v = product ]; then lib=nope-nerf_amd/nnr/libnnr.so; else lib=nope-nerf_amd/nnr/libnnr_ws_$v.so; fi
  [ -f $lib ] || continue
  echo "== $v"
  NNR_LIB=$PWD/$lib timeout 200 python tools/ws_check.py --timing-only 2>&1 | grep -o '"fused": [a-z]*\|"ms_[a-z_]*": [0-9.]*' | tr '\n' ' '
  echo
done
