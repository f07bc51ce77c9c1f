#!/bin/bash
# The reference's UNMODIFIED train.py on the HIP kernels.  Run from the authoring container:
#     bash tools/gpu_dropin.sh
# 1. stages /root/reference/{train.py, configs/default.yaml} into gpurun_stage/ref/ -- git-ignored (reference sources never enter the
#    history) but shipped to the GPU box by gpurun, like the built libnnr.so;
# 2. runs train.py HERE on the CPU with the oracle-backed operator (tests/dropin_runner.py), random draws from torch's CPU generator;
# 3. runs the very same script on the MI355X with the HIP kernels and the same draws, and compares the logged scalars step by step
#    (tools/dropin_compare.py) -> gpurun_out/dropin/compare.json (copy into profiles/<round>/ to keep it).
set -e
cd "$(dirname "$0")/.."
REF=${NNR_REFERENCE:-/root/reference}
rm -rf gpurun_stage/out_cpu gpurun_stage/out_gpu        # train.py RESUMES from a checkpoint it finds in out_dir
mkdir -p gpurun_stage/ref/configs gpurun_stage/scene gpurun_out/dropin
cp "$REF/train.py" gpurun_stage/ref/train.py
cp "$REF/configs/default.yaml" gpurun_stage/ref/configs/default.yaml
python tools/dropin_compare.py prepare
( cd gpurun_stage/ref && DROPIN_CPU_DRAWS=1 DROPIN_SCALARS=../scalars_cpu.json NNR_REFERENCE=$PWD PYTHONPATH= \
    python ../../tests/dropin_runner.py train.py ../dropin_cpu.yaml > ../train_cpu.log 2>&1 ) || { tail -30 gpurun_stage/train_cpu.log; exit 1; }
grep -E "PSNR|ATE" gpurun_stage/train_cpu.log | tail -3
/usr/local/graft/bin/gpurun --timeout 900 -- 'mkdir -p gpurun_out/dropin && cd gpurun_stage/ref && DROPIN_BACKEND=hip DROPIN_CPU_DRAWS=1 DROPIN_SCALARS=../../gpurun_out/dropin/scalars_gpu.json NNR_REFERENCE=$PWD PYTHONPATH= timeout 600 python ../../tests/dropin_runner.py train.py ../dropin_gpu.yaml > ../../gpurun_out/dropin/train_gpu.log 2>&1; echo "train.py exit $?"; tail -5 ../../gpurun_out/dropin/train_gpu.log; cd ../.. && python tools/dropin_compare.py compare'
