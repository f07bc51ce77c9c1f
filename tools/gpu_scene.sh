#!/bin/bash
# Offline convergence + loop-throughput runs on synthetic scenes.  Run on the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/gpu_scene.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/scene; mkdir -p $O
S=/tmp/scenes
python tools/scene_writer.py $S --scene small --frames 16 --size 120 160 --seed 0 > $O/write.log 2>&1
python tools/scene_writer.py $S --scene full --frames 16 --size 540 960 --seed 0 >> $O/write.log 2>&1
# convergence: joint pose + field training from identity poses, both rendering styles of the reference's configs
timeout 400 python tools/train_scene.py $S small --style tanks --epochs 150 --log-every 10 --eval-epochs 100 --out $O/conv_tanks.json > $O/conv_tanks.log 2>&1
timeout 400 python tools/train_scene.py $S small --style llff --epochs 150 --log-every 10 --eval-epochs 100 --out $O/conv_llff.json > $O/conv_llff.log 2>&1
# loop throughput at Tanks resolution (540x960 frames, 384x672-like depth maps are written at frame size here): host loader vs resident
timeout 300 python tools/train_scene.py $S full --epochs 12 --log-every 100 --out $O/loop_resident.json > $O/loop_resident.log 2>&1
timeout 300 python tools/train_scene.py $S full --epochs 12 --log-every 100 --host-loader --out $O/loop_host.json > $O/loop_host.log 2>&1
timeout 300 python tools/train_scene.py $S full --epochs 12 --log-every 100 --no-aux --out $O/loop_resident_noaux.json > $O/loop_resident_noaux.log 2>&1
timeout 300 python tools/train_scene.py $S full --epochs 12 --log-every 100 --no-aux --host-loader --out $O/loop_host_noaux.json > $O/loop_host_noaux.log 2>&1
tail -n 2 $O/*.log
