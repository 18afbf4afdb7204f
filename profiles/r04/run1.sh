#!/bin/bash
# round 4, run 1: Gowalla to the reference's 1000-epoch horizon, fp32 and bf16 activation storage
set -e
mkdir -p gpurun_out/r04
python tools/gowalla_trajectory.py --epochs 1000 --act_dtype fp32 --prefetch_epoch 1 --out gpurun_out/r04/gowalla_1000ep_fp32.json > gpurun_out/r04/traj_fp32.log 2>&1
tail -1 gpurun_out/r04/traj_fp32.log
python tools/gowalla_trajectory.py --epochs 1000 --act_dtype bf16 --prefetch_epoch 1 --out gpurun_out/r04/gowalla_1000ep_bf16.json > gpurun_out/r04/traj_bf16.log 2>&1
tail -1 gpurun_out/r04/traj_bf16.log
