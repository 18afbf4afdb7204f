#!/bin/bash
# round 4, run 7: the whole GPU suite (new: column-sharded DP, fp8, upstream loss, eval K <= 64 / masks), then the long test, then 1000 epochs in fp8
mkdir -p gpurun_out/r04
LGCN_SKIP_LARGE=1 LGCN_SKIP_LONG=1 python -m pytest tests -m gpu -q > gpurun_out/r04/pytest_run7.txt 2>&1; echo "suite rc=$?"; grep -n "^E \|^FAILED" gpurun_out/r04/pytest_run7.txt | cut -c1-300 | head -30; tail -3 gpurun_out/r04/pytest_run7.txt
timeout -k 10 300 python tools/gowalla_trajectory.py --epochs 1000 --act_dtype fp8 --prefetch_epoch 1 --reg_rows ego --quiet 1 --out gpurun_out/r04/gowalla_1000ep_ego_fp8.json > gpurun_out/r04/traj_ego_fp8.log 2>&1; tail -1 gpurun_out/r04/traj_ego_fp8.log | cut -c1-700
python -m pytest tests/test_gpu_long.py -m gpu -q -s > gpurun_out/r04/pytest_long_run7.txt 2>&1; echo "long rc=$?"; grep "1000 epochs" gpurun_out/r04/pytest_long_run7.txt; tail -2 gpurun_out/r04/pytest_long_run7.txt
