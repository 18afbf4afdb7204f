#!/bin/bash
# round 4, run 2: GPU parity suite with the new tests, then Gowalla to 1000 epochs with upstream's loss (--reg_rows ego), then bench
set -e
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/pytest_gpu_run2.txt 2>&1 || { tail -40 gpurun_out/r04/pytest_gpu_run2.txt; exit 1; }
tail -5 gpurun_out/r04/pytest_gpu_run2.txt
for a in fp32 bf16; do
python tools/gowalla_trajectory.py --epochs 1000 --act_dtype $a --prefetch_epoch 1 --reg_rows ego --quiet 1 --out gpurun_out/r04/gowalla_1000ep_ego_$a.json > gpurun_out/r04/traj_ego_$a.log 2>&1
tail -1 gpurun_out/r04/traj_ego_$a.log
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_run2.txt 2> gpurun_out/r04/bench_run2.err
tail -1 gpurun_out/r04/bench_run2.txt
