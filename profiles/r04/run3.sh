#!/bin/bash
# round 4, run 3: GPU parity suite (large tests last), Gowalla 1000 epochs with upstream's loss (--reg_rows ego), bench
mkdir -p gpurun_out/r04
LGCN_SKIP_LARGE=1 python -m pytest tests -m gpu -q > gpurun_out/r04/pytest_gpu_run3.txt 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r04/pytest_gpu_run3.txt
for a in fp32 bf16; do
python tools/gowalla_trajectory.py --epochs 1000 --act_dtype $a --prefetch_epoch 1 --reg_rows ego --quiet 1 --out gpurun_out/r04/gowalla_1000ep_ego_$a.json > gpurun_out/r04/traj_ego_$a.log 2>&1
tail -1 gpurun_out/r04/traj_ego_$a.log | cut -c1-600
done
python bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_run3.txt 2> gpurun_out/r04/bench_run3.err; echo "bench rc=$?"
tail -1 gpurun_out/r04/bench_run3.txt | cut -c1-2500
python -m pytest tests/test_gpu_large.py -m gpu -q -s > gpurun_out/r04/pytest_large_run3.txt 2>&1; echo "large rc=$?"; grep "c5 adjoint" gpurun_out/r04/pytest_large_run3.txt; tail -3 gpurun_out/r04/pytest_large_run3.txt
