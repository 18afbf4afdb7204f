#!/bin/bash
# round 4, run 5: fp8 activation storage -- parity tests, Gowalla 10 epochs (Recall delta), bench on Gowalla and on C5
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_fp8.py tests/test_gpu_parity.py -m gpu -q -k "fp8 or upstream_loss or eval_topk" > gpurun_out/r04/pytest_run5.txt 2>&1; echo "rc=$?"; grep -n "^E \|^FAILED" gpurun_out/r04/pytest_run5.txt | cut -c1-300 | head -40; tail -3 gpurun_out/r04/pytest_run5.txt
timeout -k 10 300 python tools/gowalla_trajectory.py --epochs 10 --act_dtype fp8 --prefetch_epoch 1 --out gpurun_out/r04/gowalla_10ep_fp8.json > gpurun_out/r04/traj10_fp8.log 2>&1; tail -2 gpurun_out/r04/traj10_fp8.log | cut -c1-900
timeout -k 10 300 python bench.py --act_dtype fp8 --no_cpu_baseline --no_epochs --no_eval > gpurun_out/r04/bench_gowalla_fp8.txt 2> gpurun_out/r04/bench_gowalla_fp8.err; echo "bench rc=$?"; tail -1 gpurun_out/r04/bench_gowalla_fp8.txt | cut -c1-700
