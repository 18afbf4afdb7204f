#!/bin/bash
# round 3, GPU run 18: pack records (one 1-KiB load per pack of 4 short rows instead of a plan load + two stream loads, d = 64): parity, bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03s
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity.log 2>&1; echo "parity rc=$?" | tee -a $OUT/status.log
tail -3 $OUT/pytest_parity.log | cut -c1-300
for dt in fp32 bf16; do
  timeout -k 10 300 python bench.py --spmm_only --spmm_reps 2000 --act_dtype $dt 2>/dev/null | tail -1 | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); r=o['roofline']; print('gowalla spmm $dt', round(r['avg_launch_us'],2), 'us')" | tee -a $OUT/ab.txt
done
for i in 1 2; do
timeout -k 10 300 python bench.py --no_cpu_baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('gowalla fp32', round(o['value']), 'bf16', round(o['config']['bf16_activation_storage_steps_per_sec']))" | tee -a $OUT/ab.txt
done
timeout -k 10 300 python bench.py --workload yelp2018-shaped --no_cpu_baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('yelp fp32', round(o['value']), 'bf16', round(o['config']['bf16_activation_storage_steps_per_sec']), round(o['roofline']['avg_launch_us'],2))" | tee -a $OUT/ab.txt
