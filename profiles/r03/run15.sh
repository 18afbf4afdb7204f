#!/bin/bash
# round 3, GPU run 15: per-epoch table of the triplet rows' CSR extents (k_slot_info): parity suite, Gowalla bench + k_triplet time
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03p
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity.log 2>&1; echo "parity rc=$?" | tee -a $OUT/status.log
tail -3 $OUT/pytest_parity.log | cut -c1-300
for i in 1 2; do
timeout -k 10 300 python bench.py --no_cpu_baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('gowalla fp32', round(o['value']), 'bf16', round(o['config']['bf16_activation_storage_steps_per_sec']))" | tee -a $OUT/ab.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_gowalla/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $OUT/trace_gowalla.log 2>&1
python3 $ROOT/profiles/summarize.py $OUT/trace_gowalla 2>&1 | grep -E "k_triplet|k_g32|k_slot" | cut -c1-140
cd $ROOT
timeout -k 10 600 python bench.py --workload synthetic-10m --no_cpu_baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('c5 fp32', round(o['value'],3))" | tee -a $OUT/ab.txt
