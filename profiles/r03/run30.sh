#!/bin/bash
# round 3, GPU run 30: sparse-input layer with several packs per wave and their row-info / index / bitmap loads hoisted
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ab
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/status.log
tail -3 $OUT/pytest.log
for i in 1 2; do
timeout -k 10 300 python bench.py --no_cpu_baseline --no_eval 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('gowalla', o['value'], o['config'].get('bf16_activation_storage_steps_per_sec'), o['roofline']['avg_launch_us'])" | tee -a $OUT/bench.txt
done
timeout -k 10 300 python bench.py --workload yelp2018-shaped --no_cpu_baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('yelp', o['value'], o['config'].get('bf16_activation_storage_steps_per_sec'))" | tee -a $OUT/bench.txt
timeout -k 10 300 python bench.py --workload amazon-book-shaped --no_cpu_baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('amazon', o['value'], o['config'].get('bf16_activation_storage_steps_per_sec'))" | tee -a $OUT/bench.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_gowalla/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline --no_eval > $OUT/trace_gowalla.log 2>&1 || echo "trace failed"
python3 $ROOT/profiles/summarize.py $OUT/trace_gowalla > $OUT/trace_gowalla_fp32_summary.txt 2>&1; head -14 $OUT/trace_gowalla_fp32_summary.txt | cut -c1-140
