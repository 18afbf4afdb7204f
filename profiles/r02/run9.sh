#!/bin/bash
# round 2, GPU run 9: dense-last option, eval kernel v2, data-parallel path at world 1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02i
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -6 $OUT/pytest.log
for wl in gowalla yelp2018-shaped amazon-book-shaped; do
  timeout -k 10 600 python bench.py --workload $wl --no_cpu_baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "bench $wl rc=$?" | tee -a $OUT/status.log
  grep '^{"metric"' $OUT/bench_$wl.json | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['config']['workload'][:40], round(j['value'],2), 'steps/s', 'bf16', j['config'].get('bf16_activation_storage_steps_per_sec'), 'step frac', round(j['step_roofline_frac'],3), 'spmm us', round(j['roofline']['avg_launch_us'],1), 'traffic', j['roofline']['traffic'])"
done
i=0
for mode in "--dp_reduce rows" "--dp_reduce dense" "--dp_shard rows"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --force_dp $mode --no_cpu_baseline --steps 200 > $OUT/dp_$i.out 2>> $OUT/dp.err; echo "dp $mode rc=$?" | tee -a $OUT/status.log
  grep '^{"metric"' $OUT/dp_$i.out | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dp1', '$mode', round(j['value'],1), j['config']['parallelism'][:70], j['config']['last_loss'])" | tee -a $OUT/dp.log
done
timeout -k 10 300 python tools/eval_time.py 2>> $OUT/eval.err | tee $OUT/eval_time.json
cd /tmp && export TMPDIR=/tmp
for wl in yelp2018-shaped amazon-book-shaped; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl/trace -- python3 $ROOT/bench.py --workload $wl --steps 100 --warmup 10 --no_cpu_baseline > $OUT/trace_$wl.log 2>&1 || echo "trace $wl failed" | tee -a $OUT/status.log
  python3 $ROOT/profiles/summarize.py $OUT/trace_$wl > $OUT/trace_${wl}_summary.txt 2>&1; head -12 $OUT/trace_${wl}_summary.txt | cut -c1-140
done
