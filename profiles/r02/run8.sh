#!/bin/bash
# round 2, GPU run 8: round-2 measurements (bench lines, rocprofv3 traces, PMC traffic, DP at world 1, eval timing)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02h
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python bench.py > $OUT/bench_gowalla.json 2> $OUT/bench_gowalla.err; echo "bench rc=$?" | tee -a $OUT/status.log
cat $OUT/bench_gowalla.json
for mode in "--dp_reduce rows" "--dp_reduce dense" "--dp_shard rows"; do
  timeout -k 10 300 python bench.py --force_dp $mode --no_cpu_baseline --steps 200 2>> $OUT/dp.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dp1', '$mode', round(j['value'],1), j['config']['parallelism'][:60], j['config']['last_loss'])" | tee -a $OUT/dp.log
done
timeout -k 10 300 python tools/eval_time.py 2>> $OUT/eval.err | tee $OUT/eval_time.json
for wl in yelp2018-shaped amazon-book-shaped synthetic-10m; do
  timeout -k 10 600 python bench.py --workload $wl --no_cpu_baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "bench $wl rc=$?" | tee -a $OUT/status.log
  python -c "import sys,json; j=json.loads(open('$OUT/bench_$wl.json').read()); print(j['config']['workload'][:40], round(j['value'],2), 'steps/s', 'step frac', round(j['step_roofline_frac'],3), 'spmm us', round(j['roofline']['avg_launch_us'],1), 'frac', round(j['roofline']['frac'],3))"
done
cd /tmp && export TMPDIR=/tmp
for wl in gowalla yelp2018-shaped amazon-book-shaped; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl/trace -- python3 $ROOT/bench.py --workload $wl --steps 100 --warmup 10 --no_cpu_baseline > $OUT/trace_$wl.log 2>&1 || echo "trace $wl failed" | tee -a $OUT/status.log
  python3 $ROOT/profiles/summarize.py $OUT/trace_$wl > $OUT/trace_${wl}_summary.txt 2>&1; head -14 $OUT/trace_${wl}_summary.txt | cut -c1-140
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5/trace -- python3 $ROOT/bench.py --workload synthetic-10m --steps 5 --warmup 1 --no_cpu_baseline > $OUT/trace_c5.log 2>&1 || echo "trace c5 failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT/trace_c5 > $OUT/trace_c5_summary.txt 2>&1; head -12 $OUT/trace_c5_summary.txt | cut -c1-140
for wl in gowalla yelp2018-shaped amazon-book-shaped; do
  mkdir -p $OUT/pmc_$wl
  for dt in fp32 bf16; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$wl/pmc_fetch_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/f_$dt.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_$wl/pmc_write_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/w_$dt.log 2>&1
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_$wl/pmc_l2_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/l_$dt.log 2>&1
  done
  python3 $ROOT/profiles/pmc_traffic.py $OUT/pmc_$wl --write $wl --out $OUT/hbm_traffic.json | tee $OUT/pmc_${wl}_summary.txt
done
