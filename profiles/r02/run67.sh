#!/bin/bash
# round 2, GPU run 67: final pass -- parity suite, smoke, PMC traffic, traces and bench lines of every workload
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bv
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -4 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for wl in gowalla yelp2018-shaped amazon-book-shaped; do
  mkdir -p $OUT/pmc_$wl
  for dt in fp32 bf16; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$wl/pmc_fetch_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/f_$dt.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_$wl/pmc_write_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/w_$dt.log 2>&1
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_$wl/pmc_l2_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/l_$dt.log 2>&1
  done
  python3 $ROOT/profiles/pmc_traffic.py $OUT/pmc_$wl --write $wl --out $OUT/hbm_traffic.json | tee $OUT/pmc_${wl}_summary.txt
done
cp $OUT/hbm_traffic.json $ROOT/profiles/hbm_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_gowalla/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $OUT/trace_gowalla.log 2>&1 || echo "trace failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT/trace_gowalla > $OUT/trace_gowalla_summary.txt 2>&1; head -16 $OUT/trace_gowalla_summary.txt | cut -c1-140
cd $ROOT
timeout -k 10 600 python bench.py > $OUT/bench_gowalla.json 2> $OUT/bench_gowalla.err; echo "bench rc=$?" | tee -a $OUT/status.log
grep '^{"metric"' $OUT/bench_gowalla.json
timeout -k 10 600 python bench.py --workload synthetic-10m --no_cpu_baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?" | tee -a $OUT/status.log
grep '^{"metric"' $OUT/bench_c5.json | cut -c1-400
for wl in yelp2018-shaped amazon-book-shaped; do
  timeout -k 10 600 python bench.py --workload $wl --no_cpu_baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "$wl rc=$?" | tee -a $OUT/status.log
  grep '^{"metric"' $OUT/bench_$wl.json | cut -c1-300
done
cd /tmp
for wl in yelp2018-shaped amazon-book-shaped; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl/trace -- python3 $ROOT/bench.py --workload $wl --steps 40 --warmup 5 --no_cpu_baseline > $OUT/trace_$wl.log 2>&1 || echo "trace $wl failed" | tee -a $OUT/status.log
  python3 $ROOT/profiles/summarize.py $OUT/trace_$wl > $OUT/trace_${wl}_summary.txt 2>&1; head -14 $OUT/trace_${wl}_summary.txt | cut -c1-140
done
cd $ROOT
timeout -k 10 300 python tools/eval_time.py 2>> $OUT/eval.err | tail -1 | tee $OUT/eval_time.json | cut -c1-400
