#!/bin/bash
# round 3, GPU run 8a: parity suite, smoke, PMC traffic of the dominant kernel (separate --pmc passes), kernel traces
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03h
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -4 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
cd /tmp && export TMPDIR=/tmp
for wl in gowalla yelp2018-shaped amazon-book-shaped synthetic-10m; do
  mkdir -p $OUT/pmc_$wl
  for dt in fp32 bf16; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$wl/pmc_fetch_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/f_$dt.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_$wl/pmc_write_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/w_$dt.log 2>&1
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_$wl/pmc_l2_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/l_$dt.log 2>&1
  done
  python3 $ROOT/profiles/pmc_traffic.py $OUT/pmc_$wl --write $wl --out $OUT/hbm_traffic.json | tee $OUT/pmc_${wl}_spmm.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_gowalla/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $OUT/trace_gowalla.log 2>&1 || echo "trace failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT/trace_gowalla > $OUT/trace_gowalla_fp32_summary.txt 2>&1; head -16 $OUT/trace_gowalla_fp32_summary.txt | cut -c1-140
for wl in yelp2018-shaped amazon-book-shaped; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl/trace -- python3 $ROOT/bench.py --workload $wl --steps 40 --warmup 5 --no_cpu_baseline > $OUT/trace_$wl.log 2>&1 || echo "trace $wl failed" | tee -a $OUT/status.log
  python3 $ROOT/profiles/summarize.py $OUT/trace_$wl > $OUT/trace_${wl}_fp32_summary.txt 2>&1; head -12 $OUT/trace_${wl}_fp32_summary.txt | cut -c1-140
done
