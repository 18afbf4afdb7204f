#!/bin/bash
# round 4, run 10 -- THE MEASUREMENT PASS of the round on the final kernels: parity suite (all of it), smoke, PMC traffic of the dominant
# kernel per workload and storage type (separate --pmc passes: FETCH_SIZE / WRITE_SIZE / TCC hit+miss), kernel traces, bench lines.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04f
mkdir -p $OUT
cd $ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -4 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
cd /tmp && export TMPDIR=/tmp
for wl in gowalla yelp2018-shaped amazon-book-shaped; do        # (synthetic-10m: run12.sh)
  mkdir -p $OUT/pmc_$wl
  for dt in fp32 bf16 fp8; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$wl/pmc_fetch_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/f_$dt.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_$wl/pmc_write_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/w_$dt.log 2>&1
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_$wl/pmc_l2_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_$wl/l_$dt.log 2>&1
  done
  python3 $ROOT/profiles/pmc_traffic.py $OUT/pmc_$wl --write $wl --out $OUT/hbm_traffic.json | tee $OUT/pmc_${wl}_spmm.txt
  echo "pmc $wl done" | tee -a $OUT/status.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_gowalla/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $OUT/trace_gowalla.log 2>&1 || echo "trace failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT/trace_gowalla > $OUT/trace_gowalla_fp32_summary.txt 2>&1; head -18 $OUT/trace_gowalla_fp32_summary.txt | cut -c1-150
for spec in "yelp2018-shaped fp32" "amazon-book-shaped fp32" "amazon-book-shaped fp8"; do
  set -- $spec
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$1_$2/trace -- python3 $ROOT/bench.py --workload $1 --act_dtype $2 --steps 40 --warmup 5 --no_cpu_baseline --no_secondary --no_steady > $OUT/trace_$1_$2.log 2>&1 || echo "trace $1 $2 failed" | tee -a $OUT/status.log
  python3 $ROOT/profiles/summarize.py $OUT/trace_$1_$2 > $OUT/trace_$1_$2_summary.txt 2>&1; head -12 $OUT/trace_$1_$2_summary.txt | cut -c1-150
done
echo "traces done" | tee -a $OUT/status.log
