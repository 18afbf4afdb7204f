#!/bin/bash
# round 2, GPU run 1: parity suite on the new kernels + SpMM sweep over row orders / dtypes + PMC traffic
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02a
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -5 $OUT/pytest.log
for wl in gowalla yelp2018-shaped amazon-book-shaped; do
  for ro in natural cocluster xcd; do
    for dt in fp32 bf16; do
      timeout -k 10 300 python bench.py --workload $wl --row_order $ro --act_dtype $dt --spmm_only >> $OUT/spmm_sweep.jsonl 2>> $OUT/spmm_sweep.err || echo "sweep $wl $ro $dt failed" | tee -a $OUT/status.log
    done
  done
done
cat $OUT/spmm_sweep.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); r = j['roofline']
    print(j['workload'], j['row_order'], j['act_dtype'], 'us', round(r['avg_launch_us'],2), 'frac', round(r['frac'],3))
"
timeout -k 10 600 python bench.py > $OUT/bench_gowalla.json 2> $OUT/bench_gowalla.err; echo "bench rc=$?" | tee -a $OUT/status.log
cat $OUT/bench_gowalla.json
cd /tmp && export TMPDIR=/tmp
for dt in fp32 bf16; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$dt -- python3 $ROOT/bench.py --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc1_$dt.log 2>&1 || echo "pmc fetch $dt failed" | tee -a $OUT/status.log
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$dt -- python3 $ROOT/bench.py --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc2_$dt.log 2>&1 || echo "pmc write $dt failed" | tee -a $OUT/status.log
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2_$dt -- python3 $ROOT/bench.py --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc3_$dt.log 2>&1 || echo "pmc l2 $dt failed" | tee -a $OUT/status.log
done
python3 $ROOT/profiles/pmc_traffic.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
