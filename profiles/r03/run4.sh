#!/bin/bash
# round 3, GPU run 4: hot-column plan (k_spmm_hot): parity, then A/B against the standard plan on the heavy-tailed shapes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03d
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hot_column" > $OUT/pytest_hot.log 2>&1; echo "hot rc=$?" | tee -a $OUT/status.log
tail -15 $OUT/pytest_hot.log
for wl in yelp2018-shaped amazon-book-shaped; do
  for hp in 1 0; do
    for dt in fp32 bf16; do
      timeout -k 10 300 python bench.py --workload $wl --spmm_only --spmm_reps 300 --act_dtype $dt --hot_plan $hp 2>/dev/null | tail -1 | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); r=o['roofline']; print('$wl hot=$hp $dt', r['kernel'], round(r['avg_launch_us'],2), 'us frac', round(r['frac'],4), 'H', r['hot_rows_in_lds'], 'cover', round(r['hot_gather_share'],3))" | tee -a $OUT/ab.txt
    done
  done
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity.log 2>&1; echo "parity rc=$?" | tee -a $OUT/status.log
tail -5 $OUT/pytest_parity.log
for wl in yelp2018-shaped amazon-book-shaped; do
  for hp in 1 0; do
    timeout -k 10 600 python bench.py --workload $wl --no_cpu_baseline --hot_plan $hp > $OUT/bench_${wl}_hot$hp.json 2> $OUT/bench_${wl}_hot$hp.err; echo "$wl hot=$hp rc=$?" | tee -a $OUT/status.log
    grep '^{"metric"' $OUT/bench_${wl}_hot$hp.json | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('$wl hot=$hp', o['value'], o['config'].get('bf16_activation_storage_steps_per_sec'), o['roofline']['avg_launch_us'])" | tee -a $OUT/ab.txt
  done
done
