#!/bin/bash
# run 42: last forward layer on the batch rows (k_triplet) vs densely, on the hub-heavy synthetic shapes, after k_triplet
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02av
mkdir -p $OUT
cd $ROOT
for wl in yelp2018-shaped amazon-book-shaped gowalla; do
  for dl in 0 1; do
    timeout -k 10 600 python3 bench.py --workload $wl --dense_last $dl --no_cpu_baseline 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$wl dense_last=$dl', round(j['value'],1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec',0),1))"
  done
done
