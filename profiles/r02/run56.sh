#!/bin/bash
# run 56: dense_last path as ONE launch (k_triplet_dense replaces k_rows_dense + k_bpr_loss): tests, the three shapes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bi
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert|Mismatch|Max abs|Max rel" $OUT/pytest.log | head -30; exit 1; }
for wl in yelp2018-shaped amazon-book-shaped gowalla; do
  timeout -k 10 600 python3 bench.py --workload $wl --no_cpu_baseline 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$wl', round(j['value'],1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec',0),1))"
done
