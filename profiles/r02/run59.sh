#!/bin/bash
# run 59: loss reduction: plain loop on one GPU, 16-deep division-free batches over the ranks' blocks in data parallel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bl
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert|Mismatch|Max abs|Max rel" $OUT/pytest.log | head -30; exit 1; }
for rep in 1 2; do
  for wl in yelp2018-shaped gowalla; do
    timeout -k 10 600 python3 bench.py --workload $wl --no_cpu_baseline --no_secondary --spmm_reps 200 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$wl', round(j['value'],1))"
  done
done
timeout -k 10 600 python3 tools/dp_emulate_time.py 2>> $OUT/err.log | tail -1 | tee $OUT/dp_emulate.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print({k: (round(v['us_per_step'],1), round(v['compute_only_weak_scaling_efficiency'],3)) for k,v in j.items()})"
