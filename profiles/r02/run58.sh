#!/bin/bash
# run 58: depth of the loss-term reduction (loads in flight of the one reducing wave) on the B = 8192 shape and on Gowalla
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bk
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
for v in base ul4 ul8 ul32; do
  if [ $v = base ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  for wl in yelp2018-shaped gowalla; do
    timeout -k 10 600 python3 bench.py --workload $wl --no_cpu_baseline --no_secondary --spmm_reps 200 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v $wl', round(j['value'],1))"
  done
done
done
