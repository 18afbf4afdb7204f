#!/bin/bash
# round 3, GPU run 31: sparse-input layer at 7 / 8 waves per SIMD (72 / 64 registers) instead of 6
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ac
mkdir -p $OUT
cd $ROOT
for v in default sp8 sp7 default sp8; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 300 python bench.py --no_cpu_baseline --no_eval --no_secondary 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('$v gowalla', o['value'])" | tee -a $OUT/bench.txt
done
