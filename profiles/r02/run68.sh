#!/bin/bash
# run 68 (experiment build): bf16 tables with 4 instead of 8 gathers in flight per lane (fewer registers, more waves)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
for v in base bu4; do
  if [ $v = base ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 300 python3 bench.py --act_dtype bf16 --no_cpu_baseline --no_secondary --spmm_reps 500 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v bf16 step', round(j['value'],1), 'dense layer us', round(j['roofline']['avg_launch_us'],2))"
done
done
