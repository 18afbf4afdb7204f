#!/bin/bash
# round 3, GPU run 11: what k_eval_topk's 2.5 ms are made of (experiment builds: no insertion after the first tiles, no marking, no MFMA)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03l
mkdir -p $OUT
cd $ROOT
for v in default noinsert nomark; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 300 python tools/eval_time.py 2>/dev/null | tail -1 | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('$v', round(o['k_eval_topk']['ms'],3), 'ms', round(o['k_eval_topk']['frac'],3), 'Test', round(o['fused']['ms_per_Test'],2), 'ms')" | tee -a $OUT/eval_ab.txt
done
