#!/bin/bash
# round 3, GPU run 22: k_eval_topk with compact lists (16-bit relative ids, unpadded score rows): 3 workgroups per CU
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03v
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "eval or trajectory or procedure" > $OUT/pytest_eval.log 2>&1; echo "pytest rc=$?" | tee $OUT/status.log
tail -3 $OUT/pytest_eval.log
for v in default p2 p4; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 300 python tools/eval_time.py 2>/dev/null | tail -1 | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('$v', round(o['k_eval_topk']['ms'],3), 'ms', round(o['k_eval_topk']['frac'],3), 'Test', round(o['fused']['ms_per_Test'],2), 'ms', o['fused']['recall'], o['torch']['recall'])" | tee -a $OUT/eval_ab.txt
done
