#!/bin/bash
# round 3, GPU run 32: k_eval_topk, the parts of a user's sweep exchange their K-th best (every 16 / 4 tiles / never)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ad
mkdir -p $OUT
cd $ROOT
for v in default nopub pub4; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -x -k "eval" 2>&1 | tail -1
  timeout -k 10 300 python tools/eval_time.py 2>/dev/null | tail -1 | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('$v', round(o['k_eval_topk']['ms'],3), 'ms', round(o['k_eval_topk']['frac'],3), 'Test', round(o['fused']['ms_per_Test'],2), 'ms', o['fused']['recall'], o['torch']['recall'])" | tee -a $OUT/eval_ab.txt
  timeout -k 10 300 python tools/eval_shapes.py 2>/dev/null | tail -1 | tee -a $OUT/eval_shapes_$v.json | cut -c1-420
done
