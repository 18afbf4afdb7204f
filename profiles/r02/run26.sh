#!/bin/bash
# run 26: k_eval_topk item sweep split over 2 (3, 4) workgroups per user block
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02ad
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "eval or trajectory" > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert" $OUT/pytest.log | head -20; exit 1; }
timeout -k 10 300 python tools/eval_time.py 2>$OUT/eval.err | tail -1 | tee $OUT/eval_time.json
for v in p3 p4; do LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so timeout -k 10 300 python tools/eval_time.py 2>>$OUT/eval.err | tail -1 | cut -c1-400; done
