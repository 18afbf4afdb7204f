#!/bin/bash
# round 4, run 23: a 16-deep gather batch (a 23-entry row in 1-2 round trips instead of 2-3) at 4 or 5 waves per SIMD vs. the default 8-deep at 6
# in flight per lane.  steps/s per workload and storage type, two runs each.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
timeout -k 10 600 env LGCN_LIB_PATH=$ROOT/build/variants/lib_u16w4.so python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "spmm or computer or fused_step or epochs_tiny or hub or boundary" 2>&1 | tail -1
for v in default u16w4 u16w5; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  for spec in "gowalla bf16" "gowalla fp32" "amazon-book-shaped bf16" "yelp2018-shaped bf16"; do
    set -- $spec
    for i in 1 2; do
      timeout -k 10 400 python bench.py --workload $1 --act_dtype $2 --no_cpu_baseline --no_epochs --no_eval --no_secondary --no_steady 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v $1 $2', round(j['value'],2), round(j['roofline']['avg_launch_us'],2))" | tee -a gpurun_out/r04/u16_ab.txt
    done
  done
done
