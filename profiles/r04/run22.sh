#!/bin/bash
# round 4, run 22: residency of the bf16 / fp32 SpMM variants -- register budget for 7 or 8 waves per SIMD (default 6), with 8 or 4 gathers
# in flight per lane.  steps/s per workload and storage type, two runs each.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
for v in default w7 w8u4 w8; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  for spec in "gowalla bf16" "gowalla fp32" "amazon-book-shaped bf16" "yelp2018-shaped bf16"; do
    set -- $spec
    for i in 1 2; do
      timeout -k 10 400 python bench.py --workload $1 --act_dtype $2 --no_cpu_baseline --no_epochs --no_eval --no_secondary --no_steady 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v $1 $2', round(j['value'],2), round(j['roofline']['avg_launch_us'],2))" | tee -a gpurun_out/r04/waves_ab.txt
    done
  done
done
