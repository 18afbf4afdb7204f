#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
for wl in gowalla amazon-book-shaped; do
for dt in fp32 bf16; do
for v in default exp4 exp5; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 300 python bench.py --workload $wl --spmm_only --spmm_reps 2000 --act_dtype $dt 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$wl $dt $v', round(j['roofline']['avg_launch_us'],2))" | tee -a gpurun_out/r04/layer_decomposition2.txt
done
done
done
