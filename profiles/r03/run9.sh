#!/bin/bash
# round 3, GPU run 9: register budget / lane groups per bf16 row / gather depth after the inner-loop change (variants)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03j
mkdir -p $OUT
cd $ROOT
for v in default w7 gpr1 w7gpr1 u4; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  for dt in fp32 bf16; do
    timeout -k 10 300 python bench.py --spmm_only --spmm_reps 2000 --act_dtype $dt 2>/dev/null | tail -1 | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); r=o['roofline']; print('$v spmm $dt', round(r['avg_launch_us'],2), 'us')" | tee -a $OUT/ab.txt
  done
  timeout -k 10 300 python bench.py --no_cpu_baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('$v step fp32', round(o['value']), 'bf16', round(o['config']['bf16_activation_storage_steps_per_sec']))" | tee -a $OUT/ab.txt
done
