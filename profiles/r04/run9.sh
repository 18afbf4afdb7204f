#!/bin/bash
# round 4, run 9: k_triplet with three waves per workgroup (every wave reads its own slot's lower-layer rows) vs four (a spare wave reads all)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
for v in default tw3; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  LGCN_SKIP_LARGE=1 LGCN_SKIP_LONG=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "triplet or fused_step or upstream or epochs_tiny or bitwise or hub" 2>&1 | tail -1
  for dt in fp32 bf16; do
    for i in 1 2; do
      timeout -k 10 300 python bench.py --act_dtype $dt --no_cpu_baseline --no_epochs --no_eval --no_secondary --no_steady 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v $dt', round(j['value'],1))" | tee -a gpurun_out/r04/triplet_waves_ab.txt
    done
  done
done
