#!/bin/bash
# round 4, run 20: bf16 tables -- the two lane groups of a row share its Adam epilogue (SPMM_ADAM_SPLIT 1, default) vs. one group
# doing all 8 columns per lane (0).  Parity tests first, then A/B on every workload in bf16.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2 | tee gpurun_out/r04/split_parity.txt
for v in default nosplit; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  for w in gowalla yelp2018-shaped amazon-book-shaped synthetic-10m; do
    for i in 1 2; do
      timeout -k 10 400 python bench.py --workload $w --act_dtype bf16 --no_cpu_baseline --no_epochs --no_eval --no_secondary --no_steady 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v $w bf16', round(j['value'],2), j['roofline'].get('avg_launch_us'))" | tee -a gpurun_out/r04/split_ab.txt
    done
  done
done
