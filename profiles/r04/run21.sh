#!/bin/bash
# round 4, run 21: bf16 tables with the shared Adam epilogue -- Adam's operands prefetched under the pack's last gather batch
# (SPMM_ADAM_PREFETCH_BF16 1) vs. fetched in the epilogue (0, default)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
for v in default prebf16; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bf16" 2>&1 | tail -1
  for w in gowalla amazon-book-shaped synthetic-10m; do
    for i in 1 2; do
      timeout -k 10 400 python bench.py --workload $w --act_dtype bf16 --no_cpu_baseline --no_epochs --no_eval --no_secondary --no_steady 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v $w bf16', round(j['value'],2))" | tee -a gpurun_out/r04/prebf16_ab.txt
    done
  done
done
