#!/bin/bash
# round 4, run 16: Adam's P/M/V operands prefetched under the last gather batch with a bf16 table too (needs a wider register budget)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
for v in base w5 pre5 pre4; do
  export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bf16" 2>&1 | tail -1
  for i in 1 2 3; do
    timeout -k 10 300 python bench.py --act_dtype bf16 --no_cpu_baseline --no_epochs --no_eval --no_secondary --no_steady 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v bf16', round(j['value'],1))" | tee -a gpurun_out/r04/adam_prefetch_bf16_ab.txt
  done
done
