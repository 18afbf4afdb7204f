#!/bin/bash
# round 4, run 37 -- the long-row chunks' descriptors (chunk record, row id, chunk count, bitmap word) through scalar loads too: parity suite, then
# steps/s and layer us against the previous build (lib_prev.so = the tree before this change), two rounds
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fp8.py -m gpu -q -x 2>&1 | tail -1 | tee gpurun_out/r04/scalar_chunks_parity.txt
for round in 1 2; do
for v in default prev; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  for spec in "gowalla bf16" "gowalla fp32" "amazon-book-shaped bf16"; do
    set -- $spec
    timeout -k 10 300 python bench.py --workload $1 --act_dtype $2 --no_cpu_baseline --no_epochs --no_eval --no_secondary --no_steady 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v $1 $2', round(j['value'],1), round(j['roofline']['avg_launch_us'],2))" | tee -a gpurun_out/r04/scalar_chunks_ab.txt
  done
done
done
