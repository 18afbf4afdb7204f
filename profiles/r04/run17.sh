#!/bin/bash
# round 4, run 17: tuning constants of the glue kernels -- 64-entry tiles per unit of a k_triplet slot row (ROWS_UNIT_TILES 1 / 2 / 4), gathers in
# flight per lane in the sparse-input layer (SPMM_U_SP 2 / 4 / 8)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
for v in default rut1 rut4 usp2 usp8; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "triplet or fused_step or epochs_tiny or hub" 2>&1 | tail -1
  for dt in fp32 bf16; do
    for i in 1 2; do
      timeout -k 10 300 python bench.py --act_dtype $dt --no_cpu_baseline --no_epochs --no_eval --no_secondary --no_steady 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v $dt', round(j['value'],1))" | tee -a gpurun_out/r04/glue_constants_ab.txt
    done
  done
done
