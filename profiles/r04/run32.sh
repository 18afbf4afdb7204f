#!/bin/bash
# round 4, run 32 -- the dispatch floor of run 31 (5.9 us for ~19 000 one-wave workgroups) against residency: two / four waves per workgroup, or
# two packs per wave, with bf16 tables (round 2 measured one-wave workgroups best with fp32 tables).  steps/s over 400 steps, two rounds.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
export LGCN_LIB_PATH=$ROOT/build/variants/lib_wpb2.so
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "spmm or computer or fused_step or epochs_tiny" 2>&1 | tail -1
for round in 1 2; do
for v in default wpb2 wpb4 wr8; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  for spec in "gowalla bf16" "gowalla fp32" "amazon-book-shaped bf16"; do
    set -- $spec
    timeout -k 10 300 python bench.py --workload $1 --act_dtype $2 --no_cpu_baseline --no_epochs --no_eval --no_secondary --no_steady 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v $1 $2', round(j['value'],1), round(j['roofline']['avg_launch_us'],2))" | tee -a gpurun_out/r04/wpb_ab.txt
  done
done
done
