#!/bin/bash
# round 4, run 28 -- the fp32-era tuning constants re-checked with bf16 tables (the headline storage type): lane groups per bf16 row, chunk
# length of long rows, window of the length sort, gathers in flight in k_triplet.  Gowalla bf16 steps/s (400 steps | steady), two rounds.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
for round in 1 2; do
for v in default gpr1 gpr4 lch256 lch1024 win1024 win4096 tu4 tu16; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 300 python bench.py --act_dtype bf16 --no_cpu_baseline --no_epochs --no_eval --no_secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v bf16', round(j['value'],1), round(j['steady_state_steps_per_sec'],1), j['config']['last_loss'])" | tee -a gpurun_out/r04/bf16_constants_ab.txt
done
done
