#!/bin/bash
# round 4, run 27 -- k_triplet with three waves per workgroup (run 9) measured again now that bf16 storage is the headline: steps/s per
# workload and storage type, three runs each; first the data-parallel optional-branch tests on the rebuilt default library
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_two_procs.py -m gpu -q -x -k "optional_branches or two_proc" 2>&1 | tail -1
for v in default tw3 default tw3; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  for spec in "gowalla bf16" "gowalla fp32" "yelp2018-shaped bf16" "amazon-book-shaped bf16"; do
    set -- $spec
    timeout -k 10 400 python bench.py --workload $1 --act_dtype $2 --no_cpu_baseline --no_epochs --no_eval --no_secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v $1 $2', round(j['value'],1), round(j['steady_state_steps_per_sec'],1))" | tee -a gpurun_out/r04/tw3_ab.txt
  done
done
