#!/bin/bash
# round 4, run 33 -- the same question on the 10M x 1M graph (2.7 M packs: the dispatch floor alone is ~0.9 ms of a 17-50 ms layer): waves per workgroup
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
for v in default wpb4 wpb2; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  for dt in fp8 bf16; do
    timeout -k 10 500 python bench.py --workload synthetic-10m --act_dtype $dt --no_cpu_baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v c5 $dt', round(j['value'],3), round(j['roofline']['avg_launch_us'],1))" | tee -a gpurun_out/r04/wpb_c5.txt
  done
done
