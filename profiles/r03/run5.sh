#!/bin/bash
# round 3, GPU run 5: hot-column plan with short-lived workgroups (HOT_IPW items per wave): A/B on the heavy-tailed shapes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03e
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hot_column" > $OUT/pytest_hot.log 2>&1; echo "hot rc=$?" | tee -a $OUT/status.log
tail -3 $OUT/pytest_hot.log
for v in default ipw1 ipw4 ipw2w16; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  for wl in yelp2018-shaped amazon-book-shaped; do
    for dt in fp32 bf16; do
      timeout -k 10 300 python bench.py --workload $wl --spmm_only --spmm_reps 300 --act_dtype $dt --hot_plan 1 2>/dev/null | tail -1 | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); r=o['roofline']; print('$v $wl $dt', r['kernel'], round(r['avg_launch_us'],2), 'us')" | tee -a $OUT/ab.txt
    done
  done
done
unset LGCN_LIB_PATH
for wl in yelp2018-shaped amazon-book-shaped; do
  for dt in fp32 bf16; do
    timeout -k 10 300 python bench.py --workload $wl --spmm_only --spmm_reps 300 --act_dtype $dt --hot_plan 0 2>/dev/null | tail -1 | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); r=o['roofline']; print('standard $wl $dt', r['kernel'], round(r['avg_launch_us'],2), 'us')" | tee -a $OUT/ab.txt
  done
done
