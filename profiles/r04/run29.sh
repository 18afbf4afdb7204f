#!/bin/bash
# round 4, run 29 -- where the time of the dense layer goes (experiment builds made from a temporary patch of k_spmm's short-row path, not in the
# tree): exp1 = no gathers (row descriptors + index tiles + staging + store), exp2 = row descriptors + store only, exp3 = every gather reads rows 0 / 1
# (all cache hits: the instruction path without memory latency).  Average launch us of the layer, Gowalla and the Amazon shape, fp32 / bf16.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
for wl in gowalla amazon-book-shaped; do
for dt in fp32 bf16; do
for v in default exp1 exp2 exp3; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  timeout -k 10 300 python bench.py --workload $wl --spmm_only --spmm_reps 2000 --act_dtype $dt 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$wl $dt $v', {k:(round(v,2) if isinstance(v,float) else v) for k,v in j.items() if k in ('avg_launch_us','t_spmm_us','us')} or j)" | tee -a gpurun_out/r04/layer_decomposition.txt
done
done
done
