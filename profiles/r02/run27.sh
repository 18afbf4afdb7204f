#!/bin/bash
# run 27: data-parallel path at world 1 through RCCL after k_g32 / k_triplet (C epoch loop, three modes; per-step torch loop), eval timing
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02ae
mkdir -p $OUT
cd $ROOT
i=0
for mode in "--dp_reduce rows" "--dp_reduce dense" "--dp_shard rows"; do
  i=$((i+1))
  timeout -k 10 300 python bench.py --force_dp $mode --no_cpu_baseline --steps 200 > $OUT/dp_$i.out 2>> $OUT/dp.err; echo "dp $mode rc=$?" | tee -a $OUT/status.log
  grep '^{"metric"' $OUT/dp_$i.out | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dp1', '$mode', round(j['value'],1), j['config']['parallelism'][:70], j['config']['last_loss'])" | tee -a $OUT/dp.log
done
LGCN_DP_PYTHON_LOOP=1 timeout -k 10 300 python bench.py --force_dp --no_cpu_baseline --steps 200 > $OUT/dp_py.out 2>> $OUT/dp.err; echo "dp python loop rc=$?" | tee -a $OUT/status.log
grep '^{"metric"' $OUT/dp_py.out | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dp1 per-step torch collectives', round(j['value'],1), j['config']['last_loss'])" | tee -a $OUT/dp.log
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --force_dp --no_cpu_baseline --steps 200 > $OUT/dp_tr.out 2>> $OUT/dp.err; echo "torchrun rc=$?" | tee -a $OUT/status.log
grep '^{"metric"' $OUT/dp_tr.out | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dp1 under torch.distributed.run', round(j['value'],1), j['config']['last_loss'])" | tee -a $OUT/dp.log
timeout -k 10 300 python bench.py --no_cpu_baseline --steps 200 2>> $OUT/dp.err | grep '^{"metric"' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('single', round(j['value'],1), j['config']['last_loss'])" | tee -a $OUT/dp.log
timeout -k 10 300 python tools/eval_time.py 2>> $OUT/eval.err | tail -1 | tee $OUT/eval_time.json | cut -c1-500
tail -5 $OUT/dp.err
