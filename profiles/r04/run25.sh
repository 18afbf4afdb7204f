#!/bin/bash
# round 4, run 25 -- the N > 1 command with the new default storage type (bf16 on gowalla): real RCCL at world size 1 in the four exchange
# modes, and the driver's N = 2 / 4 command end to end on one GPU (LGCN_BENCH_ONE_GPU=1: gloo, per-step loop; a functional rehearsal)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04j
mkdir -p $OUT
cd $ROOT
for mode in "--dp_reduce rows" "--dp_reduce dense" "--dp_shard rows" "--dp_shard cols"; do
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python bench.py --force_dp $mode --no_cpu_baseline --no_secondary --no_epochs --no_eval 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('world-1 RCCL $mode', j['dtype'][:4], round(j['value'],1), j['rccl_ranks_observed'], j['rccl_ranks_source'], j['config']['last_loss'])" | tee -a $OUT/dp_world1_rccl_bf16.txt
done
for n in 2 4; do
  LGCN_BENCH_ONE_GPU=1 timeout -k 10 500 python bench.py --gpus $n --steps 20 --warmup 5 2>$OUT/rehearsal_$n.err | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('rehearsal n=$n', j['dtype'][:4], round(j['value'],1), j['n_gpus'], j['rccl_ranks_observed'], j['rccl_ranks_source'], j['config']['multi_gpu_status'][:40])" | tee -a $OUT/rehearsal_one_gpu_bf16.txt
done
