#!/bin/bash
# round 4, run 18: REHEARSAL of the driver's N > 1 command on one GPU (LGCN_BENCH_ONE_GPU=1: all ranks on GPU 0, gloo, per-step loop): the script's
# whole multi-rank path -- self-launch, rendezvous, sharded batches / column shards, barrier + max-over-ranks timing, rank 0's JSON line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04
cd $ROOT
export LGCN_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for spec in "2 --dp_reduce rows" "2 --dp_reduce dense" "2 --dp_shard cols" "4 --dp_reduce rows" "2 --dp_reduce rows --scaling weak"; do
  set -- $spec; n=$1; shift
  timeout -k 10 300 python bench.py --gpus $n --steps 20 --warmup 5 --no_cpu_baseline "$@" 2>gpurun_out/r04/rehearsal_$n.err | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('gpus $n $*:', 'value', round(j['value'],1), 'n_gpus', j['n_gpus'], 'scaling', j['scaling'], 'ranks_observed', j['rccl_ranks_observed'], j['rccl_ranks_source'], 'steady', j['steady_state_steps_per_sec'] and round(j['steady_state_steps_per_sec'],1), 'last_loss', j['config']['last_loss'], '|', j['config']['parallelism'][:60])" | tee -a gpurun_out/r04/rehearsal_one_gpu.txt
done
