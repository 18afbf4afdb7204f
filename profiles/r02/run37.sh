#!/bin/bash
# run 37: the driver-style invocation after moving the secondary measurements ahead of the headline region
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02aq
mkdir -p $OUT
cd $ROOT
for i in 1 2 3; do
  timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('driver-style', round(j['value'],1), 'ms/step', round(j['ms_per_step'],4), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec',0),1), 'cpu', j.get('cpu_baseline',{}).get('value'))"
done
timeout -k 10 300 python3 bench.py --no_cpu_baseline 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('default 400', round(j['value'],1))"
