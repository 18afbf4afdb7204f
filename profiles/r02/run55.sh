#!/bin/bash
# run 55: HEAD sanity -- gpu suite, smoke, the driver's own bench invocation
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bh
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('driver-style', round(j['value'],1), 'traffic', j['roofline']['traffic'], 'frac', round(j['roofline']['frac'],3), 'cpu', j['cpu_baseline']['value'], j['cpu_baseline']['torch_eager']['value'], 'bf16', j['config']['bf16_activation_storage_steps_per_sec'])"
