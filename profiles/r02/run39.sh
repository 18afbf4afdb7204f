#!/bin/bash
# run 39: loss reduction 16 loads deep + k_scatter lane=column; tests, emulated data-parallel compute at world 1..8
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02as
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert|Mismatch" $OUT/pytest.log | head -30; exit 1; }
timeout -k 10 600 python3 tools/dp_emulate_time.py 2> $OUT/err.log | tail -1 | tee $OUT/dp_emulate.json
timeout -k 10 300 python3 bench.py --no_cpu_baseline 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('default 400', round(j['value'],1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec',0),1))"
