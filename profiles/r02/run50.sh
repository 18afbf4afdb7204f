#!/bin/bash
# run 50: bf16 copy of the parameter table as the input of layer 1 in bf16 activation mode; tests, bench, trajectory in bf16
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bc
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert|Mismatch|Max abs|Max rel" $OUT/pytest.log | head -30; }
for i in 1 2; do
timeout -k 10 300 python3 bench.py --no_cpu_baseline 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('default 400', round(j['value'],1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec',0),1))"
done
timeout -k 10 600 python3 tools/gowalla_trajectory.py --act_dtype bf16 2>> $OUT/err.log | python3 -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
ep=[r for r in rows if 'epoch' in r and 'seconds' in r]
print('bf16 trajectory: max |recall diff|', max(r['abs_diff']['recall'] for r in ep if 'abs_diff' in r), 'max |ndcg diff|', max(r['abs_diff']['ndcg'] for r in ep if 'abs_diff' in r), 'epoch ms', round(ep[-1]['seconds']*1e3,1))
"
