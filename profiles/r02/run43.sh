#!/bin/bash
# run 43: 10 Gowalla epochs end to end (sampling + shuffle + training + Test at 1/2/5/10), next epoch prefetched or not
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02aw
mkdir -p $OUT
cd $ROOT
for pf in 0 1; do
  timeout -k 10 600 python3 tools/gowalla_trajectory.py --prefetch_epoch $pf --out $OUT/traj_pf$pf.json 2> $OUT/err$pf.log | python3 -c "
import sys, json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
tot=rows[-1]; ep=[r for r in rows if 'epoch' in r and 'seconds' in r]
print('prefetch $pf: total train s', round(tot['train_seconds_total'],3), 'per epoch ms', [round(r['seconds']*1e3,1) for r in ep])
print('   max |recall diff|', max(r['abs_diff']['recall'] for r in ep if 'abs_diff' in r), 'max |ndcg diff|', max(r['abs_diff']['ndcg'] for r in ep if 'abs_diff' in r), 'last', ep[-1]['info'][:60])
"
done
