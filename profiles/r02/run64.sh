#!/bin/bash
# run 64: the C5 shape: item degree statistics, and the step with the last forward layer on the batch rows vs densely
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02br
mkdir -p $OUT
cd $ROOT
python3 - <<'PY' 2>/dev/null | tail -2
import importlib, numpy as np, torch
pkg = importlib.import_module('graph-and-sequential-recommendation-systems_amd')
ip, ix = pkg.synthetic.power_law_bipartite(10_000_000, 1_000_000, 200_000_000, seed=2020, device='cuda')
di = np.bincount(ix, minlength=1_000_000).astype(np.float64); du = np.diff(ip).astype(np.float64)
per = du.mean() + (di * di).sum() / di.sum() + di.mean()
print("max item degree", int(di.max()), "items > 32768 nnz:", int((di > 32768).sum()), "their share of the edges", round(float(di[di > 32768].sum() / di.sum()), 3),
      "sum d^2 / sum d", round(float((di * di).sum() / di.sum())), "B * per_triplet / nnz(A_hat)", round(2048 * per / (2 * len(ix)), 3))
PY
for dl in 0 1; do
  timeout -k 10 900 python3 bench.py --workload synthetic-10m --dense_last $dl --no_cpu_baseline 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('synthetic-10m dense_last=$dl', round(j['value'],3), 'steps/s', round(j['ms_per_step'],1), 'ms')"
done
