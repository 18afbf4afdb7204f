#!/bin/bash
# run 66: threshold (and, in an experiment build, chunk length) of the hub plan on the C5 shape
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bu
mkdir -p $OUT
cd $ROOT
for thr in 8192 32768 65536 131072 262144 524288; do
  LGCN_TRIPLET_HUB_NNZ=$thr timeout -k 10 900 python3 bench.py --workload synthetic-10m --no_cpu_baseline --steps 10 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('hub > $thr nnz:', round(j['value'],3), 'steps/s', round(j['ms_per_step'],1), 'ms')"
done
