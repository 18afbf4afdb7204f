#!/bin/bash
# run 48 (46-47 were an instrumented variant, see experiments.txt): coordinate descent of the XCD slice cuts on the product kernel launch time
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02ba
mkdir -p $OUT
cd $ROOT
timeout -k 10 500 python3 tools/slice_times.py --descend $OUT/descend.json 2>$OUT/err.log | cut -c1-300
