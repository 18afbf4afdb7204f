#!/bin/bash
# run 29: steps/s of the optional branches (autograd path) on Gowalla
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02ag
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python tools/variants_time.py 2> $OUT/variants.err | tail -1 | tee $OUT/variants_time.json
tail -3 $OUT/variants.err
