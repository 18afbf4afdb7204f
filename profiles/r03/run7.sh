#!/bin/bash
# round 3, GPU run 7: steps/s of the optional branches inside the fused step vs the autograd path (Gowalla), kernel trace of the gate step
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03g
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python tools/variants_time.py 2> $OUT/variants.err | tail -1 | tee $OUT/variants_time.json | cut -c1-1500
