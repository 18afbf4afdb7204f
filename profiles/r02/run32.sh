#!/bin/bash
# run 32: where a dense launch spends its time (row classes alone)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02ak
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python tools/spmm_split.py 2> $OUT/split.err | tee $OUT/spmm_split.jsonl
