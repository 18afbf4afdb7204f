#!/bin/bash
# kernel trace of the bench for one library variant: bash profiles/trace_only.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/tr_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline "$@" > $OUT/bench.log 2>&1 || exit 1
python3 $ROOT/profiles/summarize.py $OUT | head -10 | tail -8 | cut -c1-140
