#!/bin/bash
# Profiling recipe used for the committed summaries (run on the GPU box through gpurun):
#   bash profiles/run_profile.sh <tag> [extra bench args]
# 1. kernel trace + stats of the default bench command, 2-4. PMC passes (own runs).
set -o pipefail
TAG=${1:-r01}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline "$@" > $OUT/bench_trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no_cpu_baseline "$@" > $OUT/bench_pmc1.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no_cpu_baseline "$@" > $OUT/bench_pmc2.log 2>&1 || exit 1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no_cpu_baseline "$@" > $OUT/bench_pmc3.log 2>&1 || exit 1
python3 $ROOT/profiles/summarize.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
