#!/bin/bash
# L2 hit/miss + fetch counters of the bench kernels for one configuration: bash profiles/pmc_l2.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no_cpu_baseline "$@" > $OUT/bench.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no_cpu_baseline "$@" > $OUT/bench2.log 2>&1 || exit 1
python3 $ROOT/profiles/summarize.py $OUT | grep -E "k_spmm<64, (float, float|__bf16, __bf16), 0>|k_bpr" 
