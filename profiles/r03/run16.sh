#!/bin/bash
# round 3, GPU run 16: the hub plan (last-layer rows of long rows computed once per step by k_spmm) at LOW thresholds on Gowalla:
# does taking the 500-1400-entry item rows out of k_triplet shorten its tail?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03q
mkdir -p $OUT
cd $ROOT
for cfg in "0 0" "1024 256" "512 256" "512 512" "256 256" "128 128"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --no_cpu_baseline --no_secondary --hub_nnz $1 --hub_chunk $2 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('gowalla hub_nnz=$1 chunk=$2 fp32', round(o['value']))" | tee -a $OUT/ab.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline --no_secondary --hub_nnz 512 --hub_chunk 256 > $OUT/trace.log 2>&1
python3 $ROOT/profiles/summarize.py $OUT/trace 2>&1 | head -12 | cut -c1-140
