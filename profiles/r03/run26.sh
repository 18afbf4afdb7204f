#!/bin/bash
# round 3, GPU run 26: kernel trace of the segmented device sampler
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03z
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sampler/trace -- python3 $ROOT/tools/sampler_time.py > $OUT/trace_sampler.log 2>&1 || echo "trace failed"
python3 $ROOT/profiles/summarize.py $OUT/trace_sampler > $OUT/trace_sampler_summary.txt 2>&1; head -14 $OUT/trace_sampler_summary.txt | cut -c1-150
