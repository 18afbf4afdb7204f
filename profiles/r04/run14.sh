#!/bin/bash
# round 4, run 14: kernel trace of the device shuffle
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_shuffle/trace -- python3 $ROOT/tools/shuffle_time.py 806166 8000000 > $OUT/shuffle.log 2>&1
cat $OUT/shuffle.log | tail -3
python3 $ROOT/profiles/summarize.py $OUT/trace_shuffle > $OUT/trace_shuffle_summary.txt 2>&1; head -16 $OUT/trace_shuffle_summary.txt | cut -c1-150
