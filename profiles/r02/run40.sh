#!/bin/bash
# run 40: kernel trace of the emulated data-parallel compute at world 1..8
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02at
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr/trace -- python3 $ROOT/tools/dp_emulate_time.py > $OUT/trace.log 2>&1
python3 $ROOT/profiles/summarize.py $OUT/tr 2>&1 | head -12 | cut -c1-140
