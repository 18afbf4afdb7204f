#!/bin/bash
# run 38: compute of one rank's data-parallel step at world 1..8 (collective replaced by a copy), kernel trace at world 8
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02ar
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python3 tools/dp_emulate_time.py 2> $OUT/err.log | tail -1 | tee $OUT/dp_emulate.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr/trace -- python3 $ROOT/tools/dp_emulate_time.py > $OUT/trace.log 2>&1
python3 $ROOT/profiles/summarize.py $OUT/tr 2>&1 | head -26 | cut -c1-140
