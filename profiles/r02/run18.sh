#!/bin/bash
# run 18: kernel trace of the default Gowalla bench after k_g32
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02v
mkdir -p $OUT/tr
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr/trace -o runc -- python3 $ROOT/bench.py --no_cpu_baseline --steps 100 --warmup 10 > $OUT/trace.log 2>&1
python3 $ROOT/profiles/summarize.py $OUT/tr 2>&1 | head -24 | cut -c1-140
