#!/bin/bash
# run 63: kernel trace at the C5 shape (10M x 1M, 200M edges, d = 256)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5/trace -- python3 $ROOT/bench.py --workload synthetic-10m --steps 6 --warmup 2 --no_cpu_baseline > $OUT/trace_c5.log 2>&1 || echo "trace failed"
python3 $ROOT/profiles/summarize.py $OUT/trace_c5 > $OUT/trace_synthetic-10m_summary.txt 2>&1; head -14 $OUT/trace_synthetic-10m_summary.txt | cut -c1-140
