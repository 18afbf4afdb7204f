#!/bin/bash
# run 19: what bounds k_rows? skeleton (ids + indptr only), no-gather, one-wave variants under the kernel trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02w
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in skel nogather bw1; do
  export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v/trace -o runc -- python3 $ROOT/bench.py --no_cpu_baseline --steps 100 --warmup 10 > $OUT/trace_$v.log 2>&1
  echo "== $v"; python3 $ROOT/profiles/summarize.py $OUT/$v 2>&1 | grep -E "k_rows|k_bpr|k_g32" | cut -c1-140
done
