#!/bin/bash
# run 49: what k_triplet's 20 us are made of: skeleton (ids + indptr), no gathers, no loss/atomics (experiment builds, not kept)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bb
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in skel nogather noloss; do
  export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v/trace -o runc -- python3 $ROOT/bench.py --no_cpu_baseline --steps 100 --warmup 10 --spmm_reps 50 > $OUT/trace_$v.log 2>&1
  echo "== $v"; python3 $ROOT/profiles/summarize.py $OUT/$v 2>&1 | grep -E "k_triplet|k_g32" | cut -c1-140
done
