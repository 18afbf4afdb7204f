#!/bin/bash
# round 3, GPU run 27: device sampler on a 200 M-triplet epoch (synthetic-10m), segments vs the one-workgroup kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03z
mkdir -p $OUT
cd $ROOT
for v in default onewg; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  echo $v $(timeout -k 10 500 python tools/sampler_time.py --c5 2>$OUT/c5_$v.err | tail -1) | tee -a $OUT/sampler_c5.txt
done
