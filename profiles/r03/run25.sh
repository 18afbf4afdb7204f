#!/bin/bash
# round 3, GPU run 25: device sampler in segments on the whole GPU (pairs / events / emit) vs the one-workgroup kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03z
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "sampler" > $OUT/pytest_sampler.log 2>&1; echo "pytest rc=$?" | tee $OUT/status.log
tail -3 $OUT/pytest_sampler.log
for v in default; do
  if [ $v = default ]; then unset LGCN_LIB_PATH; else export LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; fi
  echo $v $(timeout -k 10 300 python tools/sampler_time.py 2>/dev/null | tail -1) | tee -a $OUT/sampler_ab.txt
done
