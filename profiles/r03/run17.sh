#!/bin/bash
# round 3, GPU run 17: optional branches, fused vs autograd, at K = 1 / 2 / 4 and with bf16 activation storage
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03r
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "optional_branches" > $OUT/pytest_variants.log 2>&1; echo "variants rc=$?" | tee -a $OUT/status.log
grep -E "passed|failed|^E  |Error" $OUT/pytest_variants.log | head -30 | cut -c1-300
