#!/bin/bash
# round 3, GPU run 6: the fork's optional branches inside the fused step (gate / item-item / both) vs the reference's fixtures
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03f
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "optional_branches" > $OUT/pytest_variants.log 2>&1; echo "variants rc=$?" | tee -a $OUT/status.log
tail -40 $OUT/pytest_variants.log | cut -c1-400
