#!/bin/bash
# run 28: optional branches (popularity gate, item-item smoothing) vs fixtures captured from the reference; full gpu suite
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02af
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert|Mismatch|Max abs" $OUT/pytest.log | head -40; }
exit 0
