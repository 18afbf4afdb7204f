#!/bin/bash
# round 3, GPU run 3: lgcn_train_epoch_dp at world 2-4 through the in-process loopback communicator (all three modes), full suite
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03c
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "loopback" > $OUT/pytest_loopback.log 2>&1; echo "loopback rc=$?" | tee -a $OUT/status.log
tail -15 $OUT/pytest_loopback.log
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/pytest_all.log 2>&1; echo "all rc=$?" | tee -a $OUT/status.log
tail -5 $OUT/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
