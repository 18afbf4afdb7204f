#!/bin/bash
# round 4, run 8: the whole GPU suite again (test helper fixed)
mkdir -p gpurun_out/r04
LGCN_SKIP_LARGE=1 LGCN_SKIP_LONG=1 python -m pytest tests -m gpu -q > gpurun_out/r04/pytest_run8.txt 2>&1; echo "suite rc=$?"; grep -n "^E \|^FAILED" gpurun_out/r04/pytest_run8.txt | cut -c1-300 | head -30; tail -3 gpurun_out/r04/pytest_run8.txt
