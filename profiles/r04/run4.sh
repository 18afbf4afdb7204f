#!/bin/bash
# round 4, run 4: the upstream-loss tests verbosely, then the rest of the GPU suite
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "upstream_loss" > gpurun_out/r04/pytest_run4a.txt 2>&1; echo "rc=$?"; grep -n "^E " gpurun_out/r04/pytest_run4a.txt | head -30; tail -8 gpurun_out/r04/pytest_run4a.txt
LGCN_SKIP_LARGE=1 python -m pytest tests -m gpu -q --deselect tests/test_gpu_parity.py::test_upstream_loss_fused_step_vs_oracle > gpurun_out/r04/pytest_run4b.txt 2>&1; echo "suite rc=$?"; grep -n "^E \|^FAILED" gpurun_out/r04/pytest_run4b.txt | head -30; tail -4 gpurun_out/r04/pytest_run4b.txt
