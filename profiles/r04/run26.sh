#!/bin/bash
# round 4, run 26 -- end-to-end Gowalla epochs when the triplets come from the host (python-mode / cpp-mode host samplers: the paths with a
# per-epoch PCIe upload), prefetch on / off, fp32 / bf16, next to the device sampler; the 9.7 MB upload alone
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04k
cd $ROOT
timeout -k 10 800 python tools/host_sampler_epoch.py --out gpurun_out/r04k/host_sampler_epoch.json
