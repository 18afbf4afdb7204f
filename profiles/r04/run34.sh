#!/bin/bash
# round 4, run 34 -- Gowalla to the 1000-epoch horizon on the LAST kernels (bf16: shared Adam epilogue + three-wave k_triplet changed its rounding), upstream's loss
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r04m
cd $ROOT
for a in bf16 fp32; do
  python tools/gowalla_trajectory.py --epochs 1000 --act_dtype $a --prefetch_epoch 1 --reg_rows ego --quiet 1 --out gpurun_out/r04m/gowalla_1000ep_ego_$a.json 2>/dev/null | tail -1 | cut -c1-900
done
