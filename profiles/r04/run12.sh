#!/bin/bash
# round 4, run 12 -- PMC traffic passes of the dense layer on the 10M x 1M graph (fp32 / bf16 / fp8), merged into the round's hbm_traffic.json
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04f
mkdir -p $OUT/pmc_synthetic-10m
cd /tmp && export TMPDIR=/tmp
wl=synthetic-10m
for dt in fp32 bf16 fp8; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_$wl/pmc_fetch_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 5 --act_dtype $dt > $OUT/pmc_$wl/f_$dt.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_$wl/pmc_write_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 5 --act_dtype $dt > $OUT/pmc_$wl/w_$dt.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_$wl/pmc_l2_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 5 --act_dtype $dt > $OUT/pmc_$wl/l_$dt.log 2>&1
  echo "c5 $dt pmc done"
done
cp $ROOT/profiles/hbm_traffic.json $OUT/hbm_traffic_c5.json
python3 $ROOT/profiles/pmc_traffic.py $OUT/pmc_$wl --write $wl --out $OUT/hbm_traffic_c5.json | tee $OUT/pmc_${wl}_spmm.txt
