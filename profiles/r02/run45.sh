#!/bin/bash
# run 45: PMC traffic of the dense layer at the C5 shape (10M x 1M, 200M edges, d = 256), fp32
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02ay
mkdir -p $OUT/pmc_c5
cp $ROOT/profiles/hbm_traffic.json $OUT/hbm_traffic.json
cd /tmp && export TMPDIR=/tmp
wl=synthetic-10m
for dt in fp32; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_c5/pmc_fetch_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 5 --act_dtype $dt > $OUT/pmc_c5/f_$dt.log 2>&1; echo "fetch rc=$?"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_c5/pmc_write_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 5 --act_dtype $dt > $OUT/pmc_c5/w_$dt.log 2>&1; echo "write rc=$?"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_c5/pmc_l2_$dt -- python3 $ROOT/bench.py --workload $wl --spmm_only --spmm_reps 5 --act_dtype $dt > $OUT/pmc_c5/l_$dt.log 2>&1; echo "l2 rc=$?"
done
python3 $ROOT/profiles/pmc_traffic.py $OUT/pmc_c5 --write $wl --out $OUT/hbm_traffic.json | tee $OUT/pmc_${wl}_summary.txt
