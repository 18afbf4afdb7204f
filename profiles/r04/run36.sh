#!/bin/bash
# round 4, run 36 -- SQ counters of the dense layer on the last kernels (fp32 / bf16 tables), the same two passes as profiles/r03/run1.sh:
# instruction counts per class (VALU, VMEM, SMEM, LDS, SALU), active / wait cycles -- the counter side of run 29-31's "bound by the vector-memory
# instructions a CU retires", and what the scalar row descriptors of run 35 moved from VMEM to SMEM
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04n
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for dt in fp32 bf16; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1_$dt -- python3 $ROOT/bench.py --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/sq1_$dt.log 2>&1 || echo "sq1 $dt failed" | tee -a $OUT/status.log
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_sq2_$dt -- python3 $ROOT/bench.py --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/sq2_$dt.log 2>&1 || echo "sq2 $dt failed" | tee -a $OUT/status.log
done
python3 $ROOT/profiles/pmc_any.py $OUT "k_spmm" 2>&1 | tee $OUT/sq_summary.txt
