#!/bin/bash
# round 3, GPU run 1: parity suite with the new C5 full-shape test (ABI 8), then the VALU / VMEM / wait counters of the
# dense SpMM layer for fp32 and bf16 tables
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03a
mkdir -p $OUT
cd $ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --durations=12 > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -25 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for dt in fp32 bf16; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1_$dt -- python3 $ROOT/bench.py --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/sq1_$dt.log 2>&1 || echo "sq1 $dt failed" | tee -a $OUT/status.log
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_sq2_$dt -- python3 $ROOT/bench.py --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/sq2_$dt.log 2>&1 || echo "sq2 $dt failed" | tee -a $OUT/status.log
done
python3 $ROOT/profiles/pmc_any.py $OUT "k_spmm<64" 2>&1 | tee $OUT/sq_summary.txt
