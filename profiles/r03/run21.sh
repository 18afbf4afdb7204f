#!/bin/bash
# round 3, GPU run 21: SQ counters of k_eval_topk (what is a tile's 9 300 cycles per wave made of?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03ae
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_e1 -- python3 $ROOT/tools/eval_time.py > $OUT/e1.log 2>&1 || echo "e1 failed"
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_e2 -- python3 $ROOT/tools/eval_time.py > $OUT/e2.log 2>&1 || echo "e2 failed"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INSTS_VALU_MFMA_F32 --output-format csv -d $OUT/pmc_e3 -- python3 $ROOT/tools/eval_time.py > $OUT/e3.log 2>&1 || echo "e3 failed"
python3 $ROOT/profiles/pmc_any.py $OUT "k_eval_topk" 2>&1 | tee $OUT/sq_summary.txt
tail -2 $OUT/e2.log $OUT/e3.log | cut -c1-300
