#!/bin/bash
# round 3, GPU run 2: gather inner loop without per-gather clamp/select/64-bit address (zero-weight padding in LDS, 32-bit offsets):
# parity suite (C5 test with its diagnostics), Gowalla bench fp32 + bf16, SQ counters of the new dense layer
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03b
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $OUT/pytest_parity.log 2>&1; echo "parity rc=$?" | tee -a $OUT/status.log
tail -5 $OUT/pytest_parity.log
timeout -k 10 600 python -m pytest tests/test_gpu_large.py -m gpu -q -x -s > $OUT/pytest_large.log 2>&1; echo "large rc=$?" | tee -a $OUT/status.log
grep -E "^\[c5|passed|failed|Error" $OUT/pytest_large.log | tail -12
timeout -k 10 600 python bench.py --no_cpu_baseline > $OUT/bench_gowalla.json 2> $OUT/bench_gowalla.err; echo "bench rc=$?" | tee -a $OUT/status.log
grep '^{"metric"' $OUT/bench_gowalla.json | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['config'].get('bf16_activation_storage_steps_per_sec'), o['roofline']['avg_launch_us'])"
timeout -k 10 300 python bench.py --spmm_only --act_dtype bf16 | tail -1 | cut -c1-300
for wl in yelp2018-shaped amazon-book-shaped; do
  timeout -k 10 600 python bench.py --workload $wl --no_cpu_baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "$wl rc=$?" | tee -a $OUT/status.log
  grep '^{"metric"' $OUT/bench_$wl.json | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print(o['value'], o['config'].get('bf16_activation_storage_steps_per_sec'), o['roofline']['avg_launch_us'])"
done
cd /tmp && export TMPDIR=/tmp
for dt in fp32 bf16; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1_$dt -- python3 $ROOT/bench.py --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/sq1_$dt.log 2>&1 || echo "sq1 $dt failed" | tee -a $OUT/status.log
done
python3 $ROOT/profiles/pmc_any.py $OUT "k_spmm" 2>&1 | tee $OUT/sq_summary.txt
