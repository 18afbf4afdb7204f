#!/bin/bash
# round 4, run 11 -- second half of the measurement pass: C5 traces (fp32, fp8), bench lines of every workload / storage type with the
# refreshed PMC record, the data-parallel paths through real RCCL at world size 1 (rows, dense, cols)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04g
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for dt in fp32 fp8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5_$dt/trace -- python3 $ROOT/bench.py --workload synthetic-10m --act_dtype $dt --steps 10 --warmup 2 --no_cpu_baseline > $OUT/trace_c5_$dt.log 2>&1 || echo "trace c5 $dt failed" | tee -a $OUT/status.log
  python3 $ROOT/profiles/summarize.py $OUT/trace_c5_$dt > $OUT/trace_synthetic-10m_${dt}_summary.txt 2>&1; head -12 $OUT/trace_synthetic-10m_${dt}_summary.txt | cut -c1-150
done
cd $ROOT
# (the default storage type on gowalla is bf16 -- BASELINE configs[1] names it; the fp32 run of the same region is in the same line, and run here on its own too)
python bench.py --steps 20 --warmup 5 > $OUT/bench_gowalla_20steps.txt 2> $OUT/bench_gowalla_20steps.err; tail -1 $OUT/bench_gowalla_20steps.txt | cut -c1-400
tail -3 $OUT/bench_gowalla_20steps.err
python bench.py --steps 20 --warmup 5 --act_dtype fp32 > $OUT/bench_gowalla_20steps_fp32.txt 2>/dev/null; tail -1 $OUT/bench_gowalla_20steps_fp32.txt | cut -c1-400
python bench.py > $OUT/bench_gowalla.json 2>/dev/null; tail -1 $OUT/bench_gowalla.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('gowalla 400 steps', j['dtype'][:4], j['value'], j['steady_state_steps_per_sec'], j.get('value_fp32'), j['roofline']['frac'], j['roofline']['traffic'])"
python bench.py --act_dtype fp32 > $OUT/bench_gowalla_fp32.json 2>/dev/null; tail -1 $OUT/bench_gowalla_fp32.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('gowalla 400 steps', j['dtype'][:4], j['value'], j['steady_state_steps_per_sec'], j.get('value_bf16'), j['roofline']['frac'], j['roofline']['traffic'])"
for spec in "yelp2018-shaped fp32" "yelp2018-shaped bf16" "amazon-book-shaped fp32" "amazon-book-shaped bf16" "amazon-book-shaped fp8" "synthetic-10m fp32" "synthetic-10m bf16" "synthetic-10m fp8"; do
  set -- $spec
  timeout -k 10 600 python bench.py --workload $1 --act_dtype $2 --no_cpu_baseline --no_secondary > $OUT/bench_$1_$2.json 2>/dev/null
  tail -1 $OUT/bench_$1_$2.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1 $2', round(j['value'],2), j.get('steady_state_steps_per_sec'), round(j['roofline']['avg_launch_us'],1), round(j['roofline']['frac'],4), j['roofline']['traffic'])" | tee -a $OUT/bench_lines.txt
done
for mode in "--dp_reduce rows" "--dp_reduce dense" "--dp_shard rows" "--dp_shard cols"; do
  HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python bench.py --force_dp $mode --act_dtype fp32 --no_cpu_baseline --no_secondary --no_epochs --no_eval 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('world-1 RCCL $mode', round(j['value'],1), j['rccl_ranks_observed'], j['rccl_ranks_source'], j['config']['last_loss'])" | tee -a $OUT/dp_world1_rccl.txt
done
