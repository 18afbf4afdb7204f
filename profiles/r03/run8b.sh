#!/bin/bash
# round 3, GPU run 8b: bench lines of every workload (uses the PMC record of run 8a), C5 trace + PMC, tools
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03i
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python bench.py > $OUT/bench_gowalla.json 2> $OUT/bench_gowalla.err; echo "bench rc=$?" | tee -a $OUT/status.log
grep '^{"metric"' $OUT/bench_gowalla.json | cut -c1-600
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('driver-style 20 steps:', o['value'])" | tee $OUT/bench_gowalla_20steps.txt
for wl in yelp2018-shaped amazon-book-shaped; do
  timeout -k 10 600 python bench.py --workload $wl --no_cpu_baseline > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err; echo "$wl rc=$?" | tee -a $OUT/status.log
  grep '^{"metric"' $OUT/bench_$wl.json | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('$wl', o['value'], o['config'].get('bf16_activation_storage_steps_per_sec'), o['roofline']['avg_launch_us'], o['roofline']['frac'])"
done
timeout -k 10 600 python bench.py --workload synthetic-10m --no_cpu_baseline > $OUT/bench_synthetic-10m.json 2> $OUT/bench_c5.err; echo "c5 rc=$?" | tee -a $OUT/status.log
grep '^{"metric"' $OUT/bench_synthetic-10m.json | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('c5 fp32', o['value'], o['roofline']['avg_launch_us'], o['roofline']['frac'])"
timeout -k 10 600 python bench.py --workload synthetic-10m --no_cpu_baseline --act_dtype bf16 > $OUT/bench_synthetic-10m_bf16.json 2> $OUT/bench_c5b.err; echo "c5 bf16 rc=$?" | tee -a $OUT/status.log
grep '^{"metric"' $OUT/bench_synthetic-10m_bf16.json | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); print('c5 bf16', o['value'], o['roofline']['avg_launch_us'], o['roofline']['frac'])"
timeout -k 10 300 python tools/eval_time.py 2>> $OUT/eval.err | tail -1 | tee $OUT/eval_time.json | cut -c1-400
timeout -k 10 300 python tools/dp_emulate_time.py 2>> $OUT/dp.err | tail -1 | tee $OUT/dp_emulate.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5/trace -- python3 $ROOT/bench.py --workload synthetic-10m --steps 6 --warmup 2 --no_cpu_baseline > $OUT/trace_c5.log 2>&1 || echo "trace c5 failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT/trace_c5 > $OUT/trace_synthetic-10m_fp32_summary.txt 2>&1; head -14 $OUT/trace_synthetic-10m_fp32_summary.txt | cut -c1-140
cd $ROOT
timeout -k 10 600 python tools/variants_time.py 2>> $OUT/variants.err | tail -1 | tee $OUT/variants_time.json | cut -c1-1200
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_variants/trace -- python3 $ROOT/tools/variants_time.py > $OUT/trace_variants.log 2>&1 || echo "trace variants failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT/trace_variants > $OUT/trace_variants_summary.txt 2>&1; head -24 $OUT/trace_variants_summary.txt | cut -c1-140
cd $ROOT
timeout -k 10 300 python tools/sampler_time.py 2>> $OUT/sampler.err | tail -1 | tee $OUT/sampler_time.json | cut -c1-400
timeout -k 10 600 python tools/gowalla_trajectory.py 2>> $OUT/traj.err | tail -3 | tee $OUT/gowalla_trajectory.txt | cut -c1-600
