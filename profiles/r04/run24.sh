#!/bin/bash
# round 4, run 24 -- closing check of the final tree: the whole -m gpu suite, smoke, the kernel trace of the DEFAULT bench command (bf16
# storage on gowalla since this round, fp32 region beside it) and the driver's 20-step command
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_gowalla/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $OUT/trace_gowalla.log 2>&1 || echo "trace failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT/trace_gowalla > $OUT/trace_gowalla_bf16_summary.txt 2>&1; head -18 $OUT/trace_gowalla_bf16_summary.txt | cut -c1-150
cd $ROOT
( time python bench.py --steps 20 --warmup 5 > $OUT/bench_gowalla_20steps.txt 2> $OUT/bench_gowalla_20steps.err ) 2>&1 | tail -3
tail -1 $OUT/bench_gowalla_20steps.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('20 steps', j['dtype'][:4], j['value'], 'fp32', j.get('value_fp32'), 'steady', j['steady_state_steps_per_sec'], j['roofline']['frac'], j['roofline']['avg_launch_us'])"
