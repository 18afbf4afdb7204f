#!/bin/bash
# round 4, run 15: final state -- the whole GPU suite, smoke, the driver's bench command and the default bench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04i
mkdir -p $OUT
cd $ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
python bench.py --steps 20 --warmup 5 > $OUT/bench_gowalla_20steps.txt 2> $OUT/bench_gowalla_20steps.err; echo "bench rc=$?"; tail -1 $OUT/bench_gowalla_20steps.txt | cut -c1-300
python bench.py > $OUT/bench_gowalla.json 2>/dev/null; tail -1 $OUT/bench_gowalla.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('gowalla 400 steps', j['value'], j['steady_state_steps_per_sec'], j['roofline']['frac'], j['roofline']['traffic'], j['end_to_end_epoch']['prefetch_on']['steps_per_sec'], j['end_to_end_epoch']['prefetch_off']['steps_per_sec'], j['eval_topk']['ms'], j['eval_topk']['fp32_equivalent_vs_fp32_mfma_peak'], j['quality']['fp32']['abs_diff_recall'], j['quality']['bf16']['abs_diff_recall'])"
