#!/bin/bash
# round 2, GPU run 2: parity suite on SpMM v2 (row per lane group) + sweep
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02b
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -15 $OUT/pytest.log
for wl in gowalla yelp2018-shaped amazon-book-shaped; do
  for dt in fp32 bf16; do
    timeout -k 10 300 python bench.py --workload $wl --act_dtype $dt --spmm_only >> $OUT/spmm_sweep.jsonl 2>> $OUT/spmm_sweep.err || echo "sweep $wl $dt failed" | tee -a $OUT/status.log
  done
done
python - <<'PY'
import sys, json, os
for l in open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02b/spmm_sweep.jsonl")):
    j = json.loads(l); r = j['roofline']
    print(j['workload'], j['row_order'], j['act_dtype'], 'us', round(r['avg_launch_us'],2), 'frac', round(r['frac'],3))
PY
timeout -k 10 600 python bench.py --no_cpu_baseline > $OUT/bench_gowalla.json 2> $OUT/bench_gowalla.err; echo "bench rc=$?" | tee -a $OUT/status.log
cat $OUT/bench_gowalla.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $OUT/bench_trace.log 2>&1 || echo "trace failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT 2>&1 | head -24 | cut -c1-150
