#!/bin/bash
# round 2, GPU run 7: per-row cost in the XCD slice balance
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02g
mkdir -p $OUT
cd $ROOT
for c in 0 8 16 32 64; do
  for dt in fp32 bf16; do
    LGCN_ROW_COST=$c timeout -k 10 300 python bench.py --act_dtype $dt --spmm_only 2>> $OUT/var.err | sed "s/^{/{\"row_cost\": $c, /" >> $OUT/var_spmm.jsonl
  done
  LGCN_ROW_COST=$c timeout -k 10 300 python bench.py --no_cpu_baseline 2>> $OUT/var.err | sed "s/^{/{\"row_cost\": $c, /" >> $OUT/var_step.jsonl
  echo "cost $c done"
done
python - <<'PY'
import json, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02g")
for l in open(os.path.join(root, "var_spmm.jsonl")):
    j = json.loads(l); r = j['roofline']
    print("spmm cost", j['row_cost'], j['act_dtype'], 'us', round(r['avg_launch_us'], 2))
for l in open(os.path.join(root, "var_step.jsonl")):
    j = json.loads(l)
    print("step cost", j['row_cost'], 'steps/s', round(j['value'], 1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec', 0), 1))
PY
for wl in yelp2018-shaped amazon-book-shaped; do
  for dt in fp32 bf16; do
    timeout -k 10 300 python bench.py --workload $wl --act_dtype $dt --spmm_only 2>> $OUT/spmm_sweep.err | tee -a $OUT/spmm_sweep.jsonl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['workload'], j['act_dtype'], round(j['roofline']['avg_launch_us'],2))"
  done
done
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -4 $OUT/pytest.log
