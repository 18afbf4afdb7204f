#!/bin/bash
# round 2, GPU run 14: two lane groups per bf16 row (4 rows per wave) vs one (8 rows per wave)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02n
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -3 $OUT/pytest.log
run_variant() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --act_dtype bf16 --spmm_only 2>> $OUT/var.err | grep '^{' | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_spmm.jsonl
  env "$@" timeout -k 10 300 python bench.py --no_cpu_baseline 2>> $OUT/var.err | grep '^{"metric' | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_step.jsonl
  echo "variant $name done"
}
run_variant gpr2 A=1
run_variant gpr1 LGCN_LIB_PATH=$ROOT/build/variants/lib_gpr1.so
run_variant gpr2b A=1
run_variant gpr1b LGCN_LIB_PATH=$ROOT/build/variants/lib_gpr1.so
python - <<'PY'
import json, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02n")
for l in open(os.path.join(root, "var_spmm.jsonl")):
    j = json.loads(l); r = j['roofline']
    print("spmm", j['variant'], j['act_dtype'], 'us', round(r['avg_launch_us'], 2))
for l in open(os.path.join(root, "var_step.jsonl")):
    j = json.loads(l)
    print("step", j['variant'], 'steps/s', round(j['value'], 1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec', 0), 1))
PY
for wl in yelp2018-shaped amazon-book-shaped; do
  timeout -k 10 300 python bench.py --workload $wl --act_dtype bf16 --spmm_only 2>> $OUT/var.err | grep '^{' | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['workload'], j['act_dtype'], round(j['roofline']['avg_launch_us'],2))"
done
