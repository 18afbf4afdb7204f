#!/bin/bash
# round 2, GPU run 15: lane groups per row / rows per wave
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02p
mkdir -p $OUT
cd $ROOT
run_variant() {  # name, env...
  local name=$1; shift
  for dt in fp32 bf16; do
    env "$@" timeout -k 10 300 python bench.py --act_dtype $dt --spmm_only 2>> $OUT/var.err | grep '^{' | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_spmm.jsonl
  done
  env "$@" timeout -k 10 300 python bench.py --no_cpu_baseline 2>> $OUT/var.err | grep '^{"metric' | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_step.jsonl
  for wl in amazon-book-shaped; do
    env "$@" timeout -k 10 300 python bench.py --workload $wl --spmm_only 2>> $OUT/var.err | grep '^{' | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_spmm.jsonl
  done
  echo "variant $name done"
}
run_variant base A=1
for v in wpb1 wpb2; do run_variant $v LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; done
python - <<'PY'
import json, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02p")
for l in open(os.path.join(root, "var_spmm.jsonl")):
    j = json.loads(l); r = j['roofline']
    print("spmm", j['variant'], j['workload'], j['act_dtype'], 'us', round(r['avg_launch_us'], 2))
for l in open(os.path.join(root, "var_step.jsonl")):
    j = json.loads(l)
    print("step", j['variant'], 'steps/s', round(j['value'], 1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec', 0), 1))
PY
