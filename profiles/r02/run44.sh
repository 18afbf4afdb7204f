#!/bin/bash
# run 44: dense k_spmm under a 64-VGPR cap (8 waves per SIMD, 12-20 B of scratch)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02ax
mkdir -p $OUT
cd $ROOT
run_variant() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no_cpu_baseline 2>> $OUT/var.err | grep '^{"metric' | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_step.jsonl
  echo "variant $name done"
}
for rep in 1 2; do
  run_variant base A=1
  for v in sw8; do run_variant $v LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; done
done
python - <<'PY'
import json, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02ax")
for l in open(os.path.join(root, "var_step.jsonl")):
    j = json.loads(l)
    print("step", j['variant'], 'steps/s', round(j['value'], 1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec', 0), 1), 'spmm', round(j['roofline']['avg_launch_us'],2))
PY
