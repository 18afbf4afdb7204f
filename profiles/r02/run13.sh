#!/bin/bash
# round 2, GPU run 13: Adam operands prefetched before the gathers
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02m
mkdir -p $OUT
cd $ROOT
run_variant() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no_cpu_baseline 2>> $OUT/var.err | grep '^{"metric' | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_step.jsonl
  echo "variant $name done"
}
run_variant pref A=1
run_variant nopref LGCN_LIB_PATH=$ROOT/build/variants/lib_nopref.so
run_variant pref2 A=1
run_variant nopref2 LGCN_LIB_PATH=$ROOT/build/variants/lib_nopref.so
python - <<'PY'
import json, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02m")
for l in open(os.path.join(root, "var_step.jsonl")):
    j = json.loads(l)
    print("step", j['variant'], 'steps/s', round(j['value'], 1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec', 0), 1))
PY
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $OUT/trace.log 2>&1
python3 $ROOT/profiles/summarize.py $OUT/trace 2>&1 | head -16 | cut -c1-140
