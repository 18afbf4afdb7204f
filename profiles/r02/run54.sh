#!/bin/bash
# run 54 (experiment, reverted): sparse-input layer with 2 / 4 rows per lane group out of one staged pack (shared plan / stream / bitmap round trips)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bg
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert|Mismatch|Max abs|Max rel" $OUT/pytest.log | head -30; exit 1; }
run_variant() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no_cpu_baseline 2>> $OUT/var.err | grep '^{"metric' | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_step.jsonl
  echo "variant $name done"
}
for rep in 1 2 3; do
  run_variant base A=1
  for v in rpg1 rpg4; do run_variant $v LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; done
done
python - <<'PY'
import json, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02bg")
for l in open(os.path.join(root, "var_step.jsonl")):
    j = json.loads(l)
    print("step", j['variant'], 'steps/s', round(j['value'], 1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec', 0), 1), 'spmm', round(j['roofline']['avg_launch_us'],2))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tr/trace -o runc -- python3 $ROOT/bench.py --no_cpu_baseline --steps 100 --warmup 10 --spmm_reps 50 > $OUT/trace.log 2>&1
python3 $ROOT/profiles/summarize.py $OUT/tr 2>&1 | grep -E "Li3E|, 3>" | cut -c1-140
