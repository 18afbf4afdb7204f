#!/bin/bash
# round 2, GPU run 11: long-row chunks walked by a whole workgroup
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02k
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -4 $OUT/pytest.log
run_variant() {  # name, env...
  local name=$1; shift
  for dt in fp32 bf16; do
    env "$@" timeout -k 10 300 python bench.py --act_dtype $dt --spmm_only 2>> $OUT/var.err | grep '^{' | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_spmm.jsonl
  done
  env "$@" timeout -k 10 300 python bench.py --no_cpu_baseline 2>> $OUT/var.err | grep '^{"metric' | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_step.jsonl
  echo "variant $name done"
}
run_variant base A=1
for v in ch256 ch1024 ch2048; do run_variant $v LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; done
python - <<'PY'
import json, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02k")
for l in open(os.path.join(root, "var_spmm.jsonl")):
    j = json.loads(l); r = j['roofline']
    print("spmm", j['variant'], j['act_dtype'], 'us', round(r['avg_launch_us'], 2))
for l in open(os.path.join(root, "var_step.jsonl")):
    j = json.loads(l)
    print("step", j['variant'], 'steps/s', round(j['value'], 1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec', 0), 1))
PY
for wl in yelp2018-shaped amazon-book-shaped; do
  timeout -k 10 600 python bench.py --workload $wl --no_cpu_baseline 2> $OUT/bench_$wl.err | grep '^{"metric' > $OUT/bench_$wl.json
  python -c "import sys,json; j=json.loads(open('$OUT/bench_$wl.json').read()); print(j['config']['workload'][:40], round(j['value'],2), 'steps/s', 'bf16', j['config'].get('bf16_activation_storage_steps_per_sec'), 'spmm us', round(j['roofline']['avg_launch_us'],1), 'frac', round(j['roofline']['frac'],3))"
done
