#!/bin/bash
# round 2, GPU run 6: software-pipelined multi-pack waves; variants; short/long split timing
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02f
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -6 $OUT/pytest.log
run_variant() {  # name, env...
  local name=$1; shift
  for dt in fp32 bf16; do
    env "$@" timeout -k 10 300 python bench.py --act_dtype $dt --spmm_only 2>> $OUT/var.err | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_spmm.jsonl
  done
  env "$@" timeout -k 10 300 python bench.py --no_cpu_baseline 2>> $OUT/var.err | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_step.jsonl
  echo "variant $name done"
}
run_variant base A=1
for v in rpw8 rpw32 w6 ch256 ch128; do run_variant $v LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; done
python - <<'PY'
import json, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02f")
for l in open(os.path.join(root, "var_spmm.jsonl")):
    j = json.loads(l); r = j['roofline']
    print("spmm", j['variant'], j['act_dtype'], 'us', round(r['avg_launch_us'], 2))
for l in open(os.path.join(root, "var_step.jsonl")):
    j = json.loads(l)
    print("step", j['variant'], 'steps/s', round(j['value'], 1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec', 0), 1))
PY
timeout -k 10 300 python tools/spmm_split.py 2>> $OUT/split.err | tee $OUT/split.jsonl
for wl in yelp2018-shaped amazon-book-shaped; do
  for dt in fp32 bf16; do
    timeout -k 10 300 python bench.py --workload $wl --act_dtype $dt --spmm_only 2>> $OUT/spmm_sweep.err | tee -a $OUT/spmm_sweep.jsonl | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['workload'], j['act_dtype'], round(j['roofline']['avg_launch_us'],2))"
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $OUT/bench_trace.log 2>&1 || echo "trace failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT 2>&1 | head -18 | cut -c1-150
