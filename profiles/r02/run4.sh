#!/bin/bash
# round 2, GPU run 4: parity (fused eval, GPU sampler, row-sharded emulation, fused triplet loss), tuning
# variants, PMC diagnostics for the bf16 table, step trace, C5 line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02d
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -12 $OUT/pytest.log
run_variant() {  # name, env...
  local name=$1; shift
  for dt in fp32 bf16; do
    env "$@" timeout -k 10 300 python bench.py --act_dtype $dt --spmm_only 2>> $OUT/var.err | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_spmm.jsonl
  done
  env "$@" timeout -k 10 300 python bench.py --no_cpu_baseline 2>> $OUT/var.err | sed "s/^{/{\"variant\": \"$name\", /" >> $OUT/var_step.jsonl
  echo "variant $name done"
}
run_variant base A=1
for v in u4 w8 ch256 ch1024 nowin; do run_variant $v LGCN_LIB_PATH=$ROOT/build/variants/lib_$v.so; done
python - <<'PY'
import json, os
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02d")
for l in open(os.path.join(root, "var_spmm.jsonl")):
    j = json.loads(l); r = j['roofline']
    print("spmm", j['variant'], j['act_dtype'], 'us', round(r['avg_launch_us'], 2))
for l in open(os.path.join(root, "var_step.jsonl")):
    j = json.loads(l)
    print("step", j['variant'], 'steps/s', round(j['value'], 1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec', 0), 1))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $OUT/bench_trace.log 2>&1 || echo "trace failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT 2>&1 | head -22 | cut -c1-150
pmc() {  # tag dtype counters...
  local tag=$1 dt=$2; shift 2
  rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_${tag}_$dt -- python3 $ROOT/bench.py --spmm_only --spmm_reps 20 --act_dtype $dt > $OUT/pmc_${tag}_$dt.log 2>&1 || echo "pmc $tag $dt failed" | tee -a $OUT/status.log
  echo "pmc $tag $dt done"
}
pmc sq1 bf16 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS
pmc sq2 bf16 SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pmc tcp bf16 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
for dt in fp32 bf16; do
  pmc l2 $dt TCC_HIT_sum TCC_MISS_sum
  pmc fetch $dt FETCH_SIZE
  pmc write $dt WRITE_SIZE
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02d")
for d in sorted(glob.glob(os.path.join(root, "pmc_*_*"))):
    if not os.path.isdir(d): continue
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_spmm" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: round(sum(v[len(v)//5:]) / max(1, len(v[len(v)//5:])), 1) for k, v in agg.items()})
PY
cd $ROOT
timeout -k 10 600 python bench.py --workload synthetic-10m --no_cpu_baseline > $OUT/bench_c5.json 2> $OUT/bench_c5.err; echo "c5 rc=$?" | tee -a $OUT/status.log
cat $OUT/bench_c5.json; tail -12 $OUT/bench_c5.err
