#!/bin/bash
# round 2, GPU run 5: packed plan-ordered index stream (fewer vector-memory instructions)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02e
mkdir -p $OUT
cd $ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/status.log
tail -12 $OUT/pytest.log
for wl in gowalla yelp2018-shaped amazon-book-shaped; do
  for dt in fp32 bf16; do
    timeout -k 10 300 python bench.py --workload $wl --act_dtype $dt --spmm_only 2>> $OUT/spmm_sweep.err >> $OUT/spmm_sweep.jsonl || echo "sweep $wl $dt failed" | tee -a $OUT/status.log
  done
done
python - <<'PY'
import json, os
for l in open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02e/spmm_sweep.jsonl")):
    j = json.loads(l); r = j['roofline']
    print(j['workload'], j['act_dtype'], 'us', round(r['avg_launch_us'],2), 'frac', round(r['frac'],3))
PY
timeout -k 10 600 python bench.py --no_cpu_baseline > $OUT/bench_gowalla.json 2> $OUT/bench_gowalla.err; echo "bench rc=$?" | tee -a $OUT/status.log
cat $OUT/bench_gowalla.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 100 --warmup 10 --no_cpu_baseline > $OUT/bench_trace.log 2>&1 || echo "trace failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT 2>&1 | head -20 | cut -c1-150
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1_fp32 -- python3 $ROOT/bench.py --spmm_only --spmm_reps 20 > $OUT/pmc_sq1_fp32.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq1_bf16 -- python3 $ROOT/bench.py --spmm_only --spmm_reps 20 --act_dtype bf16 > $OUT/pmc_sq1_bf16.log 2>&1
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r02e")
for d in sorted(glob.glob(os.path.join(root, "pmc_*_*"))):
    if not os.path.isdir(d): continue
    agg = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_spmm" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: round(sum(v[len(v)//5:]) / max(1, len(v[len(v)//5:])), 1) for k, v in agg.items()})
PY
