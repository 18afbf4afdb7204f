#!/bin/bash
# run 52: the C5 shape with bf16 activation storage (headline stays fp32; this is the reported second mode at that shape)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02be
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python3 bench.py --workload synthetic-10m --act_dtype bf16 --no_cpu_baseline > $OUT/bench_c5_bf16.json 2> $OUT/err.log; echo "rc=$?"
grep '^{"metric"' $OUT/bench_c5_bf16.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('c5 bf16', round(j['value'],2), 'steps/s', round(j['ms_per_step'],1), 'ms; dense layer', round(r['avg_launch_us']/1e3,2), 'ms frac', round(r['frac'],3), j['dtype'])"
tail -3 $OUT/err.log
