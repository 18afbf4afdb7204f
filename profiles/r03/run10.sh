#!/bin/bash
# round 3, GPU run 10: C5 shape -- streaming cache policy for the item rows' gathers (user table 10 GB, never re-used before eviction)
# and the XCD cut weighted towards the item rows; dense layer time + one PMC traffic pass of the best
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03k
mkdir -p $OUT
cd $ROOT
for cfg in "0 1.0" "auto 1.0" "auto 1.5" "auto 2.0" "auto 3.0" "0 2.0"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --workload synthetic-10m --spmm_only --stream_items $1 --xcd_item_weight $2 2>/dev/null | tail -1 | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); r=o['roofline']; print('c5 fp32 stream=$1 item_weight=$2 dense layer', round(r['avg_launch_us']/1000,2), 'ms frac', round(r['frac'],4))" | tee -a $OUT/ab.txt
done
for cfg in "0 1.0" "auto 2.0"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --workload synthetic-10m --spmm_only --act_dtype bf16 --stream_items $1 --xcd_item_weight $2 2>/dev/null | tail -1 | python3 -c "import sys,json; o=json.loads(sys.stdin.read()); r=o['roofline']; print('c5 bf16 stream=$1 item_weight=$2 dense layer', round(r['avg_launch_us']/1000,2), 'ms frac', round(r['frac'],4))" | tee -a $OUT/ab.txt
done
