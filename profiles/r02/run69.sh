#!/bin/bash
# run 69: processing order on the structure-free shapes: spectral (xcd) vs natural (users then items: an XCD sees one side's columns only) vs cocluster
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for wl in yelp2018-shaped amazon-book-shaped; do
  for ro in xcd natural cocluster; do
    timeout -k 10 600 python3 bench.py --workload $wl --row_order $ro --no_cpu_baseline --no_secondary --spmm_reps 300 2>/dev/null | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$wl $ro', round(j['value'],1), 'steps/s; dense layer us', round(j['roofline']['avg_launch_us'],1))"
  done
done
