#!/bin/bash
# run 61: data parallel rows mode with the rank's own rows added in part 1 (k_scatter: the other ranks' blocks only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bo
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert|Mismatch|Max abs|Max rel" $OUT/pytest.log | head -30; exit 1; }
timeout -k 10 600 python3 tools/dp_emulate_time.py 2>> $OUT/err.log | tail -1 | tee $OUT/dp_emulate.json | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print({k: (round(v['us_per_step'],1), round(v['compute_only_weak_scaling_efficiency'],3)) for k,v in j.items()})"
for mode in "--dp_reduce rows" "--dp_reduce dense" "--dp_shard rows"; do
  timeout -k 10 300 python bench.py --force_dp $mode --no_cpu_baseline --steps 200 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('dp1', '$mode', round(j['value'],1), j['config']['last_loss'])"
done
timeout -k 10 300 python bench.py --no_cpu_baseline --steps 200 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('single', round(j['value'],1), j['config']['last_loss'])"
