#!/bin/bash
# run 65: hub plan for k_triplet (rows > 32768 non-zeros computed by k_spmm once per step): tests, C5 and Gowalla steps, C5 trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02bs
mkdir -p $OUT
cd $ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $OUT/pytest.log
[ $rc -ne 0 ] && { grep -E "^(FAILED|ERROR)|Error|assert|Mismatch|Max abs|Max rel" $OUT/pytest.log | head -30; exit 1; }
timeout -k 10 900 python3 bench.py --workload synthetic-10m --no_cpu_baseline 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('synthetic-10m', round(j['value'],3), 'steps/s', round(j['ms_per_step'],1), 'ms', 'loss', j['config']['last_loss'])"
LGCN_TRIPLET_HUB_NNZ=0 timeout -k 10 900 python3 bench.py --workload synthetic-10m --no_cpu_baseline 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('synthetic-10m no hub plan', round(j['value'],3), 'steps/s', round(j['ms_per_step'],1), 'ms', 'loss', j['config']['last_loss'])"
timeout -k 10 300 python3 bench.py --no_cpu_baseline 2>> $OUT/err.log | grep '^{"metric"' | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('gowalla', round(j['value'],1), 'bf16', round(j['config'].get('bf16_activation_storage_steps_per_sec',0),1))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c5/trace -- python3 $ROOT/bench.py --workload synthetic-10m --steps 6 --warmup 2 --no_cpu_baseline > $OUT/trace_c5.log 2>&1 || echo "trace failed"
python3 $ROOT/profiles/summarize.py $OUT/trace_c5 > $OUT/trace_synthetic-10m_summary.txt 2>&1; head -10 $OUT/trace_synthetic-10m_summary.txt | cut -c1-140
