#!/bin/bash
# round 3, GPU run 13: gate kernels after unrolling (k_gate_adam 32 partials per round trip, LDS loops of k_triplet_gate): fixtures, rates, trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03n
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "optional_branches" > $OUT/pytest_variants.log 2>&1; echo "variants rc=$?" | tee -a $OUT/status.log
tail -3 $OUT/pytest_variants.log | cut -c1-300
timeout -k 10 600 python tools/variants_time.py 2>> $OUT/variants.err | tail -1 | tee $OUT/variants_time.json | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_variants/trace -- python3 $ROOT/tools/variants_time.py > $OUT/trace_variants.log 2>&1 || echo "trace variants failed" | tee -a $OUT/status.log
python3 $ROOT/profiles/summarize.py $OUT/trace_variants > $OUT/trace_variants_summary.txt 2>&1; head -14 $OUT/trace_variants_summary.txt | cut -c1-140
