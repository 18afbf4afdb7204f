#!/bin/bash
# round 4, run 19: fp8 tables with chunk-interleaved columns (coalesced fp32 side of the epilogues): parity, then C4 / C5 in fp8
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_fp8.py tests/test_gpu_parity.py -m gpu -q -k "fp8 or fused_step_vs_oracle or call_shapes" > gpurun_out/r04/pytest_run19.txt 2>&1; echo "rc=$?"; grep -n "^E \|^FAILED" gpurun_out/r04/pytest_run19.txt | cut -c1-300 | head -20; tail -2 gpurun_out/r04/pytest_run19.txt
for w in amazon-book-shaped; do
timeout -k 10 400 python bench.py --workload $w --act_dtype fp8 --no_cpu_baseline --no_secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$w fp8', j['value'], j.get('steady_state_steps_per_sec'), j['roofline']['avg_launch_us'])"
done
timeout -k 10 600 python bench.py --workload synthetic-10m --act_dtype fp8 --no_cpu_baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('c5 fp8', j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'])"
timeout -k 10 300 python bench.py --act_dtype fp8 --no_cpu_baseline --no_epochs --no_eval --no_secondary --no_steady 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('gowalla fp8', j['value'])"
