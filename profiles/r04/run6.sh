#!/bin/bash
# round 4, run 6: fp8 tests again (fixed tolerances), then fp8 activation storage on the structure-free shapes C3 / C4 / C5
mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_fp8.py -m gpu -q > gpurun_out/r04/pytest_run6.txt 2>&1; echo "rc=$?"; grep -n "^E \|^FAILED" gpurun_out/r04/pytest_run6.txt | cut -c1-300 | head -20; tail -2 gpurun_out/r04/pytest_run6.txt
for w in yelp2018-shaped amazon-book-shaped; do
timeout -k 10 400 python bench.py --workload $w --act_dtype fp8 --no_cpu_baseline --no_secondary > gpurun_out/r04/bench_${w}_fp8.txt 2> gpurun_out/r04/bench_${w}_fp8.err; echo "$w rc=$?"; tail -1 gpurun_out/r04/bench_${w}_fp8.txt | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j.get('steady_state_steps_per_sec'), j['roofline']['avg_launch_us'], j['roofline']['frac'])"
done
timeout -k 10 600 python bench.py --workload synthetic-10m --act_dtype fp8 --no_cpu_baseline > gpurun_out/r04/bench_synthetic-10m_fp8.txt 2> gpurun_out/r04/bench_synthetic-10m_fp8.err; echo "c5 rc=$?"; tail -1 gpurun_out/r04/bench_synthetic-10m_fp8.txt | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'], j['roofline']['frac'])"; tail -3 gpurun_out/r04/bench_synthetic-10m_fp8.err
