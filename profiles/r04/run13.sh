#!/bin/bash
# round 4, run 13: the device shuffle -- bit-exactness tests, then its time next to the host's, then the end-to-end epoch in strict order
mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "shuffle or epochs_tiny or procedure_epoch or gowalla_full" > gpurun_out/r04/pytest_run13.txt 2>&1; echo "rc=$?"; grep -n "^E \|^FAILED" gpurun_out/r04/pytest_run13.txt | cut -c1-300 | head; tail -2 gpurun_out/r04/pytest_run13.txt
timeout -k 10 300 python - <<'PY' 2>&1 | tail -8 | tee gpurun_out/r04/shuffle_time.txt
import importlib, time, torch, numpy as np
pkg = importlib.import_module("graph-and-sequential-recommendation-systems_amd")
U = pkg.utils
for n in (806166, 1237259, 8 * 1000 * 1000, 50 * 1000 * 1000):
    U.set_seed(1)
    t0 = time.perf_counter(); h = U.shuffle_indices(n); th = time.perf_counter() - t0
    U.set_seed(1)
    d = U.shuffle_indices_device(n, "cuda:0"); torch.cuda.synchronize()
    U.set_seed(1)
    t0 = time.perf_counter(); d = U.shuffle_indices_device(n, "cuda:0"); torch.cuda.synchronize(); td = time.perf_counter() - t0
    print(f"n {n}: host {th*1e3:.2f} ms, device {td*1e3:.2f} ms, equal {bool((d.cpu().numpy() == h).all())}")
PY
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_eval --no_secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps(j['end_to_end_epoch'])[:500]); print(json.dumps(j['quality']['fp32']))"
