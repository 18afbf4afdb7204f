#!/usr/bin/env python3
"""What a K-step timed region pays besides K steps: model.fused_epoch on Gowalla for K = 1, 20, 400 -- wall time of the call + wait
(synchronised before and after), the host's own time until the call returns (Python + K x 7 launches enqueued), and the GPU-side
span between an event recorded right before the call and one right after it."""
import importlib, io, contextlib, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import materialize_gowalla, GOWALLA_NPZ
act = sys.argv[1] if len(sys.argv) > 1 else "bf16"
sys.argv = [sys.argv[0]]
import torch
pkg = importlib.import_module("graph-and-sequential-recommendation-systems_amd")
w = pkg.world
w.configure(["--dataset", "gowalla", "--tensorboard", "0", "--act_dtype", act])
d = materialize_gowalla(GOWALLA_NPZ, "/tmp/lgcn_ov_gowalla")
with contextlib.redirect_stdout(io.StringIO()):
    ds = pkg.dataloader.Loader(w.config, path=d)
    pkg.sampling.seed(2020); pkg.utils.set_seed(2020)
    model = pkg.model.LightGCN(w.config, ds).to(w.device)
B = 2048
u, p, n = pkg.Procedure.sample_epoch_to_device(ds, w.device)
model.fused_epoch(u[:400 * B], p[:400 * B], n[:400 * B], B); torch.cuda.synchronize()
out = {"act_dtype": act}
for K in (1, 20, 400):
    reps = 40 if K < 400 else 5
    wall, host, gpu = [], [], []
    for r in range(reps):
        us, ps, ns = u[:K * B], p[:K * B], n[:K * B]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e0.record()
        model.fused_epoch(us, ps, ns, B)
        t1 = time.perf_counter()
        e1.record()
        while not e1.query():
            pass
        t2 = time.perf_counter()
        wall.append(t2 - t0); host.append(t1 - t0); gpu.append(e0.elapsed_time(e1) * 1e-3)
    med = lambda x: sorted(x)[len(x) // 2]
    out[f"K={K}"] = {"wall_us": 1e6 * med(wall), "host_return_us": 1e6 * med(host), "gpu_span_us": 1e6 * med(gpu)}
per = out["K=400"]["wall_us"] / 400
out["us_per_step_at_400"] = per
out["fixed_us_in_a_20_step_region"] = out["K=20"]["wall_us"] - 20 * per
print(json.dumps(out))
