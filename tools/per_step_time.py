#!/usr/bin/env python3
"""BPRLoss.stageOne called per step (the reference's call shape, utils.py:53-64 / main.py:223-225) against the one-C-call epoch: Gowalla, 394 steps.
ids as the reference passes them (int64 device tensors) and as int32; with the per-step .item() of the reference and without (bpr.lazy)."""
import contextlib, importlib, io, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import materialize_gowalla, GOWALLA_NPZ
sys.argv = [sys.argv[0]]
import torch
pkg = importlib.import_module("graph-and-sequential-recommendation-systems_amd")
w = pkg.world
w.configure(["--dataset", "gowalla", "--tensorboard", "0"])
d = materialize_gowalla(GOWALLA_NPZ, "/tmp/lgcn_ps_gowalla")
with contextlib.redirect_stdout(io.StringIO()):
    ds = pkg.dataloader.Loader(w.config, path=d)
    pkg.sampling.seed(2020); pkg.utils.set_seed(2020)
    m = pkg.model.LightGCN(w.config, ds).to(w.device)
    bpr = pkg.utils.BPRLoss(m, w.config)
    u, p, n = pkg.Procedure.sample_epoch_to_device(ds, w.device)
B = 2048
steps = len(u) // B
out = {}
def loop(uu, pp, nn, lazy):
    bpr.lazy = lazy
    tot = 0.0
    for s in range(steps):
        r = bpr.stageOne(uu[s * B:(s + 1) * B], pp[s * B:(s + 1) * B], nn[s * B:(s + 1) * B])
        if not lazy:
            tot += r
    return tot
for name, ids in (("int64 ids", (u.long(), p.long(), n.long())), ("int32 ids", (u, p, n))):
    for lazy in (True, False):
        loop(*ids, lazy); torch.cuda.synchronize()
        t0 = time.perf_counter(); loop(*ids, lazy); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[f"stageOne per step, {name}, {'no host sync (bpr.lazy)' if lazy else '.item() per step (the reference)'}"] = steps / dt
bpr.lazy = False; bpr.deferred = True
for name, ids in (("int64 ids", (u.long(), p.long(), n.long())),):
    def loop_d():
        tot = 0.0
        for s in range(steps):
            tot += bpr.stageOne(ids[0][s * B:(s + 1) * B], ids[1][s * B:(s + 1) * B], ids[2][s * B:(s + 1) * B])
        return f"{tot / steps:.6f}"
    loop_d(); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = loop_d(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[f"stageOne per step, {name}, --lazy_loss 1 (DeferredLoss summed and formatted once per epoch: mean {r})"] = steps / dt
bpr.deferred = False
m.fused_epoch(u, p, n, B); torch.cuda.synchronize()
t0 = time.perf_counter(); m.fused_epoch(u[:steps * B], p[:steps * B], n[:steps * B], B); torch.cuda.synchronize()
out["one C call per epoch (Procedure.BPR_train_original)"] = steps / (time.perf_counter() - t0)
print(json.dumps(out, indent=1))
