#!/usr/bin/env python3
"""GPU: steps/s on Gowalla (K=3, d=64, B=2048) with the fork's optional branches on -- popularity gate, item-item
smoothing (a co-occurrence graph built here: pairs seen >= 3 times, rows scaled to sum 1), both -- inside the fused HIP
step (per-step host calls through BPRLoss.stageOne AND one C call per epoch through fused_epoch) and through the autograd
path (--fused_variants 0: torch MLPs and torch Adam around the HIP propagation kernels), next to the default model.
One JSON line."""
import contextlib, importlib, io, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import scipy.sparse as sp
import torch
import bench
sys.argv = [sys.argv[0]]
pkg = importlib.import_module(bench.PKG)
dev = torch.device("cuda", 0)
d = bench.materialize_gowalla(bench.GOWALLA_NPZ, "/tmp/lgcn_bench_data/gowalla_r0")
z = np.load(bench.GOWALLA_NPZ)
rows = np.repeat(np.arange(len(z["train_ptr"]) - 1), np.diff(z["train_ptr"]))
R = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, z["train_items"])), shape=(len(z["train_ptr"]) - 1, 40981))
C = (R.T @ R).tocsr(); C.setdiag(0); C.data[C.data < 3] = 0; C.eliminate_zeros()
rs = np.asarray(C.sum(axis=1)).ravel(); rs[rs == 0] = 1
C = (sp.diags((1.0 / rs).astype(np.float32)) @ C).tocsr().astype(np.float32)
i2i_path = "/tmp/lgcn_bench_data/i2i_gowalla.npz"
sp.save_npz(i2i_path, C)
out = {"i2i_nnz": int(C.nnz)}
rng = np.random.Generator(np.random.PCG64(1))
B, steps = 2048, 200
U = torch.from_numpy(rng.integers(0, 29858, (steps + 5, B))).to(dev)
P = torch.from_numpy(rng.integers(0, 40981, (steps + 5, B))).to(dev)
N = torch.from_numpy(rng.integers(0, 40981, (steps + 5, B))).to(dev)
for name, flags in (("default", {}), ("pop_gate", {"use_pop_gate": True}),
                    ("item_item", {"use_item_item": True, "i2i_path": i2i_path, "i2i_alpha": 0.2}),
                    ("gate_and_item_item", {"use_pop_gate": True, "use_item_item": True, "i2i_path": i2i_path, "i2i_alpha": 0.2})):
    for fused in ((1,) if not flags else (1, 0)):
        w = pkg.world; w.configure(["--tensorboard", "0", "--fused_variants", str(fused)])
        w.config.update(flags)
        with contextlib.redirect_stdout(io.StringIO()):
            ds = pkg.dataloader.Loader(w.config, path=d)
            pkg.utils.set_seed(2020)
            m = pkg.model.LightGCN(w.config, ds).to(dev)
        bpr = pkg.utils.BPRLoss(m, w.config); bpr.lazy = True
        n = steps if fused else 30
        for i in range(5):
            l = bpr.stageOne(U[i], P[i], N[i])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(5, 5 + n):
            l = bpr.stageOne(U[i], P[i], N[i])
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rec = {"steps_per_sec_stageOne": n / dt, "last_loss": float(l), "fused": bpr.fused}
        if fused:
            uu, pp, nn_ = (t[5:5 + steps].reshape(-1).to(torch.int32) for t in (U, P, N))
            m.fused_epoch(uu[:5 * B], pp[:5 * B], nn_[:5 * B], B)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            losses = m.fused_epoch(uu, pp, nn_, B)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            rec["steps_per_sec_epoch_call"] = steps / dt
            rec["epoch_last_loss"] = float(losses[-1, 0])
            m.check_device_errors()
        out[name + ("" if fused else "_autograd")] = rec
        del bpr, m
print(json.dumps(out))
