#!/usr/bin/env python3
"""GPU: time one full Procedure.Test on Gowalla through the fused evaluation kernels and through the torch
harness; report the matrix-core rate of k_eval_topk (2*users*items*d flop / time vs the fp32 MFMA peak)."""
import importlib, io, contextlib, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
sys.argv = [sys.argv[0]]
pkg = importlib.import_module(bench.PKG)
w = pkg.world; w.configure(["--tensorboard", "0", "--checkpoint_dir", "/tmp/lgcn_eval_ckpt"])
dev = torch.device("cuda", 0)
d = bench.materialize_gowalla(bench.GOWALLA_NPZ, "/tmp/lgcn_bench_data/gowalla_r0")
with contextlib.redirect_stdout(io.StringIO()):
    ds = pkg.dataloader.Loader(w.config, path=d)
    pkg.utils.set_seed(2020)
    m = pkg.model.LightGCN(w.config, ds).to(dev)
    m.eval()
    out = {}
    for fused in (1, 0):
        w.config['eval_fused'] = fused
        pkg.Procedure.Test(ds, m, 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            r = pkg.Procedure.Test(ds, m, 0)
        torch.cuda.synchronize()
        out["fused" if fused else "torch"] = {"ms_per_Test": (time.perf_counter() - t0) / 5 * 1e3, "recall": float(r["recall"][0])}
    ev = ds._lgcn_eval_index
    E = m.propagated_table()
    for _ in range(3):
        pkg.Procedure._test_fused(m, ev, 20)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    topk = torch.empty(len(ev.users), 20, dtype=torch.int32, device=dev)
    L = pkg._lib; lib = L.load()
    e0.record()
    for _ in range(10):
        L.check(lib.lgcn_eval_topk(L.tp(E), m.n_users, m.m_items, m.latent_dim, L.tp(ev.users32), len(ev.users), L.tp(ev.train_ptr),
                                   L.tp(ev.train_idx32), 20, L.tp(topk), None, L.current_stream()), "topk")
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    flop = 2.0 * len(ev.users) * m.m_items * m.latent_dim
    out["k_eval_topk"] = {"ms": t * 1e3, "TFLOP/s": flop / t / 1e12, "peak_fp32_mfma_TFLOP/s": 157.3, "frac": flop / t / 1e12 / 157.3,
                          "users": len(ev.users), "items": m.m_items, "d": m.latent_dim}
print(json.dumps(out))
