#!/usr/bin/env python3
"""GPU: k_eval_topk on synthetic full-catalogue shapes (every user evaluated, random tables, ~45 train positives per user):
lgcn_eval_topk (fp32 product from bf16 planes where the shape allows) beside lgcn_eval_topk_fp32.  One JSON line."""
import importlib, json, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import bench
pkg = importlib.import_module(bench.PKG)
L, lib = pkg._lib, pkg._lib.load()
dev = torch.device("cuda", 0)
out = {}
for name, (nu, mi, d) in {"gowalla-size d=64": (29858, 40981, 64), "amazon-book-size d=128": (52643, 91599, 128), "yelp2018-size d=64": (31668, 38048, 64)}.items():
    g = torch.Generator(device=dev); g.manual_seed(1)
    E = torch.randn(nu + mi, d, device=dev, generator=g) * 0.1
    rng = np.random.Generator(np.random.PCG64(3))
    deg = rng.integers(5, 90, nu)
    ptr = np.zeros(nu + 1, np.int64); ptr[1:] = np.cumsum(deg)
    idx = np.concatenate([np.sort(rng.choice(mi, size=int(k), replace=False)) for k in deg]).astype(np.int32)
    users = torch.arange(nu, dtype=torch.int32, device=dev)
    d_ptr, d_idx = torch.from_numpy(ptr).to(dev), torch.from_numpy(idx).to(dev)
    topk = torch.empty(nu, 20, dtype=torch.int32, device=dev)
    res = {}
    for fn in ("lgcn_eval_topk", "lgcn_eval_topk_fp32"):
        f = getattr(lib, fn)
        for _ in range(2):
            L.check(f(L.tp(E), nu, mi, d, L.tp(users), nu, L.tp(d_ptr), L.tp(d_idx), 20, L.tp(topk), None, L.current_stream()), fn)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.check(f(L.tp(E), nu, mi, d, L.tp(users), nu, L.tp(d_ptr), L.tp(d_idx), 20, L.tp(topk), None, L.current_stream()), fn)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5 * 1e-3
        res[fn] = {"ms": t * 1e3, "frac_fp32_mfma_peak": 2.0 * nu * mi * d / t / 1e12 / 157.3}
    out[name] = res
print(json.dumps(out))
