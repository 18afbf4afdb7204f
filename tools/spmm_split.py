#!/usr/bin/env python3
"""Development tool (GPU): where does a dense SpMM launch spend its time?  Times the Gowalla launch on
the whole graph, on the short rows only (<= 64 nnz), on the long rows only (and their single-chunk / split parts), fp32 and bf16 tables."""
import importlib, io, contextlib, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
sys.argv = [sys.argv[0]]
pkg = importlib.import_module(bench.PKG)
w = pkg.world; w.configure(["--tensorboard", "0"])
dev = torch.device("cuda", 0)
d = bench.materialize_gowalla(bench.GOWALLA_NPZ, "/tmp/lgcn_bench_data/gowalla_r0")
with contextlib.redirect_stdout(io.StringIO()):
    ds = pkg.dataloader.Loader(w.config, path=d)
adj = ds.getSparseGraphCSR()
order, xs = pkg.reorder.row_order("xcd", ds, adj, cache_dir=d)
deg = np.diff(adj.indptr)
ip, ix, vv = (torch.from_numpy(x).to(dev) for x in (adj.indptr.astype(np.int32), adj.indices.astype(np.int32), adj.data))
L = pkg._lib; lib = L.load()
N = adj.shape[0]
sets = {"all": order, "short": order[deg[order] <= 64], "long": order[deg[order] > 64],
        "mid (65..512, one chunk)": order[(deg[order] > 64) & (deg[order] <= 512)], "hub (> 512, split)": order[deg[order] > 512]}
for name, rows in sets.items():
    g = L.Graph(ip, ix, vv, d_max=64, row_order=rows, xcd_start=xs if name == "all" else None)
    for dt, tdt in ((0, torch.float32), (1, torch.bfloat16)):
        x = (torch.randn(N, 64, device=dev) * 0.1).to(tdt); y = torch.empty_like(x)
        for _ in range(10):
            L.check(lib.lgcn_spmm_csr(g.handle, L.tp(x), dt, L.tp(y), dt, 64, L.current_stream()), "spmm")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            L.check(lib.lgcn_spmm_csr(g.handle, L.tp(x), dt, L.tp(y), dt, 64, L.current_stream()), "spmm")
        e1.record(); torch.cuda.synchronize()
        print(json.dumps({"rows": name, "n_rows": int(len(rows)), "nnz": int(deg[rows].sum()), "dtype": "fp32" if dt == 0 else "bf16",
                          "us": e0.elapsed_time(e1) * 10.0}))
    g.close()
