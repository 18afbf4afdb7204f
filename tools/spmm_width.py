#!/usr/bin/env python3
"""Development tool (GPU): the dense Gowalla SpMM launch at table widths 32 / 64 / 128 / 256, fp32 and bf16 tables -- what a
column-sharded (tensor-parallel over d) propagation would run per GPU.  JSON lines."""
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
ip, ix, vv = (torch.from_numpy(x).to(dev) for x in (adj.indptr.astype(np.int32), adj.indices.astype(np.int32), adj.data))
L = pkg._lib; lib = L.load()
N = adj.shape[0]
for width in (32, 64, 128, 256):
    g = L.Graph(ip, ix, vv, d_max=width, row_order=order, xcd_start=xs)
    for dt, tdt in ((0, torch.float32), (1, torch.bfloat16)):
        x = (torch.randn(N, width, device=dev) * 0.1).to(tdt); y = torch.empty_like(x)
        for _ in range(10):
            L.check(lib.lgcn_spmm_csr(g.handle, L.tp(x), dt, L.tp(y), dt, width, L.current_stream()), "spmm")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            L.check(lib.lgcn_spmm_csr(g.handle, L.tp(x), dt, L.tp(y), dt, width, L.current_stream()), "spmm")
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 5.0
        print(json.dumps({"d": width, "dtype": "fp32" if dt == 0 else "bf16", "us": us, "gathered_TB_per_s": adj.nnz * width * (4 if dt == 0 else 2) / us / 1e6}))
    g.close()
