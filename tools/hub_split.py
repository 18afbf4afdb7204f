#!/usr/bin/env python3
"""EXPERIMENT (GPU): the matrix-split form of the hot-column idea.  A = A_rest + A_hub with the H most gathered
columns in A_hub; A_rest through the product's k_spmm (lgcn_spmm_csr, its own plan), A_hub X_hub by the persistent-workgroup kernel of
tools/exp/hub_split.hip with X_hub in LDS.  Times full / rest / hub launches per shape and checks rest + hub against full."""
import ctypes as C, importlib, io, contextlib, json, os, subprocess, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
argv = sys.argv[1:]
sys.argv = [sys.argv[0]]
pkg = importlib.import_module(bench.PKG)
L, lib = pkg._lib, pkg._lib.load()
so = os.path.join(REPO, "build", "exp_hub_split.so")
exp = C.CDLL(so)
exp.hub_spmm.restype = C.c_int
exp.hub_spmm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device("cuda", 0)
w = pkg.world; w.configure(["--tensorboard", "0"])


def timeit(fn, reps=100):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name in (argv or ["gowalla", "yelp2018-shaped", "amazon-book-shaped"]):
    d = bench.WORKLOADS[name][4]
    with contextlib.redirect_stdout(io.StringIO()):
        if name == "gowalla":
            ds = pkg.dataloader.Loader(w.config, path=bench.materialize_gowalla(bench.GOWALLA_NPZ, "/tmp/lgcn_bench_data/gowalla_r0"))
        else:
            ds = bench.synthetic_dataset(pkg, name, w.config, dev)
    adj = ds.getSparseGraphCSR()
    N = adj.shape[0]
    ip, ix, vv = adj.indptr.astype(np.int64), adj.indices.astype(np.int32), adj.data.astype(np.float32)
    colcnt = np.bincount(ix, minlength=N)
    H = 48 * 1024 // (d * 4)
    hub = np.sort(np.argsort(-colcnt)[:H]).astype(np.int32)
    slot = np.full(N, -1, np.int32); slot[hub] = np.arange(H, dtype=np.int32)
    is_hub = slot[ix] >= 0
    rows = np.repeat(np.arange(N), np.diff(ip))
    def csr(mask):
        cnt = np.bincount(rows[mask], minlength=N)
        p = np.zeros(N + 1, np.int64); p[1:] = np.cumsum(cnt)
        return p, ix[mask], vv[mask]
    pr, ir, vr = csr(~is_hub)
    ph, ih, vh = csr(is_hub)
    ent = np.empty((len(ih), 2), np.int32); ent[:, 0] = slot[ih]; ent[:, 1] = vh.view(np.int32)
    order, xs = pkg.reorder.row_order("xcd", ds, adj, cache_dir="/tmp/lgcn_hub_" + name)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    g_full = L.Graph(t(ip, torch.int32), t(ix, torch.int32), t(vv, torch.float32), d_max=d, row_order=order, xcd_start=xs)
    g_rest = L.Graph(t(pr, torch.int32), t(ir, torch.int32), t(vr, torch.float32), d_max=d, row_order=order, xcd_start=xs)
    X = torch.randn(N, d, device=dev) * 0.1
    d_hub, d_ph, d_ent = t(hub, torch.int32), t(ph, torch.int32), torch.from_numpy(ent).to(dev)
    Yh = torch.empty(N, d, device=dev)
    blocks = 256
    def hub_launch():
        rc = exp.hub_spmm(X.data_ptr(), d_hub.data_ptr(), H, d_ph.data_ptr(), d_ent.data_ptr(), N, d, Yh.data_ptr(), blocks, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    y_full = g_full.spmm(X); y_rest = g_rest.spmm(X); hub_launch(); torch.cuda.synchronize()
    err = float((y_rest + Yh - y_full).abs().max())
    res = {"workload": name, "d": d, "hub_rows": H, "hub_share_of_nnz": float(is_hub.mean()), "max_abs_err": err,
           "full_us": timeit(lambda: g_full.spmm(X)), "rest_us": timeit(lambda: g_rest.spmm(X)), "hub_us": timeit(hub_launch)}
    res["split_us"] = res["rest_us"] + res["hub_us"]
    print(json.dumps(res), flush=True)
    g_full.close(); g_rest.close()
